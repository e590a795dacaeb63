#!/bin/bash
# the round's measured numbers (one GPU): bench line, BASELINE configs[2..4], the bench under rocprofv3, per-operator table
out=${1:-gpurun_out}
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > $out/r02_bench_smooth_hann.json 2> $out/r02_bench_smooth_hann.err
: > $out/r02_bench_workloads.jsonl
for w in peaks morph percentile; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline >> $out/r02_bench_workloads.jsonl 2>> $out/r02_bench_workloads.err
  python bench.py --workload $w --nofuse --steps 10 --warmup 2 --no-cpu-baseline >> $out/r02_bench_workloads.jsonl 2>> $out/r02_bench_workloads.err
done
python bench.py --workload peaks --mode exact --steps 10 --warmup 2 --no-cpu-baseline >> $out/r02_bench_workloads.jsonl 2>> $out/r02_bench_workloads.err
python bench.py --workload peaks --mode exact --nofuse --steps 10 --warmup 2 --no-cpu-baseline >> $out/r02_bench_workloads.jsonl 2>> $out/r02_bench_workloads.err
# the same fused chain through the opt-in filtered kernel, for the A/B on one box
GDSP_PEAKS_FILTER=1 python bench.py --workload peaks --mode exact --steps 10 --warmup 2 --no-cpu-baseline >> $out/r02_bench_workloads.jsonl 2>> $out/r02_bench_workloads.err
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/r02_bench_smooth_hann_under_rocprof.json 2> $out/prof_bench.err
# independent chromosomes over three alternating streams (hides the drain between kernels; the fused morphology chain gains most)
for w in morph percentile; do python bench.py --workload $w --streams 3 --steps 10 --warmup 2 --no-cpu-baseline >> $out/r02_bench_workloads.jsonl 2>> $out/r02_bench_workloads.err; done
python tools/bench_ops.py > $out/r02_ops_throughput.txt 2>&1
BURST=10 TAG='(10 launches back to back)' python tools/bench_one.py smooth_exact,smooth_fma,smooth_hann,smooth_hann201,smooth_hann501,smooth_hann1001,smooth_hann1501,smooth_hann1701,smooth_hann2001,smooth_hann4001,smooth_hann5001,smooth_hann20001,smooth_hann50001,sum300,sum500,sum1000,sum2000,sum4000,sum1000real,sum2000real,close,open,dilate20001,peaks_exact,peaks_exact_depth,peaks_fma >> $out/r02_ops_throughput.txt 2>&1
BURST=20 TAG='(20 launches back to back)' python tools/bench_one.py peaks_exact,peaks_exact_depth >> $out/r02_ops_throughput.txt 2>&1
BURST=20 GDSP_PEAKS_FILTER=1 TAG='(20 back to back, GDSP_PEAKS_FILTER=1)' python tools/bench_one.py peaks_exact,peaks_exact_depth >> $out/r02_ops_throughput.txt 2>&1
python tools/bench_percentile.py >> $out/r02_ops_throughput.txt 2>&1
