#!/bin/bash
# the round's measured numbers (one GPU): bench line, BASELINE configs[2..4], the bench under rocprofv3, per-operator table
# usage: tools/run_round.sh [outdir] [round tag, default r03]
out=${1:-gpurun_out}; R=${2:-r05}
export TMPDIR=/tmp
python bench.py --steps 20 --warmup 3 > $out/${R}_bench_smooth_hann.json 2> $out/${R}_bench_smooth_hann.err
: > $out/${R}_bench_workloads.jsonl
W=$out/${R}_bench_workloads
for w in peaks morph percentile; do
  python bench.py --workload $w --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err
  python bench.py --workload $w --nofuse --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err
done
# configs[2] in the reference's arithmetic: the filtered route (default), the direct kernel (GDSP_PEAKS_FILTER=0), unfused
python bench.py --workload peaks --mode exact --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err
GDSP_PEAKS_FILTER=0 python bench.py --workload peaks --mode exact --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err
python bench.py --workload peaks --mode exact --nofuse --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err
# --smooth=fma in front of localmax: the filtered route (default since round 4) against the direct kernel
GDSP_PEAKS_FILTER=exact python bench.py --workload peaks --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err
# one launch per chromosome instead of one per operator (what round 2 measured), for the A/B on one box
python bench.py --launch chromosome --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err
for w in morph percentile; do python bench.py --launch chromosome --workload $w --steps 10 --warmup 2 --no-cpu-baseline >> $W.jsonl 2>> $W.err; done
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/${R}_bench_smooth_hann_under_rocprof.json 2> $out/prof_bench.err
python tools/bench_ops.py > $out/${R}_ops_throughput.txt 2>&1
BURST=10 TAG='(10 launches back to back)' python tools/bench_one.py smooth_exact,smooth_fma,smooth_hann,smooth_hann201,smooth_hann501,smooth_hann1001,smooth_hann1501,smooth_hann1701,smooth_hann2001,smooth_hann4001,smooth_hann5001,smooth_hann20001,smooth_hann50001,sum300,sum500,sum1000,sum2000,sum4000,sum1000real,sum2000real,close,open,dilate20001,peaks_exact,peaks_exact_depth,peaks_fma >> $out/${R}_ops_throughput.txt 2>&1
BURST=20 TAG='(20 launches back to back)' python tools/bench_one.py peaks_exact,peaks_exact_depth >> $out/${R}_ops_throughput.txt 2>&1
BURST=20 GDSP_PEAKS_FILTER=0 TAG='(20 back to back, GDSP_PEAKS_FILTER=0: the direct kernel)' python tools/bench_one.py peaks_exact,peaks_exact_depth >> $out/${R}_ops_throughput.txt 2>&1
GENOME=1 python tools/bench_percentile.py >> $out/${R}_ops_throughput.txt 2>&1
GDSP_PERCENTILE_LDS_SELECT=0 GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/(digit passes, GDSP_PERCENTILE_LDS_SELECT=0) /' >> $out/${R}_ops_throughput.txt
GDSP_PERCENTILE_COUNT_PER_SOURCE=1 GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/(a counting launch per source, GDSP_PERCENTILE_COUNT_PER_SOURCE=1) /' >> $out/${R}_ops_throughput.txt
