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
rocprofv3 --kernel-trace --stats --output-format csv -d $out/prof_bench -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/r02_bench_smooth_hann_under_rocprof.json 2> $out/prof_bench.err
python tools/bench_ops.py > $out/r02_ops_throughput.txt 2>&1
python tools/bench_one.py smooth_hann,smooth_hann201,smooth_hann501,smooth_hann1001,smooth_hann2001,sum500,sum1000,sum2000,close,open,dilate20001 >> $out/r02_ops_throughput.txt 2>&1
