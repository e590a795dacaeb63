#!/bin/bash
# round 4, first measurement call: the sliding-accumulator FIR, the peaks filter's in-place ties, the split resident percentile
set -o pipefail
O=gpurun_out
: > $O/s1_summary.txt
for strip in 0 1024 2048 4096; do
  GDSP_FIR_SLIDE=1 GDSP_FIR_SLIDE_STRIP=$strip BURST=10 TAG="slide strip=$strip" python tools/bench_one.py smooth_exact 2>&1 | tail -1 | tee -a $O/s1_summary.txt
done
GDSP_FIR_SLIDE=0 BURST=10 TAG="direct" python tools/bench_one.py smooth_exact,smooth_fma,smooth_hann 2>&1 | tail -3 | tee -a $O/s1_summary.txt
BURST=10 TAG="filtered" python tools/bench_one.py peaks_exact,peaks_exact_depth 2>&1 | tail -2 | tee -a $O/s1_summary.txt
GENOME=1 python tools/bench_percentile.py 2>&1 | tee -a $O/s1_summary.txt
python bench.py --workload peaks --mode exact --steps 10 --warmup 3 --no-cpu-baseline > $O/s1_bench_peaks.json 2> $O/s1_bench_peaks.err; echo "bench peaks rc=$?" | tee -a $O/s1_summary.txt
python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s1_bench_pct.json 2> $O/s1_bench_pct.err; echo "bench pct rc=$?" | tee -a $O/s1_summary.txt
GDSP_FIR_SLIDE=1 python bench.py --mode exact --steps 10 --warmup 3 --no-cpu-baseline > $O/s1_bench_smooth_slide.json 2> $O/s1_bench_smooth_slide.err; echo "bench smooth slide rc=$?" | tee -a $O/s1_summary.txt
cat $O/s1_bench_peaks.json $O/s1_bench_pct.json $O/s1_bench_smooth_slide.json | cut -c1-700
