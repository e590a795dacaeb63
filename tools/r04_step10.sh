#!/bin/bash
# round 4, tenth call: the peaks filter's form for flat stretches (a run's value written to all its bases), chosen by the host
# from the probe's counts; GDSP_PEAKS_FLAT=0 is the route without it
O=gpurun_out
python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_batch.py tests/test_hip_u32max.py -x -q -k "peaks or filtered or fused or smooth_local or local or smooth_and" > $O/s10_tests.log 2>&1; echo "peaks tests rc=$?" > $O/s10_summary.txt
for f in 1 0 1 0; do
GDSP_PEAKS_FLAT=$f BURST=10 TAG="GDSP_PEAKS_FLAT=$f" python tools/bench_one.py peaks_exact,peaks_exact_depth,peaks_fma 2>&1 | tail -3 >> $O/s10_summary.txt
done
for f in 1 0; do
GDSP_PEAKS_FLAT=$f python bench.py --workload peaks --mode exact --steps 10 --warmup 3 --no-cpu-baseline 2> $O/s10_bench.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('GDSP_PEAKS_FLAT=$f bench peaks exact', d['value'], d['ms_per_step'])" >> $O/s10_summary.txt
done
cat $O/s10_summary.txt; tail -3 $O/s10_tests.log
