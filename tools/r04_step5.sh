#!/bin/bash
# round 4, fifth call: the LDS selects after their rework (wave-local sort stages, aggregated appends, searches side by side);
# the peaks filter on block sums of all 101 taps (no direct end taps)
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_percentile_binarize.py tests/test_hip_multirank.py tests/test_hip_parity.py -x -q -k "percentile or select or rank" > $O/s5_tests.log 2>&1; echo "percentile tests rc=$?" > $O/s5_summary.txt
python -m pytest tests/test_hip_parity.py tests/test_hip_fullsize.py tests/test_hip_batch.py tests/test_hip_u32max.py -x -q -k "peaks or filtered or fused or smooth_local or local or smooth_and" >> $O/s5_tests.log 2>&1; echo "peaks tests rc=$?" >> $O/s5_summary.txt
python -m pytest tests/test_cli_hip.py tests/test_cli_seams.py -x -q >> $O/s5_tests.log 2>&1; echo "cli tests rc=$?" >> $O/s5_summary.txt
GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/selects in LDS:  /' >> $O/s5_summary.txt
GDSP_PERCENTILE_LDS_SELECT=0 GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/digit passes:    /' >> $O/s5_summary.txt
BURST=10 TAG="filter on all-tap block sums" python tools/bench_one.py peaks_exact,peaks_exact_depth 2>&1 | tail -2 >> $O/s5_summary.txt
python bench.py --workload peaks --mode exact --steps 10 --warmup 3 --no-cpu-baseline > $O/s5_bench_peaks.json 2> $O/s5_bench_peaks.err; echo "bench peaks rc=$?" >> $O/s5_summary.txt
python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s5_bench_pct.json 2> $O/s5_bench_pct.err; echo "bench pct rc=$?" >> $O/s5_summary.txt
bash tools/prof_any.sh $O/prof5 percentile 3 248956422 > /dev/null 2>&1
SQ=1 bash tools/prof_any.sh $O/prof5 peaks_exact_batch 3 > /dev/null 2>&1
cat $O/s5_summary.txt; tail -3 $O/s5_tests.log; cut -c1-200 $O/s5_bench_peaks.json $O/s5_bench_pct.json; cat $O/prof5/percentile.txt; head -12 $O/prof5/peaks_exact_batch.txt
