#!/bin/bash
# round 4, fourth call: the selects of the resident percentile route in one workgroup's LDS (pc_ls_*) against the digit passes
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_percentile_binarize.py tests/test_hip_multirank.py tests/test_hip_parity.py -x -q -k "percentile or select or rank" > $O/s4_pct_tests.log 2>&1; echo "percentile tests rc=$?" > $O/s4_summary.txt
python -m pytest tests/test_cli_hip.py tests/test_cli_seams.py -x -q -k "percentile or rccl or random or seam" >> $O/s4_pct_tests.log 2>&1; echo "cli tests rc=$?" >> $O/s4_summary.txt
GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/selects in LDS:  /' >> $O/s4_summary.txt
GDSP_PERCENTILE_LDS_SELECT=0 GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/digit passes:    /' >> $O/s4_summary.txt
python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s4_bench_pct.json 2> $O/s4_bench_pct.err; echo "bench pct rc=$?" >> $O/s4_summary.txt
bash tools/prof_any.sh $O/prof4 percentile 3 248956422 > /dev/null 2>&1
cat $O/s4_summary.txt; tail -4 $O/s4_pct_tests.log; cut -c1-300 $O/s4_bench_pct.json; cat $O/prof4/percentile.txt
