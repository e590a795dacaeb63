#!/bin/bash
# round 4, eighth call: the pick kernel's final scan with loads in flight; the fused counting pass under other launch shapes
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_percentile_binarize.py -x -q > $O/s8_tests.log 2>&1; echo "percentile tests rc=$?" > $O/s8_summary.txt
ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/selects in LDS:  /' >> $O/s8_summary.txt
bash tools/prof_any.sh $O/prof8 percentile 3 248956422 > /dev/null 2>&1
bash tools/exp_kernel.sh "--workload percentile --steps 5 --warmup 2" "pc_partition_tab_kernel<2, false, true, true>" gdsp_percentile.hip plain PC_WGS_PER_CU=3 PC_WGS_PER_CU=4 PC_TILES_PER_WG=4 PC_TILES_PER_WG=64 >> $O/s8_summary.txt 2>&1
cat $O/s8_summary.txt; tail -3 $O/s8_tests.log; cat $O/prof8/percentile.txt
