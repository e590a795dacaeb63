#!/bin/bash
# round 4, ninth call: the counting pass one tile per workgroup (per-tile slots) against persistent workgroups
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_percentile_binarize.py tests/test_hip_multirank.py tests/test_hip_parity.py -x -q -k "percentile or select or rank" > $O/s9_tests.log 2>&1; echo "percentile tests rc=$?" > $O/s9_summary.txt
python -m pytest tests/test_cli_hip.py tests/test_cli_seams.py -x -q -k "percentile or rccl or random or seam" >> $O/s9_tests.log 2>&1; echo "cli tests rc=$?" >> $O/s9_summary.txt
for t in 1 0; do
GDSP_PERCENTILE_TILES=$t GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed "s/^/tiles=$t: /" >> $O/s9_summary.txt
GDSP_PERCENTILE_TILES=$t python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline 2> $O/s9_bench_$t.err | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('tiles=$t: bench percentile=binarize', d['value'], 'Gbases/s', d['ms_per_step'], 'ms', d['roofline']['frac'], d['percentile_stats'])" >> $O/s9_summary.txt
done
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof9 -- python3 tools/prof_op.py percentile_binarize_genome 3 > $O/prof9.log 2>&1
cat $O/s9_summary.txt; tail -3 $O/s9_tests.log
