#!/bin/bash
# round 4, sixth call: the LDS selects, second rework (a stage's loads before its stores, 2048-key grids, one global atomic per workgroup)
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_percentile_binarize.py tests/test_hip_multirank.py tests/test_hip_parity.py -x -q -k "percentile or select or rank" > $O/s6_tests.log 2>&1; echo "percentile tests rc=$?" > $O/s6_summary.txt
python -m pytest tests/test_cli_hip.py tests/test_cli_seams.py -x -q -k "percentile or rccl or random or seam" >> $O/s6_tests.log 2>&1; echo "cli tests rc=$?" >> $O/s6_summary.txt
GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/selects in LDS:  /' >> $O/s6_summary.txt
GDSP_PERCENTILE_LDS_SELECT=0 GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/digit passes:    /' >> $O/s6_summary.txt
python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s6_bench_pct.json 2> $O/s6_bench_pct.err; echo "bench pct rc=$?" >> $O/s6_summary.txt
bash tools/prof_any.sh $O/prof6 percentile 3 248956422 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof6/percentile_genome_stats -- python3 tools/prof_op.py percentile_genome 3 > $O/prof6/percentile_genome.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof6/percentile_binarize_genome_stats -- python3 tools/prof_op.py percentile_binarize_genome 3 > $O/prof6/percentile_binarize_genome.log 2>&1
cat $O/s6_summary.txt; tail -3 $O/s6_tests.log; cut -c1-200 $O/s6_bench_pct.json; cat $O/prof6/percentile.txt
