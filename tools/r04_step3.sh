#!/bin/bash
# round 4, third call: the counting pass as one launch per device (A/B against a launch per source), the percentile tests,
# and counters for the LDS-DMA form of the sliding FIR
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_percentile_binarize.py tests/test_hip_multirank.py tests/test_hip_parity.py -x -q -k "percentile or select or rank" > $O/s3_pct_tests.log 2>&1; echo "percentile tests rc=$?" > $O/s3_summary.txt
python -m pytest tests/test_cli_hip.py -x -q -k "percentile or rccl" >> $O/s3_pct_tests.log 2>&1; echo "cli percentile tests rc=$?" >> $O/s3_summary.txt
GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/one launch per device: /' >> $O/s3_summary.txt
GDSP_PERCENTILE_COUNT_PER_SOURCE=1 GENOME=1 ROUTES=resident python tools/bench_percentile.py 2>&1 | sed 's/^/a launch per source:   /' >> $O/s3_summary.txt
python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s3_bench_pct.json 2> $O/s3_bench_pct.err; echo "bench pct rc=$?" >> $O/s3_summary.txt
GDSP_PERCENTILE_COUNT_PER_SOURCE=1 python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s3_bench_pct_per_source.json 2> $O/s3_bench_pct_per_source.err
GDSP_FIR_SLIDE=2 GDSP_FIR_SLIDE_STRIP=1024 SQ=1 bash tools/prof_any.sh $O/prof_slide smooth_exact 3 > /dev/null 2>&1
cat $O/s3_summary.txt; cut -c1-400 $O/s3_bench_pct.json $O/s3_bench_pct_per_source.json; cat $O/prof_slide/smooth_exact.txt
