#!/bin/bash
# the round's numbers on the final build, part 1: bench lines, workloads, the bench under rocprofv3, operator table
O=gpurun_out
python -m pytest tests/test_hip_percentile_binarize.py -x -q > $O/e1_tests.log 2>&1; echo "percentile tests rc=$?"
bash tools/run_round.sh $O r04
tail -2 $O/e1_tests.log; cut -c1-260 $O/r04_bench_smooth_hann.json; wc -l $O/r04_bench_workloads.jsonl; tail -12 $O/r04_ops_throughput.txt
