#!/bin/bash
# all four BASELINE GPU configs on one GPU, one JSON line each (chains fused and unfused)
for w in "smooth" "peaks" "peaks --nofuse" "morph" "morph --nofuse" "percentile"; do
  timeout -k 10 300 python bench.py --workload $w --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null
done
