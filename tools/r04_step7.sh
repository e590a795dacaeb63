#!/bin/bash
# where the fused configs[4] step of bench.py spends 20 ms with the LDS selects (9.3 with the digit passes)
O=gpurun_out
export TMPDIR=/tmp
python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s7_a.json 2> $O/s7_a.err
GDSP_PERCENTILE_LDS_SELECT=0 python bench.py --workload percentile --steps 10 --warmup 3 --no-cpu-baseline > $O/s7_b.json 2> $O/s7_b.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof7 -- python3 bench.py --workload percentile --steps 5 --warmup 2 --no-cpu-baseline > $O/s7_c.json 2> $O/s7_c.err
cut -c1-220 $O/s7_a.json $O/s7_b.json $O/s7_c.json
