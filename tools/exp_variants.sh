#!/bin/bash
# usage: tools/exp_variants.sh "<bench args>" FLAG1 FLAG2 ...   (rebuilds gdsp_percentile.hip with -D<FLAG> per variant; "" = plain)
args="$1"; shift
BASE='--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include'
for f in "$@"; do
  touch genodsp_amd/csrc/gdsp_percentile.hip
  if [ "$f" = "plain" ]; then make -C genodsp_amd/csrc HIPFLAGS="$BASE" > /dev/null 2>&1; else make -C genodsp_amd/csrc HIPFLAGS="$BASE -D$f" > /dev/null 2>&1; fi
  python bench.py $args --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'])"
done
