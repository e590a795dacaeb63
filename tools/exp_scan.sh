#!/bin/bash
# usage: tools/exp_scan.sh FLAG1 FLAG2 ...   ("plain" = none; "A+B" = -DA -DB)
# rebuilds genodsp_amd/csrc/gdsp_sums.hip with the flags of each variant and runs tools/bench_scan.py (cumulativesum, clump)
BASE='--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include'
for f in "$@"; do
  touch genodsp_amd/csrc/gdsp_sums.hip
  defs=""
  if [ "$f" != "plain" ]; then for d in ${f//+/ }; do defs="$defs -D$d"; done; fi
  make -C genodsp_amd/csrc HIPFLAGS="$BASE $defs" > /dev/null 2>&1 || { echo "build failed: $f"; continue; }
  echo "== $f"
  python3 tools/bench_scan.py ${SCAN_N:-248956422} 7 2>&1 | grep cumulativesum
done
touch genodsp_amd/csrc/gdsp_sums.hip
make -C genodsp_amd/csrc > /dev/null 2>&1
