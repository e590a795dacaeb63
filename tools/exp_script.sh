#!/bin/bash
# usage: tools/exp_script.sh "<command>" <source file under genodsp_amd/csrc> FLAG1 FLAG2 ...
# rebuilds the source with -D<FLAG> per variant ("plain" = none) and runs the command after each build
cmd="$1"; src="$2"; shift 2
BASE='--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include'
for f in "$@"; do
  touch genodsp_amd/csrc/$src
  if [ "$f" = "plain" ]; then make -C genodsp_amd/csrc HIPFLAGS="$BASE" > /dev/null 2>&1; else make -C genodsp_amd/csrc HIPFLAGS="$BASE -D$f" > /dev/null 2>&1; fi
  echo "== $f"; eval "$cmd"
done
