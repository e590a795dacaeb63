#!/bin/bash
# usage: tools/exp_kernel.sh "<bench.py args>" <kernel name pattern> <source file under genodsp_amd/csrc> FLAG1 FLAG2 ...
# rebuilds the source with -D<FLAG> per variant ("plain" = none) and prints the kernel's average duration while bench.py
# runs under rocprofv3 (a warm GPU: a few launches from a cold process run at a third of the speed)
args="$1"; pat="$2"; src="$3"; shift 3
export TMPDIR=/tmp
BASE='--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include'
for f in "$@"; do
  touch genodsp_amd/csrc/$src
  if [ "$f" = "plain" ]; then make -C genodsp_amd/csrc HIPFLAGS="$BASE" > /dev/null 2>&1; else make -C genodsp_amd/csrc HIPFLAGS="$BASE -D$f" > /dev/null 2>&1; fi
  out=gpurun_out/ek_$f; rm -rf $out; mkdir -p $out
  rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py $args --no-cpu-baseline > $out/log.txt 2>&1 || tail -3 $out/log.txt
  python3 - "$out" "$pat" "$f" <<'PY'
import csv, glob, sys
out, pat, tag = sys.argv[1:4]
for f in glob.glob(out + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print("%-16s %-56s calls %4s avg %9.1f us total %9.1f ms" % (tag, r["Name"][:56], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
done
