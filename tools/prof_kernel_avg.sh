#!/bin/bash
# average duration of the kernels matching <pattern> while tools/prof_op.py <op> runs: tools/prof_kernel_avg.sh <tag> <op> <pattern> [n]
tag=$1; op=$2; pat=$3; n=${4:-248956422}
export TMPDIR=/tmp
out=gpurun_out/pk_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/prof_op.py $op 4 $n > $out/log.txt 2>&1 || tail -3 $out/log.txt
python3 - "$out" "$pat" "$tag" <<'PY'
import csv, glob, sys
out, pat, tag = sys.argv[1:4]
for f in glob.glob(out + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Name"]:
            print("%-14s %-60s calls %4s avg %9.1f us" % (tag, r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
