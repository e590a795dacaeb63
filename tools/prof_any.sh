#!/bin/bash
# rocprofv3 evidence for one operator (tools/prof_op.py <op>): kernel trace + stats, then FETCH_SIZE and WRITE_SIZE in
# passes of their own (the guide's HBM recipe; the program itself follows `--`).
# usage: tools/prof_any.sh <outdir> <op> [launches] [n]         -> <outdir>/<op>_{stats,fetch,write}/..., <outdir>/<op>.txt
out=$1; op=$2; launches=${3:-3}; n=${4:-145138636}
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/${op}_stats -- python3 tools/prof_op.py $op $launches $n > $out/${op}_stats.log 2>&1 || tail -3 $out/${op}_stats.log
rocprofv3 --kernel-trace --output-format csv -d $out/${op}_fetch --pmc FETCH_SIZE -- python3 tools/prof_op.py $op $launches $n > $out/${op}_fetch.log 2>&1 || tail -3 $out/${op}_fetch.log
rocprofv3 --kernel-trace --output-format csv -d $out/${op}_write --pmc WRITE_SIZE -- python3 tools/prof_op.py $op $launches $n > $out/${op}_write.log 2>&1 || tail -3 $out/${op}_write.log
python3 - "$out" "$op" "$n" <<'PY' > $out/$op.txt
import csv, glob, sys, collections
out, op, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
print("# %s on %d bases; rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE in separate passes" % (op, n))
print("# FETCH_SIZE doubled (gfx950 tallies 128-B read requests at 64 B: MI355X_MICROARCH.md, HBM); units KiB -> bytes")
stats = {}
for f in glob.glob(out + "/%s_stats/*/*kernel_stats.csv" % op):
    for r in csv.DictReader(open(f)):
        stats[r["Name"]] = (int(r["Calls"]), float(r["AverageNs"]), float(r["Percentage"]))
cnt = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("fetch", "write"):
    for f in glob.glob(out + "/%s_%s/*/*counter_collection.csv" % (op, which)):
        for r in csv.DictReader(open(f)):
            cnt[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("%-74s %6s %11s %7s %14s %14s %9s" % ("kernel", "calls", "avg us", "%time", "fetch B/launch", "write B/launch", "B/base"))
for k, (calls, avg, pct) in sorted(stats.items(), key=lambda kv: -kv[1][2]):
    c = cnt.get(k, {})
    fetch = 2 * 1024 * sum(c["FETCH_SIZE"]) / len(c["FETCH_SIZE"]) if c.get("FETCH_SIZE") else float("nan")
    write = 1024 * sum(c["WRITE_SIZE"]) / len(c["WRITE_SIZE"]) if c.get("WRITE_SIZE") else float("nan")
    print("%-74s %6d %11.1f %7.2f %14.0f %14.0f %9.2f" % (k[:74], calls, avg / 1e3, pct, fetch, write, (fetch + write) / n))
PY
cat $out/$op.txt
