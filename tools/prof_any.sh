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
if [ -n "$SQ" ]; then   # SQ=1: two more passes of issue / wait / LDS counters (8 SQ slots per pass)
rocprofv3 --kernel-trace --output-format csv -d $out/${op}_sq1 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS -- python3 tools/prof_op.py $op $launches $n > $out/${op}_sq1.log 2>&1 || tail -3 $out/${op}_sq1.log
rocprofv3 --kernel-trace --output-format csv -d $out/${op}_sq2 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU GRBM_GUI_ACTIVE -- python3 tools/prof_op.py $op $launches $n > $out/${op}_sq2.log 2>&1 || tail -3 $out/${op}_sq2.log
fi
lib=$(python3 -c "import genodsp_amd as gd; print(gd.lib().gdsp_version().decode())" 2>/dev/null | tail -1)
python3 - "$out" "$op" "$n" "$lib" <<'PY' > $out/$op.txt
import csv, glob, sys, collections
out, op, n = sys.argv[1], sys.argv[2], int(sys.argv[3])
print("# %s on %d bases; rocprofv3 --kernel-trace --stats, then --pmc FETCH_SIZE / WRITE_SIZE in separate passes" % (op, n))
print("# library: %s" % sys.argv[4])
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
sq = collections.defaultdict(lambda: collections.defaultdict(list))
for which in ("sq1", "sq2"):
    for f in glob.glob(out + "/%s_%s/*/*counter_collection.csv" % (op, which)):
        for r in csv.DictReader(open(f)):
            sq[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, c in sq.items():
    if k not in stats or stats[k][2] < 5:
        continue
    print("\n# SQ counters, mean per dispatch: %s" % k[:100])
    m = {name: sum(v) / len(v) for name, v in c.items()}
    for name in sorted(m):
        print("%-24s %.6g" % (name, m[name]))
    if "SQ_WAVE_CYCLES" in m:
        wc = m["SQ_WAVE_CYCLES"]
        print("share of wave-cycles: issuing %.3f, issue-stalled %.3f (of which LDS %.3f), parked (waitcnt/barrier) %.3f" % (
            m.get("SQ_ACTIVE_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_ANY", 0) / wc, m.get("SQ_WAIT_INST_LDS", 0) / wc, m.get("SQ_WAIT_ANY", 0) / wc))
    if "GRBM_GUI_ACTIVE" in m and k in stats:
        print("effective clock %.2f GHz (GRBM_GUI_ACTIVE / 8 XCDs / kernel time)" % (m["GRBM_GUI_ACTIVE"] / 8 / stats[k][1]))
PY
cat $out/$op.txt
