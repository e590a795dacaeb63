export TMPDIR=/tmp
for d in 0 3; do
out=gpurun_out/f_prof_$d; rm -rf $out; mkdir -p $out
GDSP_PEAKS_DBG=$d BURST=20 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 tools/bench_one.py peaks_exact > $out/log.txt 2>&1 || tail -3 $out/log.txt
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "synth" in r["Name"] or "rocclr" in r["Name"]: continue
        print("%-60s calls %5s avg %10.1f us" % (r["Name"][:60], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
