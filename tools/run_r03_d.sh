mkdir -p gpurun_out
export TMPDIR=/tmp
out=gpurun_out/d_prof; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 bench.py --workload peaks --mode ${MODE:-exact} --steps 10 --warmup 2 > $out/log.txt 2>&1 || tail -3 $out/log.txt
python3 - "$out" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print("%-70s calls %5s avg %10.1f us total %10.1f ms" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"])/1e6))
PY
python3 - <<'PY'
import os, numpy as np
import genodsp_amd as gd
from oracle import cpu
# how many bases does the filter queue on the bench signal / on depth?  (read the control words back)
import ctypes as C
for mode, name in ((1, "real"), (0, "depth")):
    n = 50_000_000
    x = gd.synth_coverage(20240611, 0, 0, n, mode)
    os.environ["GDSP_PEAKS_ROUTE"] = "filter"
    out = gd.smooth_local_extrema(x, 101, 11, True, 0.0)
    gd.sync()
    del os.environ["GDSP_PEAKS_ROUTE"]
    print(name, "survivors", int(np.count_nonzero(out.numpy())), "of", n)
PY
