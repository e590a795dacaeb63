#!/bin/bash
# where a tile of the peaks filter spends its life: builds gdsp_peaks.hip with -DPK_STAMPS (thread 0 of every workgroup adds
# the core-clock cycles since its entry to a counter at seven points) and runs the fused chain over one chromosome
# usage: tools/exp_peaks_stamps.sh [extra -D flags]
BASE='--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wall -Wno-unused-result -I../../include'
defs="-DPK_STAMPS"; for d in "$@"; do defs="$defs -D$d"; done
touch genodsp_amd/csrc/gdsp_peaks.hip
make -C genodsp_amd/csrc HIPFLAGS="$BASE $defs" > /dev/null 2>&1 || { echo "build failed"; exit 1; }
python3 - <<'PY'
import ctypes as C, sys
sys.path.insert(0, ".")
import genodsp_amd as gd
n = 248956422
L = C.CDLL(gd.SO_PATH)
names = ["", "staged (loads landed, first barrier)", "phase 1 + barrier", "phase 2 (thread 0)", "statistics, words, edges, B1",
         "classification, lists, barrier", "exact values (wave 0's chain), in place", "store loop issued"]
for kind, mode in (("real-valued", 1), ("read depth", 0)):
    v = gd.synth_coverage(20240611, 0, 0, n, mode)
    out = v.like()
    S = gd.Stream()
    for _ in range(3):
        gd.smooth_local_extrema(v, 101, 11, True, 0.0, out=out, mode=gd.FIR_EXACT, stream=S.handle)
    gd.sync()
    buf = (C.c_ulonglong * 16)()
    L.gdsp_peaks_stamps(buf, 1)
    reps = 10
    e0, e1 = gd.Event(), gd.Event()
    e0.record(S.handle)
    for _ in range(reps):
        gd.smooth_local_extrema(v, 101, 11, True, 0.0, out=out, mode=gd.FIR_EXACT, stream=S.handle)
    e1.record(S.handle)
    gd.sync()
    ms = e0.elapsed_ms(e1) / reps
    L.gdsp_peaks_stamps(buf, 1)
    tiles = buf[15]
    print("%s: %.3f ms per call, %d tiles stamped; mean core-clock cycles since a workgroup's entry (thread 0):" % (kind, ms, tiles))
    prev = 0.0
    for i in range(1, 8):
        c = buf[i] / max(tiles, 1)
        print("   %-44s %9.0f   (+%.0f)" % (names[i], c, c - prev))
        prev = c
    for i, what in ((8, "... classified (before the lists)"), (9, "... leaders listed"), (10, "... all lists made")):
        print("   %-44s %9.0f" % (what, buf[i] / max(tiles, 1)))
PY
touch genodsp_amd/csrc/gdsp_peaks.hip
make -C genodsp_amd/csrc > /dev/null 2>&1
