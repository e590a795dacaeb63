#!/usr/bin/env python3
"""Time one operator of tools/prof_op.py's list with HIP events: python3 tools/bench_one.py <op> [n] [reps]"""
import os
import subprocess
import sys
sys.argv = [sys.argv[0]] + sys.argv[1:]
op = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 248956422
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C  # noqa: E402
import genodsp_amd as gd  # noqa: E402
gd.set_device(0)
S = gd.Stream(); s = S.handle
depth = gd.synth_coverage(20240611, 0, 0, n, 0, stream=s)
real = gd.synth_coverage(20240611, 0, 0, n, 1, stream=s)
a = gd.DeviceVector(n); b = gd.DeviceVector(n)
l, r = gd.split_length(1001)
work = gd.DeviceBuffer(max(gd.lib().gdsp_clump_work(n), gd.lib().gdsp_cumulative_sum_work(n)))
cp = lambda: gd.call("gdsp_memcpy_d2d", a.ptr, depth.ptr, n * 8, gd._sp(s))
OPS = {
    "clump": (cp, lambda: gd.call("gdsp_clump", a.ptr, n, 30.5, 1000, 1, 1.0, 0.0, C.c_void_p(work.ptr), gd._sp(s))),
    "cumsum": (cp, lambda: gd.call("gdsp_cumulative_sum", a.ptr, n, C.c_void_p(work.ptr), gd._sp(s))),
    "sum1000": (cp, lambda: gd.window_sum(a, 1000, stream=s)),
    "sum2000": (cp, lambda: gd.window_sum(a, 2000, stream=s)),
    "sum500": (cp, lambda: gd.window_sum(a, 500, stream=s)),
    "sum600": (cp, lambda: gd.window_sum(a, 600, stream=s)),
    "sum600real": (lambda: gd.call("gdsp_memcpy_d2d", a.ptr, real.ptr, n * 8, gd._sp(s)), lambda: gd.window_sum(a, 600, stream=s)),
    "sum400real": (lambda: gd.call("gdsp_memcpy_d2d", a.ptr, real.ptr, n * 8, gd._sp(s)), lambda: gd.window_sum(a, 400, stream=s)),
    "sum300": (cp, lambda: gd.window_sum(a, 300, stream=s)),
    "sum1000real": (lambda: gd.call("gdsp_memcpy_d2d", a.ptr, real.ptr, n * 8, gd._sp(s)), lambda: gd.window_sum(a, 1000, stream=s)),
    "sum2000real": (lambda: gd.call("gdsp_memcpy_d2d", a.ptr, real.ptr, n * 8, gd._sp(s)), lambda: gd.window_sum(a, 2000, stream=s)),
    "sum4000": (cp, lambda: gd.window_sum(a, 4000, stream=s)),
    "sum8000": (cp, lambda: gd.window_sum(a, 8000, stream=s)),
    "sum300real": (lambda: gd.call("gdsp_memcpy_d2d", a.ptr, real.ptr, n * 8, gd._sp(s)), lambda: gd.window_sum(a, 300, stream=s)),
    "sum100": (cp, lambda: gd.window_sum(a, 100, stream=s)),
    "close": (None, lambda: gd.close(depth, 1001, out=b, stream=s)),
    "open": (None, lambda: gd.open_(depth, 1001, out=b, stream=s)),
    "close100": (None, lambda: gd.close(depth, 100, out=b, stream=s)),
    "localmax3": (None, lambda: gd.localmax(real, 3, out=b, stream=s)),
    "bestmax33": (None, lambda: gd.best_extrema(real, 33, True, out=b, stream=s)),
    "localmax11": (None, lambda: gd.localmax(real, 11, out=b, stream=s)),
    "bestmax1001": (None, lambda: gd.best_extrema(real, 1001, True, out=b, stream=s)),
    "binarize": (cp, lambda: gd.binarize(a, 30.0, stream=s)),
    "dilate": (None, lambda: gd.dilate(depth, l, r, out=b, stream=s)),
    "dilate20001": (None, lambda: gd.dilate(depth, 10000, 10001, out=b, stream=s)),
    "smooth_hann": (None, lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann201": (None, lambda: gd.smooth(real, 201, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann501": (None, lambda: gd.smooth(real, 501, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann1001": (None, lambda: gd.smooth(real, 1001, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann2001": (None, lambda: gd.smooth(real, 2001, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann3201": (None, lambda: gd.smooth(real, 3201, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann4001": (None, lambda: gd.smooth(real, 4001, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann4003": (None, lambda: gd.smooth(real, 4003, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann5001": (None, lambda: gd.smooth(real, 5001, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann20001": (None, lambda: gd.smooth(real, 20001, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_hann50001": (None, lambda: gd.smooth(real, 50001, out=b, mode=gd.FIR_HANN, stream=s)),
    "smooth_exact": (None, lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_EXACT, stream=s)),
    "smooth_fma": (None, lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_FMA, stream=s)),
    "peaks_exact": (None, lambda: gd.smooth_local_extrema(real, 101, 11, True, 0.0, out=b, stream=s)),
    "peaks_exact_depth": (None, lambda: gd.smooth_local_extrema(depth, 101, 11, True, 0.0, out=b, stream=s)),
    "peaks_fma": (None, lambda: gd.smooth_local_extrema(real, 101, 11, True, 0.0, out=b, mode=gd.FIR_FMA, stream=s)),
}
for name in op.split(","):
    if name not in OPS and name.startswith("smooth_hann"):          # any window: smooth_hann<W>
        OPS[name] = (None, lambda W=int(name[len("smooth_hann"):]): gd.smooth(real, W, out=b, mode=gd.FIR_HANN, stream=s))
    prep, fn = OPS[name]
    best = 1e30
    for _ in range(reps):
        if prep:
            prep()
        gd.sync(s)
        e0, e1 = gd.Event(), gd.Event()
        burst = int(os.environ.get("BURST", "1"))        # launches back to back inside one timing (sustained clocks)
        e0.record(s)
        for _ in range(burst):
            fn()
        e1.record(s)
        best = min(best, e0.elapsed_ms(e1) / burst)
    print("%-18s %s %8.3f ms %7.1f Gbases/s %6.1f%% of 8 TB/s at 16 B/base" % (name, os.environ.get("TAG", ""), best, n / best / 1e6, 100 * 16 * n / best / 1e6 / 8000))
