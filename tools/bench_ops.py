#!/usr/bin/env python3
"""Per-operator throughput on one chromosome-sized vector (default chr1, 249 Mbp):
Gbases/s and algorithmic GB/s (SURVEY.md 8d bytes per base) for every hot-path kernel.
usage: python3 tools/bench_ops.py [n] [reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 248956422
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
gd.set_device(0)
SEED = 20240611
src = gd.synth_coverage(SEED, 0, 0, n, 0)        # integer depth
real = gd.synth_coverage(SEED, 0, 0, n, 1)
a = gd.DeviceVector(n)
b = gd.DeviceVector(n)
stream = gd.Stream()
S = stream.handle


def timeit(name, fn, bytes_per_base=16, prep=None):
    best = 1e30
    for _ in range(reps):
        if prep:
            prep()
        gd.sync(S)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(S)
        fn()
        e1.record(S)
        ms = e0.elapsed_ms(e1)
        best = min(best, ms)
    print("%-28s %8.3f ms  %7.1f Gbases/s  %7.1f GB/s (%d B/base)  %4.1f%% of 8 TB/s" % (
        name, best, n / best / 1e6, bytes_per_base * n / best / 1e6, bytes_per_base,
        100 * bytes_per_base * n / best / 1e6 / 8000))
    sys.stdout.flush()


def copy_into(dst, s):
    gd.call("gdsp_memcpy_d2d", dst.ptr, s.ptr, n * 8, gd._sp(S))


timeit("memcpy d2d (hipMemcpyAsync)", lambda: copy_into(a, src))
timeit("smooth W=101 fma", lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_FMA, stream=S))
timeit("smooth W=101 exact", lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_EXACT, stream=S))
timeit("smooth W=21 fma (generic)", lambda: gd.smooth(real, 21, out=b, mode=gd.FIR_FMA, stream=S))
timeit("smooth W=1001 fma (generic)", lambda: gd.smooth(real, 1001, out=b, mode=gd.FIR_FMA, stream=S))
timeit("localmax N=11", lambda: gd.localmax(real, 11, out=b, stream=S))
timeit("localmax N=3", lambda: gd.localmax(real, 3, out=b, stream=S))
timeit("localmax N=101", lambda: gd.localmax(real, 101, out=b, stream=S))
timeit("bestmax W=100", lambda: gd.best_extrema(real, 100, True, out=b, stream=S))
timeit("bestmax W=1001", lambda: gd.best_extrema(real, 1001, True, out=b, stream=S))
l, r = gd.split_length(1001)
timeit("dilate 1001", lambda: gd.dilate(src, l, r, out=b, stream=S))
timeit("erode 1001", lambda: gd.erode(src, l, r, out=b, stream=S))
timeit("close 1001", lambda: gd.close(src, 1001, out=b, stream=S))
timeit("open 1001", lambda: gd.open_(src, 1001, out=b, stream=S))
timeit("binarize", lambda: gd.binarize(a, 10.0, stream=S), prep=lambda: copy_into(a, src))
timeit("clip", lambda: gd.clip(a, 1.0, 20.0, stream=S), prep=lambda: copy_into(a, src))
timeit("addconst", lambda: gd.add_constant(a, 1.5, stream=S), prep=lambda: copy_into(a, src))
timeit("slidingsum W=101", lambda: gd.sliding_sum(src, 101, out=b, stream=S))
timeit("sum W=100", lambda: gd.window_sum(a, 100, stream=S), prep=lambda: copy_into(a, src))
timeit("sum W=101", lambda: gd.window_sum(a, 101, stream=S), prep=lambda: copy_into(a, src))
timeit("sum W=1000", lambda: gd.window_sum(a, 1000, stream=S), prep=lambda: copy_into(a, src))
work = gd.DeviceBuffer(gd.lib().gdsp_cumulative_sum_work(n))
timeit("cumulativesum", lambda: gd.call("gdsp_cumulative_sum", a.ptr, n, gd.C.c_void_p(work.ptr), gd._sp(S)),
       prep=lambda: copy_into(a, src))
cwork = gd.DeviceBuffer(gd.lib().gdsp_clump_work(n))
timeit("clump T=30.5 L=1000", lambda: gd.call("gdsp_clump", a.ptr, n, 30.5, 1000, 1, 1.0, 0.0, gd.C.c_void_p(cwork.ptr), gd._sp(S)),
       prep=lambda: copy_into(a, src))
del cwork
hist = gd.DeviceBuffer((8192 + 2) * 8)
import ctypes as C  # noqa: E402
timeit("select histogram pass (depth)", lambda: gd.call("gdsp_select_histogram", src.ptr, n, 1, -gd.DBL_MAX, gd.DBL_MAX, 52, 12,
                                                 C.c_uint64(0), C.c_void_p(hist.ptr), gd._sp(S)), 8)
timeit("select histogram pass (real)", lambda: gd.call("gdsp_select_histogram", real.ptr, n, 1, -gd.DBL_MAX, gd.DBL_MAX, 39, 13,
                                                C.c_uint64(0x4030000000000000), C.c_void_p(hist.ptr), gd._sp(S)), 8)
acc = gd.DeviceBuffer(32)
timeit("minmax reduce", lambda: gd.call("gdsp_minmax_update", real.ptr, n, 1, -gd.DBL_MAX, gd.DBL_MAX, C.c_void_p(acc.ptr), gd._sp(S)), 8)
wk = gd.DeviceBuffer(gd.lib().gdsp_report_runs_work(n))
cnt = gd.DeviceBuffer(16)
timeit("report runs (count pass)", lambda: gd.call("gdsp_report_runs", src.ptr, n, 1, 0, None, None, None, 0, C.c_void_p(cnt.ptr),
                                            C.c_void_p(wk.ptr), gd._sp(S)), 8)
# PCIe: pinned host <-> HBM, 1 GiB (what the boundary would cost if it handed over host buffers)
hp = C.c_void_p()
gd.call("gdsp_host_alloc", C.byref(hp), 1 << 30)
for name, fn in (("h2d pinned 1 GiB", lambda: gd.call("gdsp_memcpy_h2d", a.ptr, hp, 1 << 30, gd._sp(S))),
                 ("d2h pinned 1 GiB", lambda: gd.call("gdsp_memcpy_d2h", hp, a.ptr, 1 << 30, gd._sp(S)))):
    best = 1e30
    for _ in range(3):
        gd.sync(S)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(S)
        fn()
        e1.record(S)
        best = min(best, e0.elapsed_ms(e1))
    print("%-28s %8.3f ms  %7.1f GB/s" % (name, best, (1 << 30) / best / 1e6))
gd.percentile([src], [50000])      # first call loads the code object
t0 = time.perf_counter()
cntv, vals = gd.percentile([src], [99000])
print("percentile 99 end-to-end (depth): %.2f ms -> %s (n=%d)" % ((time.perf_counter() - t0) * 1e3, vals, cntv))
t0 = time.perf_counter()
cntv, vals = gd.percentile([real], [99000])
print("percentile 99 end-to-end (real):  %.2f ms -> %s (n=%d)" % ((time.perf_counter() - t0) * 1e3, vals, cntv))
