#!/usr/bin/env python3
"""cumulativesum (and clump) on one chromosome-sized vector: ms per call, Gbases/s, fraction of 8 TB/s on 16 B/base.
usage: [GDSP_CUMSUM=3] python3 tools/bench_scan.py [n] [reps]     (GDSP_CUMSUM=3: the three-launch form of rounds 1-4)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 248956422
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 7
gd.set_device(0)
stream = gd.Stream()
S = stream.handle
C = gd.C
work = gd.DeviceBuffer(gd.lib().gdsp_cumulative_sum_work(n))
cwork = gd.DeviceBuffer(gd.lib().gdsp_clump_work(n))
a = gd.DeviceVector(n)
for kind, mode in (("read depth", 0), ("real-valued", 1)):
    src = gd.synth_coverage(20240611, 0, 0, n, mode)
    for name, fn in (("cumulativesum", lambda: gd.call("gdsp_cumulative_sum", a.ptr, n, C.c_void_p(work.ptr), gd._sp(S))),
                     ("clump T=30.5 L=1000", lambda: gd.call("gdsp_clump", a.ptr, n, 30.5, 1000, 1, 1.0, 0.0, C.c_void_p(cwork.ptr), gd._sp(S)))):
        best, times = 1e30, []
        for _ in range(reps):
            gd.call("gdsp_memcpy_d2d", a.ptr, src.ptr, n * 8, gd._sp(S))
            gd.sync(S)
            e0, e1 = gd.Event(), gd.Event()
            e0.record(S)
            fn()
            e1.record(S)
            gd.sync(S)
            ms = e0.elapsed_ms(e1)
            times.append(ms)
            best = min(best, ms)
        print("%-22s %-12s GDSP_CUMSUM=%s  best %7.3f ms  median %7.3f ms  %6.1f Gbases/s  %.3f of 8 TB/s on 16 B/base" % (
            name, kind, os.environ.get("GDSP_CUMSUM", "1"), best, sorted(times)[len(times) // 2], n / best / 1e6, 16 * n / best / 1e6 / 8000))
        sys.stdout.flush()
