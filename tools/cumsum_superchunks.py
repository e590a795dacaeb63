import os, sys, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd
gd.set_device(0)
S = gd.Stream(); s = S.handle
n = 248956422
depth = gd.synth_coverage(20240611, 0, 0, n, 0, stream=s)
a = gd.DeviceVector(n)
work = gd.DeviceBuffer(gd.lib().gdsp_cumulative_sum_work(n))
def run(chunk):
    best = 1e30
    for _ in range(4):
        gd.call("gdsp_memcpy_d2d", a.ptr, depth.ptr, n * 8, gd._sp(s)); gd.sync(s)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(s)
        off = 0
        while off < n:
            m = min(chunk, n - off)
            gd.call("gdsp_cumulative_sum", C.c_void_p(a.ptr.value + off * 8), m, C.c_void_p(work.ptr), gd._sp(s))
            off += m
        e1.record(s)
        best = min(best, e0.elapsed_ms(e1))
    print("chunk %10d bases (%6.1f MB): %8.3f ms  %5.1f%% of 8 TB/s at 16 B/base" % (chunk, chunk * 8 / 1e6, best, 100 * 16 * n / best / 1e6 / 8000))
for chunk in (n, 1 << 25, 1 << 24, 3 << 22, 1 << 23, 1 << 22, 1 << 21):
    run(chunk)
