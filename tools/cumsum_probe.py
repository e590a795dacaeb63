import sys, os
sys.path.insert(0, os.getcwd())
import ctypes as C
import genodsp_amd as gd
gd.set_device(0)
S = gd.Stream(); s = S.handle
for n in (4<<20, 8<<20, 16<<20, 32<<20, 64<<20, 248956422):
    v = gd.synth_coverage(20240611, 0, 0, n, 0, stream=s)
    work = gd.DeviceBuffer(gd.lib().gdsp_cumulative_sum_work(n))
    best = 1e9
    for rep in range(6):
        gd.sync(s)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(s)
        for k in range(4):
            gd.call("gdsp_cumulative_sum", v.ptr, n, C.c_void_p(work.ptr), gd._sp(s))
        e1.record(s)
        best = min(best, e0.elapsed_ms(e1) / 4)
    print("n=%10d (%5.0f MB) %8.3f ms  %6.1f Gbases/s  %6.0f GB/s algorithmic" % (n, n*8/1e6, best, n/best/1e6, 16*n/best/1e6))
