import sys
sys.path.insert(0, ".")
import genodsp_amd as gd
n = 248956422
S = gd.Stream()
real = gd.synth_coverage(20240611, 0, 0, n, 1)
out = real.like()
import os
for W in [int(w) for w in os.environ.get("WINDOWS", "101,501,1001,1501,2001,3001,5001,20001,50001").split(",")]:
    best = 1e30
    for _ in range(4):
        gd.sync(S.handle)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(S.handle)
        gd.smooth(real, W, out=out, mode=gd.FIR_HANN, stream=S.handle)
        e1.record(S.handle)
        best = min(best, e0.elapsed_ms(e1))
    print("smooth W=%-6d hann %8.3f ms  %7.1f Gbases/s  %5.2f of 8 TB/s at 16 B/base" % (W, best, n / best / 1e6, 16 * n / best / 1e9 / 8))
    sys.stdout.flush()
