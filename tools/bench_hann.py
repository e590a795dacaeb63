"""smooth W=101 in its three arithmetic modes on one chromosome-sized vector (device timing)."""
import sys

import numpy as np

sys.path.insert(0, ".")
import genodsp_amd as gd  # noqa: E402

n = int(float(sys.argv[1])) if len(sys.argv) > 1 else 248956422
S = gd.Stream()
real = gd.synth_coverage(20240611, 0, 0, n, 1)
out = real.like()
for name, mode in (("exact", gd.FIR_EXACT), ("fma", gd.FIR_FMA), ("hann", gd.FIR_HANN)):
    best = 1e30
    for _ in range(6):
        gd.sync(S.handle)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(S.handle)
        gd.smooth(real, 101, out=out, mode=mode, stream=S.handle)
        e1.record(S.handle)
        best = min(best, e0.elapsed_ms(e1))
    print("smooth W=101 %-5s %8.3f ms  %7.1f Gbases/s  %6.2f TB/s" % (name, best, n / best / 1e6, 16 * n / best / 1e9))
    sys.stdout.flush()
a = gd.smooth(real, 101, mode=gd.FIR_EXACT).numpy()
b = gd.smooth(real, 101, mode=gd.FIR_HANN).numpy()
x = real.numpy()
print("max |hann - exact| / max|x| = %.3g" % (np.abs(a - b).max() / np.abs(x).max()))
for W in (81, 201, 301, 1001, 2001):
    for name, mode in (("fma", gd.FIR_FMA), ("hann", gd.FIR_HANN)):
        best = 1e30
        for _ in range(3):
            gd.sync(S.handle)
            e0, e1 = gd.Event(), gd.Event()
            e0.record(S.handle)
            gd.smooth(real, W, out=out, mode=mode, stream=S.handle)
            e1.record(S.handle)
            best = min(best, e0.elapsed_ms(e1))
        print("smooth W=%-5d %-5s %8.3f ms  %7.1f Gbases/s  %6.2f TB/s" % (W, name, best, n / best / 1e6, 16 * n / best / 1e9))
        sys.stdout.flush()
