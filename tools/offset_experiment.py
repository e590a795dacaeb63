#!/usr/bin/env python3
"""Does the relative placement of the input and the output vector matter (both come 2 MiB-aligned from hipMalloc)?
smooth --smooth=hann and an in-place operator's worth of traffic with the output shifted by a few offsets."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402
n = 248956422
gd.set_device(0)
S = gd.Stream(); s = S.handle
real = gd.synth_coverage(20240611, 0, 0, n, 1, stream=s)
big = gd.DeviceBuffer(n * 8 + (64 << 20))
print("in %x  out base %x" % (real.buf.ptr + real.offset, big.ptr))
for off in (0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, (3 << 20) + 8192 + 256, 33 << 20):
    out = gd.DeviceVector(n, buf=big, offset=off)
    best = 1e30
    for _ in range(4):
        gd.sync(s)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(s)
        for _ in range(10):
            gd.smooth(real, 101, out=out, mode=gd.FIR_HANN, stream=s)
        e1.record(s)
        best = min(best, e0.elapsed_ms(e1) / 10)
    print("out offset %10d B: %7.3f ms  %6.1f Gbases/s  %5.1f%% of 8 TB/s" % (off, best, n / best / 1e6, 100 * 16 * n / best / 1e6 / 8000))
