#!/usr/bin/env python3
"""Do independent chromosomes on alternating streams hide the drain between kernels?  smooth --smooth=hann over the
24 chromosomes of the bench, 10 passes, on 1, 2, 3 and 4 streams (wall clock around device-wide syncs)."""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402
import bench  # noqa: E402
gd.set_device(0)
lengths = [n for _, n in bench.GENOME]
total = sum(lengths)
s0 = gd.Stream()
vin = [gd.synth_coverage(bench.SEED, c, 0, n, mode=1, stream=s0.handle) for c, n in enumerate(lengths)]
vout = [v.like() for v in vin]
gd.sync()
for op, fn in (("smooth hann", lambda i, s: gd.smooth(vin[i], 101, out=vout[i], mode=gd.FIR_HANN, stream=s)),
               ("binarize", lambda i, s: gd.binarize(vout[i], 30.0, stream=s))):
    for ns in (1, 2, 3, 4, 1, 2):
        streams = [gd.Stream() for _ in range(ns)]
        best = 1e30
        for _ in range(3):
            gd.sync()
            t0 = time.perf_counter()
            for rep in range(10):
                for i in range(len(vin)):
                    fn(i, streams[i % ns].handle)
            gd.sync()
            best = min(best, (time.perf_counter() - t0) / 10)
        print("%-12s %d stream(s): %7.3f ms per genome  %6.1f Gbases/s  %5.1f%% of 8 TB/s" % (op, ns, best * 1e3, total / best / 1e9, 100 * 16 * total / best / 8e12))
