#!/usr/bin/env python3
"""Small driver for rocprofv3: a few smooth W=101 launches on one 145 Mbp chromosome.
usage: python3 tools/prof_smooth.py [fma|exact|hann] [launches]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402

mode = {"fma": gd.FIR_FMA, "exact": gd.FIR_EXACT, "hann": gd.FIR_HANN}[sys.argv[1] if len(sys.argv) > 1 else "fma"]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 145138636
gd.set_device(0)
vin = gd.synth_coverage(20240611, 7, 0, n, 1)
vout = gd.DeviceVector(n)
for _ in range(launches):
    gd.smooth(vin, 101, out=vout, mode=mode)
gd.sync()
print("done", mode, launches)
