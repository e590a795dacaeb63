#!/usr/bin/env python3
"""Keeps GPU 0 busy until it is killed: `smooth W=101` (hann, then exact) launches over a 64 Mbp vector, back to back
(tools/flake_matrix.py runs it beside the command lines under test)."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402
gd.set_device(0)
S = gd.Stream()
n = 64 << 20
x = gd.synth_coverage(20240611, 0, 0, n, 1, stream=S.handle)
y = gd.DeviceVector(n)
while True:
    for _ in range(20):
        gd.smooth(x, 101, out=y, mode=gd.FIR_HANN, stream=S.handle)
        gd.smooth(x, 101, out=y, mode=gd.FIR_EXACT, stream=S.handle)
    gd.sync(S.handle)
