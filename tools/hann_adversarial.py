#!/usr/bin/env python3
"""Worst |hann - reference| / bound over the adversarial inputs of tests/hann_cases.py, on the GPU of this box.
    python tools/hann_adversarial.py > gpurun_out/hann_adversarial.txt      (copied to profiles/ per round)"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import genodsp_amd as gd            # noqa: E402
import hann_cases as hc             # noqa: E402
from oracle import cpu              # noqa: E402  (the checker)


def hann(x, W):
    return gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_HANN).numpy()


print("library:", gd.lib().gdsp_version().decode())
print("bound = W * 2^-52 * sum|w_k v_k| (+ W * 2^-1074); ratio = max over outputs of |hann - reference| / bound")
W, n, spacing = 101, 10 * hc.TILE_OUT_W101 + 77, 119
worst, at = 0.0, None
for s, x in hc.impulse_trains(W, n, spacing, range(spacing), seed=1):
    r, k = hc.worst_ratio(hann(x, W), cpu.smooth(x, W), x, W)
    assert k
    if r > worst:
        worst, at = r, s
print("impulse at every base of 10 tiles, W=101 (101 taps x 16 block phases x both sides of every seam): %.4f (shift %s)" % (worst, at))
for W in (81, 201, 427, 1001, 2001):
    spacing = W + 18 + (W + 18) % 2 + 1
    worst = 0.0
    for s, x in hc.impulse_trains(W, 3 * 4096 + 55, spacing, range(0, spacing, 5), seed=W):
        r, k = hc.worst_ratio(hann(x, W), cpu.smooth(x, W), x, W)
        assert k
        worst = max(worst, r)
    print("impulse trains, W=%d: %.4f" % (W, worst))
for W in (101, 301, 1001):
    worst = 0.0
    for seed in range(3):
        x = hc.wide_dynamic_range(30011, seed)
        r, k = hc.worst_ratio(hann(x, W), cpu.smooth(x, W), x, W)
        assert k
        worst = max(worst, r)
    print("1e-300..1e+300 with alternating signs inside every window, W=%d: %.4f" % (W, worst))
x = hc.wide_dynamic_range(20000, 9, -323, -300)
for W in (101, 201):
    r, k = hc.worst_ratio(hann(x, W), cpu.smooth(x, W), x, W)
    print("subnormal inputs (1e-323..1e-300), W=%d: %.4f" % (W, r))
for W in (101, 201):
    for name, x in hc.nonfinite_cases(5 * hc.TILE_OUT_W101 + 123, 4):
        got = hann(x, W)
        r, k = hc.worst_ratio(got, cpu.smooth(x, W), x, W)
        fma = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_FMA).numpy()
        with np.errstate(all="ignore"):
            touched = cpu.fir((~(np.abs(x) < 2.0 ** 1017)).astype(np.float64), np.ones(W)) > 0
        same = got[touched].tobytes() == fma[touched].tobytes()
        print("W=%d %-42s finite outputs %.4f, non-finite agree in kind: %s, outputs under them == fma bits: %s"
              % (W, name + ":", r, k, same))
