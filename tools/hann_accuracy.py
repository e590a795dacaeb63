"""Error of smooth W=101 against an extended-precision evaluation, per arithmetic mode (GPU) and for
the reference's own loop (oracle); errors in units of 2^-53 * sum|w_k v_k| at each output."""
import sys

import numpy as np

sys.path.insert(0, ".")
import genodsp_amd as gd  # noqa: E402
from oracle import cpu  # noqa: E402

W, n = 101, 400000
taps = cpu.hann_window(W)
rng = np.random.default_rng(1)
for kind in ("depth", "real", "noise", "islands"):
    if kind == "noise":
        x = rng.standard_normal(n) * 5
    elif kind == "islands":
        x = cpu.synth_coverage(20240611, 3, 0, n, 1)
        x[(np.arange(n) // 700) % 3 != 0] = 0.0
    else:
        x = cpu.synth_coverage(20240611, 3, 0, n, 0 if kind == "depth" else 1)
    sel = rng.integers(0, n, 4000)
    xl = np.concatenate([np.zeros(W // 2), x, np.zeros(W // 2)]).astype(np.longdouble)
    tl = taps.astype(np.longdouble)
    truth = np.array([np.dot(tl, xl[i:i + W]) for i in sel])
    unit = cpu.fir(np.abs(x), taps)[sel] * 2.0 ** -53
    ok = unit > 0
    ref = cpu.smooth(x, W)[sel]
    d = gd.DeviceVector.from_numpy(x)
    line = "%-8s reference max %.2f rms %.2f" % (kind, np.max(np.abs(ref - truth)[ok] / unit[ok]),
                                                 np.sqrt(np.mean((np.abs(ref - truth)[ok] / unit[ok]) ** 2)))
    for name, mode in (("fma", gd.FIR_FMA), ("hann", gd.FIR_HANN)):
        got = gd.smooth(d, W, mode=mode).numpy()[sel]
        e = (np.abs(got - truth)[ok] / unit[ok]).astype(np.float64)
        line += " | %s max %.2f rms %.2f" % (name, e.max(), np.sqrt(np.mean(e ** 2)))
        assert np.all(got[~ok] == 0.0)
    print(line)
