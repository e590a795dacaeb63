"""Adversarial inputs for `smooth --smooth=hann` (gdsp_hann.hip), shared by tests/test_hip_hann_adversarial.py and
tools/hann_adversarial.py (which writes the worst error/bound ratios to profiles/).

The bar is the north star's one rounding per floating-point operation, written as
    |hann - reference| <= W * 2^-52 * sum_k |w_k v_k|  (+ W * 2^-1074 where products are subnormal)
at every output, reference = the oracle's restatement of sum.c:632-663 (bit-identical to the reference)."""
import numpy as np

from oracle import cpu

TILE_OUT_W101 = 3984                 # outputs per tile of hann_blocks_kernel<101>


def bound(x, W):
    taps = cpu.hann_window(W)
    with np.errstate(all="ignore"):
        return W * 2.0 ** -52 * cpu.fir(np.abs(x), taps) + W * 2.0 ** -1074


def worst_ratio(got, want, x, W):
    """(max |got - want| / bound over outputs where both are finite, whether the others agree in kind).
    In kind: NaN where the reference has NaN, the same infinity where it has one -- except that a sum within the
    bound of DBL_MAX may cross into overflow by one rounding on either side."""
    b = bound(x, W)
    big = np.finfo(np.float64).max
    both = np.isfinite(want) & np.isfinite(got)
    with np.errstate(all="ignore"):
        ratio = np.abs(got[both] - want[both]) / b[both]
        g, w = got[~both], want[~both]
        same = (np.isnan(g) & np.isnan(w)) | (g == w)
        near = big * (1 - W * 2.0 ** -52)
        crossing = (np.isinf(g) & np.isfinite(w) & (np.sign(g) == np.sign(w)) & (np.abs(w) >= near)) | \
                   (np.isinf(w) & np.isfinite(g) & (np.sign(g) == np.sign(w)) & (np.abs(g) >= near))
    return (float(ratio.max()) if ratio.size else 0.0), bool(np.all(same | crossing))


def impulse_trains(W, n, spacing, shifts, seed):
    """Vectors holding unit-ish impulses `spacing` apart (> W + 16: no two under one window or one block sum),
    one vector per shift: over all shifts every base of [0, n) carries an impulse once when shifts = range(spacing),
    i.e. every tap offset x every phase within a 16-block x every position relative to a tile seam is met."""
    rng = np.random.default_rng(seed)
    for s in shifts:
        x = np.zeros(n)
        pos = np.arange(s, n, spacing)
        x[pos] = rng.choice([-1.0, 1.0], pos.size) * 10.0 ** rng.uniform(-5, 5, pos.size)
        yield s, x


def wide_dynamic_range(n, seed, lo_exp=-300, hi_exp=300):
    """alternating signs, magnitudes 1e-300 .. 1e+300 side by side inside every window"""
    rng = np.random.default_rng(seed)
    return (-1.0) ** np.arange(n) * 10.0 ** rng.uniform(lo_exp, hi_exp, n)


def nonfinite_cases(n, seed):
    """(name, x): a benign real signal with the values the documented range used to exclude"""
    rng = np.random.default_rng(seed)
    base = cpu.synth_coverage(20240611, 2, 0, n, 1) + rng.standard_normal(n)
    big = np.finfo(np.float64).max
    out = []
    x = base.copy(); x[n // 3] = np.inf; out.append(("one +inf", x))
    x = base.copy(); x[n // 3] = -np.inf; x[n // 3 + 2000] = np.inf; out.append(("-inf and +inf far apart", x))
    x = base.copy(); x[n // 2] = np.inf; x[n // 2 + 30] = -np.inf; out.append(("+inf and -inf under one window", x))
    x = base.copy(); x[n // 2 + 7] = np.nan; out.append(("one NaN", x))
    x = base.copy(); x[5000:5600] = big; out.append(("a stretch of DBL_MAX", x))
    x = base.copy(); x[4000::1500] = big / 64; out.append(("isolated values just past DBL_MAX/128", x))
    x = base.copy(); x[3984 * 2 - 60] = -big; out.append(("-DBL_MAX 60 bases left of a tile seam", x))
    x = base.copy(); x[0] = np.inf; x[n - 1] = np.nan; out.append(("inf first, NaN last", x))
    return out
