"""GPU: `smooth --smooth=hann` (block sums, gdsp_hann.hip) on inputs chosen to break it.

 * unit impulses at every base of ten tiles (so: each of the 101 tap offsets x each of the 16 phases inside a block x
   every position either side of a 3984-output tile seam), compared tap by tap with the reference's window;
 * 1e-300 .. 1e+300 side by side with alternating signs inside every window; subnormal inputs;
 * DBL_MAX (what `localmin` leaves, minmax.c:901), +-inf, NaN: those tiles are evaluated tap by tap
   (hann_direct_tile), bit-identical to --smooth=fma, and agree with the reference in kind.
Bound: tests/hann_cases.py.  Worst ratios measured on the MI355X: profiles/r02_hann_adversarial.txt."""
import numpy as np
import pytest

import hann_cases as hc
from conftest import bits_equal
from oracle import cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gd():
    import genodsp_amd
    assert genodsp_amd.device_count() >= 1
    return genodsp_amd


def hann(gd, x, W):
    return gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_HANN).numpy()


def test_impulse_at_every_base_of_ten_tiles_w101(gd):
    W, n, spacing = 101, 10 * hc.TILE_OUT_W101 + 77, 119
    worst = 0.0
    for s, x in hc.impulse_trains(W, n, spacing, range(spacing), seed=1):
        r, kinds = hc.worst_ratio(hann(gd, x, W), cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (s, r)
        worst = max(worst, r)
    assert worst > 0.0            # (the comparison is not vacuous: block sums do differ from the reference's bits)


@pytest.mark.parametrize("W", [81, 201, 427, 1001, 1501, 1701, 1703, 2001, 4001])
def test_impulse_trains_runtime_windows(W, gd):
    spacing = W + 18 + (W + 18) % 2 + 1                       # odd, > W + 16
    n = 3 * 12288 + 55
    step = max(5, spacing // 160) | 1                         # odd: every phase inside a block of 16 comes up; <= 161 shifts
    for s, x in hc.impulse_trains(W, n, spacing, range(0, spacing, step), seed=W):
        r, kinds = hc.worst_ratio(hann(gd, x, W), cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (W, s, r)


@pytest.mark.parametrize("W", [101, 301, 1001, 3001])
def test_wide_dynamic_range_inside_every_window(W, gd):
    for seed in range(3):
        x = hc.wide_dynamic_range(30011, seed)
        r, kinds = hc.worst_ratio(hann(gd, x, W), cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (W, seed, r)


def test_subnormal_inputs(gd):
    x = hc.wide_dynamic_range(20000, 9, -323, -300)
    for W in (101, 201):
        r, kinds = hc.worst_ratio(hann(gd, x, W), cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (W, r)


@pytest.mark.parametrize("W", [101, 201, 1001])
def test_nonfinite_and_huge_inputs_follow_direct_evaluation(W, gd):
    n = 5 * hc.TILE_OUT_W101 + 123
    clean = None
    for name, x in hc.nonfinite_cases(n, 4):
        got = hann(gd, x, W)
        fma = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_FMA).numpy()
        want = cpu.smooth(x, W)
        # in kind like the reference (inf stays inf, inf-inf is NaN in both, NaN exactly where a window holds one) ...
        r, kinds = hc.worst_ratio(got, want, x, W)
        assert kinds, name
        assert r <= 1.0, (name, r)
        # ... and every output under a non-finite or huge input carries the bits of direct evaluation
        with np.errstate(all="ignore"):
            touched = cpu.fir((~(np.abs(x) < 2.0 ** 1017)).astype(np.float64), np.ones(W)) > 0
        assert touched.any()
        assert bits_equal(got[touched], fma[touched]), name


def test_localmin_then_smooth_hann(gd):
    """`= localmin N=11 = smooth` fills with DBL_MAX (minmax.c:901): wrong by design before this round."""
    x = cpu.synth_coverage(20240611, 1, 0, 40000, 1)
    lm = gd.localmin(gd.DeviceVector.from_numpy(x), 11)
    got = gd.smooth(lm, 101, mode=gd.FIR_HANN).numpy()
    y = cpu.local_extrema(x, 11, 0, np.finfo(np.float64).max)
    want = cpu.smooth(y, 101)
    r, kinds = hc.worst_ratio(got, want, y, 101)
    assert kinds and r <= 1.0, r


# ---- windows beyond one LDS tile: block totals in HBM (gdsp_hann_far.hip), 3201 .. 50001 taps

@pytest.mark.parametrize("W", [4003, 4005, 5001, 9999, 20001, 50001])
def test_far_windows_within_one_rounding_per_op(W, gd):
    rng = np.random.default_rng(W)
    for kind, n in (("real", 70001), ("noise", 3 * 3072 + 17), ("depth", 1), ("real", W // 2 + 3), ("islands", 140000)):
        if kind == "real":
            x = cpu.synth_coverage(20240611, 3, 0, n, 1)
        elif kind == "depth":
            x = cpu.synth_coverage(20240611, 3, 0, n, 0)
        elif kind == "noise":
            x = rng.standard_normal(n) * 5
        else:
            x = cpu.synth_coverage(20240611, 3, 0, n, 0)
            x[(np.arange(n) // 9000) % 3 == 1] = 0.0
        r, kinds = hc.worst_ratio(hann(gd, x, W), cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (W, kind, n, r)


@pytest.mark.parametrize("W", [4003, 20001])
def test_far_windows_impulse_trains(W, gd):
    spacing = W + 18 + (W + 18) % 2 + 1
    n = 8 * 3072 + 55 + 2 * W
    for s, x in hc.impulse_trains(W, n, spacing, range(0, spacing, max(1, spacing // 29) | 1), seed=W):
        r, kinds = hc.worst_ratio(hann(gd, x, W), cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (W, s, r)


def test_far_windows_wide_dynamic_range_and_nonfinite(gd):
    W = 5001
    for seed in range(2):
        x = hc.wide_dynamic_range(40011, seed)
        r, kinds = hc.worst_ratio(hann(gd, x, W), cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (seed, r)
    n = 9 * 3072 + 123
    for name, x in hc.nonfinite_cases(n, 4):
        got = hann(gd, x, W)
        fma = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_FMA).numpy()
        r, kinds = hc.worst_ratio(got, cpu.smooth(x, W), x, W)
        assert kinds and r <= 1.0, (name, r)
        with np.errstate(all="ignore"):
            touched = cpu.fir((~(np.abs(x) < 2.0 ** 1017)).astype(np.float64), np.ones(W)) > 0
        assert touched.any() and bits_equal(got[touched], fma[touched]), name
