"""GPU: the HIP path (through the C ABI) against the golden vectors and the CPU oracle.

Bars (BASELINE.json north_star): interval/index/predicate outputs bit-exact; smooth
bit-exact in EXACT mode and within one rounding per floating-point op in FMA mode;
running-sum operators bit-exact on exactly-summable signals (read depth) and within
the reference's own accumulated rounding on arbitrary reals (bound stated inline).
"""
import os

import numpy as np
import pytest

from backends import GpuBackend, OracleBackend
from conftest import bits_equal, first_diff, golden
from oracle import cpu
from pipeline import Runner

pytestmark = pytest.mark.gpu

EPS = 2.0 ** -53
SEED = 20240611
VECTOR_CASES = golden().vector_cases()
# running-sum ops on non-dyadic reals cannot be bit-reproduced by any parallel evaluation
TOLERANT = ("slidingsum_real", "cumsum_real", "sum_real_chrom")


@pytest.fixture(scope="module")
def gd():
    import genodsp_amd
    assert genodsp_amd.device_count() >= 1
    return genodsp_amd


def _tolerant(name):
    return any(name.startswith(t) for t in TOLERANT)


@pytest.mark.parametrize("case", VECTOR_CASES, ids=[c["name"] for c in VECTOR_CASES])
def test_hip_matches_reference_golden_vectors(case, gold, gd):
    chroms = [tuple(c) for c in case["chroms"]]
    r = Runner(GpuBackend(), chroms, gold.inputs(case), case.get("files")).run(case["pipeline"])
    if "percentile" in case["pipeline"] and "--preserve" not in case["pipeline"]:
        # non-destructive here (the reference scrambles the signal): inputs must be intact
        for c, _ in chroms:
            assert bits_equal(r.result(c), gold.inputs(case)[c])
    else:
        want = gold.outputs(case)
        for c, _ in chroms:
            got = r.result(c)
            if _tolerant(case["name"]):
                x = gold.inputs(case)[c]
                bound = 4 * x.size * EPS * np.abs(np.cumsum(np.abs(x))).max()
                assert np.abs(got - want[c]).max() <= bound
            else:
                assert bits_equal(got, want[c]), "%s %s first differing index %s" % (
                    case["name"], c, first_diff(got, want[c]))
    for name, hexval in case["globals"].items():
        assert float.fromhex(hexval) == r.globals[name], (name, float.fromhex(hexval), r.globals[name])


# ------------------------------------------------------------------ smooth ----

def _signal(kind, n, rng):
    if kind == "depth":
        return cpu.synth_coverage(SEED, 3, 0, n, 0)
    if kind == "real":
        return cpu.synth_coverage(SEED, 3, 0, n, 1)
    return rng.standard_normal(n) * 5


@pytest.mark.parametrize("n", [1, 2, 50, 51, 52, 101, 2303, 2304, 2305, 2404, 4608, 100003, 1000000])
@pytest.mark.parametrize("kind", ["depth", "real", "noise"])
def test_smooth_w101_exact_is_bit_identical(n, kind, gd):
    rng = np.random.default_rng(n)
    x = _signal(kind, n, rng)
    got = gd.smooth(gd.DeviceVector.from_numpy(x), 101, mode=gd.FIR_EXACT).numpy()
    want = cpu.smooth(x, 101)
    assert bits_equal(got, want), first_diff(got, want)


@pytest.mark.parametrize("W", [3, 5, 9, 11, 51, 99, 103, 301, 1001, 1027, 2053, 5001])
@pytest.mark.parametrize("n", [1, 7, 2304, 30011])
def test_smooth_generic_windows_exact(W, n, gd):
    rng = np.random.default_rng(W * 7 + n)
    x = _signal("real", n, rng)
    got = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_EXACT).numpy()
    want = cpu.smooth(x, W)
    assert bits_equal(got, want), first_diff(got, want)


def test_smooth_max_window_50001_exact(gd):
    """Largest window the reference accepts (sum.c:478): 49 LDS stages of taps."""
    rng = np.random.default_rng(5)
    x = _signal("real", 30011, rng)
    got = gd.smooth(gd.DeviceVector.from_numpy(x), 50001, mode=gd.FIR_EXACT).numpy()
    want = cpu.smooth(x, 50001)
    assert bits_equal(got, want), first_diff(got, want)


@pytest.mark.parametrize("W,n", [(101, 100003), (101, 2304), (21, 5000), (1001, 20000)])
@pytest.mark.parametrize("kind", ["depth", "real", "noise"])
def test_smooth_fma_within_one_rounding_per_op(W, n, kind, gd):
    """FMA mode fuses each tap's multiply and add.  Both it and the reference's unfused
    loop err by at most one rounding of the running sum per tap, so they differ by at
    most W * 2^-52 * sum_k |w_k v_k| at any output."""
    rng = np.random.default_rng(n + W)
    x = _signal(kind, n, rng)
    taps = cpu.hann_window(W)
    got = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_FMA).numpy()
    want = cpu.smooth(x, W)
    scale = cpu.fir(np.abs(x), taps)
    assert np.all(np.abs(got - want) <= W * 2 * EPS * scale)
    # and it is no further from an extended-precision evaluation than the reference is
    xl = np.concatenate([np.zeros(W // 2), x, np.zeros(W // 2)]).astype(np.longdouble)
    truth = np.array([np.dot(taps.astype(np.longdouble), xl[i:i + W]) for i in range(0, n, max(1, n // 400))])
    sel = np.arange(0, n, max(1, n // 400))
    assert np.abs(got[sel] - truth).max() <= np.abs(want[sel] - truth).max() * 1.5 + 1e-300


@pytest.mark.parametrize("n", [1, 2, 49, 50, 51, 101, 3983, 3984, 3985, 4096, 7968, 100003, 1000000])
@pytest.mark.parametrize("kind", ["depth", "real", "noise"])
def test_smooth_hann_block_sums_within_one_rounding_per_op(n, kind, gd):
    """HANN mode evaluates the same window through block sums (gdsp_hann.hip): a different association
    and exact cosines, so not the reference's bits -- the bar is the north star's one rounding per
    floating-point operation, W * 2^-52 * sum|w_k v_k| (as for FMA), and an error against an
    extended-precision evaluation like the reference's own.  3984 = outputs per tile."""
    W = 101
    rng = np.random.default_rng(n)
    x = _signal(kind, n, rng)
    taps = cpu.hann_window(W)
    got = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_HANN).numpy()
    want = cpu.smooth(x, W)
    scale = cpu.fir(np.abs(x), taps)
    assert np.all(np.abs(got - want) <= W * 2 * EPS * scale), first_diff(got, want)
    if n >= 100000:
        # against an extended-precision evaluation it errs like the reference's own loop does (measured,
        # tools/hann_accuracy.py: rms 2.3-2.9 vs 2.4-3.0, max 17-20 vs 9-12, in units of 2^-53 * sum|w_k v_k|)
        sel = np.arange(0, n, n // 4000)
        xl = np.concatenate([np.zeros(W // 2), x, np.zeros(W // 2)]).astype(np.longdouble)
        truth = np.array([np.dot(taps.astype(np.longdouble), xl[i:i + W]) for i in sel])
        unit = scale[sel] * EPS
        ok = unit > 0
        mine = (np.abs(got[sel] - truth)[ok] / unit[ok]).astype(np.float64)
        refs = (np.abs(want[sel] - truth)[ok] / unit[ok]).astype(np.float64)
        assert np.sqrt(np.mean(mine ** 2)) <= 1.5 * np.sqrt(np.mean(refs ** 2))
        assert mine.max() <= 3.0 * refs.max()
        assert np.all(got[sel][~ok] == 0.0)


@pytest.mark.parametrize("W", [81, 83, 85, 99, 103, 201, 301, 427, 501, 999, 1001, 1003, 1499, 1501, 1699, 1701, 1703, 1705, 1999, 2001, 2999, 3001, 3999, 4001])
def test_smooth_hann_any_window_within_one_rounding_per_op(W, gd):
    """Windows of 81..2001 taps go through the run-time form of the block-sum kernel (direct taps at the ends
    growing like sqrt(0.15 W)); same bar as W=101, sizes around its tile seams."""
    rng = np.random.default_rng(W)
    taps = cpu.hann_window(W)
    for kind, n in (("depth", 20011), ("real", 9000), ("noise", 4096), ("real", W // 2 + 3), ("depth", 1), ("real", 60013)):
        x = _signal(kind, n, rng)
        if kind == "depth":
            x[(np.arange(n) // 900) % 3 == 1] = 0.0          # islands: windows that only touch the small end taps
        got = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_HANN).numpy()
        want = cpu.smooth(x, W)
        scale = cpu.fir(np.abs(x), taps)
        assert np.all(np.abs(got - want) <= W * 2 * EPS * scale), (kind, n, first_diff(got, want),
                                                                   float(np.max(np.abs(got - want) / np.maximum(W * 2 * EPS * scale, 1e-300))))


@pytest.mark.parametrize("factor", [1e-300, 1e-150, 1e150, 1e290])
def test_smooth_hann_bound_is_relative_to_the_signal(factor, gd):
    """The same bar at very small and very large magnitudes (the block sums add up to 85 unweighted inputs:
    the documented range ends at DBL_MAX/128)."""
    rng = np.random.default_rng(5)
    for W in (101, 301):
        taps = cpu.hann_window(W)
        for kind in ("real", "noise"):
            x = _signal(kind, 30011, rng) * factor
            got = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_HANN).numpy()
            want = cpu.smooth(x, W)
            scale = cpu.fir(np.abs(x), taps)
            assert np.all(np.isfinite(got))
            assert np.all(np.abs(got - want) <= W * 2 * EPS * scale), (W, kind, first_diff(got, want))


def test_smooth_hann_mode_other_windows_fall_back_to_fma(gd):
    x = _signal("real", 30000, np.random.default_rng(3))
    for W in (21, 79):
        a = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_HANN).numpy()
        b = gd.smooth(gd.DeviceVector.from_numpy(x), W, mode=gd.FIR_FMA).numpy()
        assert bits_equal(a, b)


def test_fir_plan_custom_taps(gd):
    rng = np.random.default_rng(11)
    x = rng.standard_normal(9000)
    taps = rng.standard_normal(101)
    plan = gd.FirPlan(taps)
    got = plan.apply(gd.DeviceVector.from_numpy(x)).numpy()
    plan.close()
    assert bits_equal(got, cpu.fir(x, taps))
    taps7 = rng.standard_normal(7)
    plan = gd.FirPlan(taps7)
    got = plan.apply(gd.DeviceVector.from_numpy(x)).numpy()
    plan.close()
    assert bits_equal(got, cpu.fir(x, taps7))


def test_smooth_rejects_bad_arguments(gd):
    v = gd.DeviceVector.from_numpy(np.ones(100))
    with pytest.raises(gd.GdspError):
        gd.smooth(v, 100)                    # even
    with pytest.raises(gd.GdspError):
        gd.smooth(v, 50003)                  # beyond sum.c:478
    with pytest.raises(gd.GdspError):
        gd.smooth(v, 101, out=v)             # aliasing


# -------------------------------------------------- extrema, morphology ----

@pytest.mark.parametrize("n", [1, 2, 3, 4095, 4096, 4097, 70001])
@pytest.mark.parametrize("N", [3, 5, 11, 31, 33, 101, 999])
def test_local_extrema_bit_exact(n, N, gd):
    rng = np.random.default_rng(n * 13 + N)
    for kind in ("depth", "noise"):
        x = _signal(kind, n, rng)
        d = gd.DeviceVector.from_numpy(x)
        assert bits_equal(gd.localmax(d, N).numpy(), cpu.local_extrema(x, N, 1, 0.0))
        assert bits_equal(gd.localmin(d, N).numpy(), cpu.local_extrema(x, N, 0, cpu.DBL_MAX))


@pytest.mark.parametrize("n", [1, 2, 5, 4096, 4097, 50021])
@pytest.mark.parametrize("W", [3, 4, 10, 32, 33, 100, 1001, 4000])
def test_best_extrema_bit_exact(n, W, gd):
    rng = np.random.default_rng(n * 17 + W)
    for kind in ("depth", "noise"):
        x = _signal(kind, n, rng)
        d = gd.DeviceVector.from_numpy(x)
        assert bits_equal(gd.best_extrema(d, W, True).numpy(), cpu.best_extrema(x, W, 1))
        assert bits_equal(gd.best_extrema(d, W, False).numpy(), cpu.best_extrema(x, W, 0))


def _islands(n, rng, max_gap=60, max_run=40):
    v = np.zeros(n)
    pos = 0
    while pos < n:
        gap = int(rng.integers(1, max_gap))
        run = int(rng.integers(1, max_run))
        v[pos + gap:pos + gap + run] = float(rng.integers(1, 9))
        pos += gap + run
    return v


@pytest.mark.parametrize("n", [1, 2, 127, 128, 129, 16384, 16385, 100000, 300007])
@pytest.mark.parametrize("L", [1, 2, 3, 10, 63, 64, 65, 1001, 5000])
def test_morphology_bit_exact(n, L, gd):
    rng = np.random.default_rng(n + L)
    x = _islands(n, rng, max_gap=3 * L + 5, max_run=3 * L + 5)
    if n > 3:
        x[rng.integers(0, n, 3)] = 0.5                     # values either side of the threshold
    d = gd.DeviceVector.from_numpy(x)
    left, right = gd.split_length(L)
    for T in (0.0, 0.75):
        assert bits_equal(gd.dilate(d, left, right, T).numpy(), cpu.dilate(x, left, right, T)), ("dilate", T)
        assert bits_equal(gd.erode(d, left, right, T).numpy(), cpu.erode(x, left, right, T)), ("erode", T)
        assert bits_equal(gd.close(d, L, T).numpy(), cpu.close(x, L, T)), ("close", T)
        assert bits_equal(gd.open_(d, L, T).numpy(), cpu.open_(x, L, T)), ("open", T)
    assert bits_equal(gd.dilate(d, 0, L, 0.0, 7.0, -1.0).numpy(), cpu.dilate(x, 0, L, 0.0, 7.0, -1.0))
    assert bits_equal(gd.erode(d, L, 0, 0.0, 7.0, -1.0).numpy(), cpu.erode(x, L, 0, 0.0, 7.0, -1.0))


def test_morphology_nan_membership_follows_the_reference_spelling(gd):
    """dilate asks `v > T` at position 0 and `!(v <= T)` elsewhere, erode asks `v > T`: they differ for NaN only
    (morphology.c:930 vs :935, :1384).  Both the block form (reach 17..3584) and the bit-mask tile must agree
    with the oracle's restatement."""
    rng = np.random.default_rng(77)
    n = 20000
    x = _islands(n, rng, max_gap=300, max_run=200)
    x[0] = np.nan
    x[rng.integers(1, n, 40)] = np.nan
    d = gd.DeviceVector.from_numpy(x)
    for left, right in ((5, 5), (50, 50), (0, 300), (700, 1), (2000, 2000)):
        assert bits_equal(gd.dilate(d, left, right, 0.0).numpy(), cpu.dilate(x, left, right, 0.0)), ("dilate", left, right)
        assert bits_equal(gd.erode(d, left, right, 0.0).numpy(), cpu.erode(x, left, right, 0.0)), ("erode", left, right)


def test_morphology_all_set_and_all_clear(gd):
    for n in (1, 500, 40000):
        ones, zeros = np.ones(n), np.zeros(n)
        for x in (ones, zeros):
            d = gd.DeviceVector.from_numpy(x)
            assert bits_equal(gd.dilate(d, 5, 6).numpy(), cpu.dilate(x, 5, 6))
            assert bits_equal(gd.erode(d, 5, 6).numpy(), cpu.erode(x, 5, 6))
            assert bits_equal(gd.close(d, 10).numpy(), cpu.close(x, 10))
            assert bits_equal(gd.open_(d, 10).numpy(), cpu.open_(x, 10))


def test_morphology_fractional_and_huge_lengths(gd):
    rng = np.random.default_rng(3)
    x = _islands(50000, rng)
    d = gd.DeviceVector.from_numpy(x)
    for L in (0.0, 0.5, 7.5, 39.999, 1e9):
        assert bits_equal(gd.close(d, L).numpy(), cpu.close(x, L)), L
        assert bits_equal(gd.open_(d, L).numpy(), cpu.open_(x, L)), L


# -------------------------------------------------------------- pointwise ----

@pytest.mark.parametrize("n", [1, 2, 3, 1023, 1024, 1025, 262145, 2000001])
def test_pointwise_bit_exact(n, gd):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) * 4
    x[::7] = 0.0
    if n > 10:
        x[3] = -0.0
        x[5] = np.nan
        x[8] = np.inf
    up = lambda: gd.DeviceVector.from_numpy(x)
    assert bits_equal(gd.binarize(up(), 0.5).numpy(), cpu.binarize(x, 0.5))
    assert bits_equal(gd.binarize(up(), 0.0, True, 3.0, -3.0).numpy(), cpu.binarize(x, 0.0, True, 3.0, -3.0))
    assert bits_equal(gd.clip(up(), -1.0, 2.0).numpy(), cpu.clip(x, -1.0, 2.0))
    assert bits_equal(gd.clip(up(), -1.0, None).numpy(), cpu.clip(x, -1.0, None))
    assert bits_equal(gd.clip(up(), None, 2.0).numpy(), cpu.clip(x, None, 2.0))
    for lo, hi in ((-1.0, 2.0), (-1.0, None), (None, 2.0)):
        for inside in (False, True):
            assert bits_equal(gd.erase(up(), lo, hi, inside, 9.0).numpy(), cpu.erase(x, lo, hi, inside, 9.0))
    assert bits_equal(gd.add_constant(up(), 1.1).numpy(), cpu.add_constant(x, 1.1))
    assert bits_equal(gd.add_constant(up(), 0.0).numpy(), x)
    assert bits_equal(gd.abs_(up()).numpy(), cpu.abs_(x))
    got, want = gd.invert(up(), 0.3).numpy(), cpu.invert(x, 0.3)
    nan = np.isnan(want)                                # the sign bit of a propagated NaN is not pinned
    assert np.array_equal(np.isnan(got), nan) and bits_equal(got[~nan], want[~nan])
    assert bits_equal(gd.fill(up(), 2.5).numpy(), np.full(n, 2.5))


def test_genome_minmax_and_invert_default(gd):
    rng = np.random.default_rng(8)
    vecs = [rng.standard_normal(n) * 3 for n in (5000, 70001, 33)]
    dv = [gd.DeviceVector.from_numpy(v) for v in vecs]
    lo, hi, cnt = gd.genome_minmax(dv)
    assert (lo, hi) == cpu.genome_minmax(vecs)
    assert cnt == sum(v.size for v in vecs)


# ------------------------------------------------------------------- sums ----

@pytest.mark.parametrize("n", [1, 2, 100, 4096, 4097, 123457])
@pytest.mark.parametrize("W", [3, 4, 100, 101, 1000, 4001])
def test_sliding_sum(n, W, gd):
    rng = np.random.default_rng(n + W)
    x = _signal("depth", n, rng)
    got = gd.sliding_sum(gd.DeviceVector.from_numpy(x), W).numpy()
    assert bits_equal(got, cpu.sliding_sum(x, W))                       # exact sums: bit-identical
    got = gd.sliding_sum(gd.DeviceVector.from_numpy(x), W, denom=float(W)).numpy()
    assert bits_equal(got, cpu.sliding_sum(x, W, float(W)))
    y = _signal("real", n, rng)
    got = gd.sliding_sum(gd.DeviceVector.from_numpy(y), W).numpy()
    want = cpu.sliding_sum(y, W)
    # reference: one accumulator, 2 roundings per step along the whole vector; ours: a tile prefix
    bound = EPS * np.abs(y).max() * (2.0 * (n + W) * W + 2.0 * (4096 + W + 2) ** 2)
    assert np.abs(got - want).max() <= bound


@pytest.mark.parametrize("W", [16, 17, 18, 33, 100, 101, 144, 145, 160, 161, 1000, 2047, 2048, 2049])
def test_sliding_sum_block_form_seams(W, gd):
    """Windows of 17..2048 bases run in the block form (sliding_blocks_kernel): (256 - (W-1)//16 - 1) * 16
    outputs per tile, 2 fewer when the alignment shift is needed; up to 8 whole blocks are added one by one,
    more come from a running sum over the tile's block totals (W >= 161)."""
    rng = np.random.default_rng(W)
    outs = (256 - ((W - 1) // 16 + 1)) * 16 if 17 <= W <= 2048 else 4096
    for n in sorted({1, W - 1, W, W + 1, outs - 2, outs - 1, outs, outs + 1, 2 * outs - 3, 3 * outs + 5, 40009}):
        x = _signal("depth", n, rng)
        got = gd.sliding_sum(gd.DeviceVector.from_numpy(x), W, denom=3.0).numpy()
        want = cpu.sliding_sum(x, W, 3.0)
        assert bits_equal(got, want), (n, first_diff(got, want))


@pytest.mark.parametrize("n", [1, 5, 100, 2350, 100003])
@pytest.mark.parametrize("W", [3, 7, 64, 100, 8192, 8193, 50000])
def test_window_sum(n, W, gd):
    rng = np.random.default_rng(n * 3 + W)
    for kind in ("depth", "real"):
        x = _signal(kind, n, rng)
        got = gd.window_sum(gd.DeviceVector.from_numpy(x), W, 1.0, False, 0.0).numpy()
        want = cpu.window_sum(x, min(W, n))
        if kind == "depth" or W <= 8192:
            assert bits_equal(got, want), (kind, first_diff(got, want))
        else:
            assert np.allclose(got, want, rtol=1e-12, atol=0)
        got = gd.window_sum(gd.DeviceVector.from_numpy(x), W, 1.0, True, -1.0).numpy()
        want = cpu.window_sum(x, min(W, n), use_actual=True, zero=-1.0)
        if kind == "depth" or W <= 8192:
            assert bits_equal(got, want)


@pytest.mark.parametrize("W", [257, 300, 500, 1000, 1023, 1024, 1025, 1500, 2000, 2001, 4097, 8191, 8192])
def test_window_sum_exactly_summable_windows(W, gd):
    """Windows of 257..8192 bases are added lane-parallel when every base in them is a multiple of 2^-20 below
    2^19 (no partial sum rounds, so the order cannot matter) and in the reference's order otherwise
    (gdsp_sums.hip: inside the tile kernel up to 1024 bases, as a first pass of one wave per window above).  The signals here put both kinds of window, the boundary values of the test, signed zeros
    and a ragged last window under the bit-for-bit comparison."""
    rng = np.random.default_rng(W)
    n = 40 * W + W // 3
    base = rng.poisson(30, n).astype(np.float64)
    cases = {"depth": base.copy()}
    x = base.copy(); x[rng.integers(0, n, 25)] += rng.random(25)              # some windows hold a non-dyadic base
    cases["mixed"] = x
    x = base * 2.0 ** -20; x[::7] *= -1.0                                      # the finest resolution the fast path takes
    cases["fine"] = x
    x = base.copy(); x[5 * W + 3] = 2.0 ** -21; x[9 * W] = 524288.0; x[11 * W + 1] = 524287.0 + 2.0 ** -20
    x[13 * W + 2] = -524288.0; x[15 * W + 4] = 2.0 ** 60; x[17 * W + 5] = np.inf; x[19 * W + 6] = np.nan
    cases["edges"] = x
    x = np.zeros(n); x[: 3 * W] = -0.0; x[4 * W + 1] = -0.0; x[6 * W: 7 * W] = -0.0; x[6 * W + 9] = 0.0
    x[8 * W] = 3.0; x[8 * W + 1] = -3.0
    cases["zeros"] = x
    cases["real"] = _signal("real", n, rng)
    for name, x in cases.items():
        for denom, actual, zero in ((1.0, False, 0.0), (float(W), False, 0.0), (1.0, True, -1.0)):
            got = gd.window_sum(gd.DeviceVector.from_numpy(x), W, denom, actual, zero).numpy()
            want = cpu.window_sum(x, W, denom=denom, use_actual=actual, zero=zero)
            assert bits_equal(got, want), (name, denom, actual, first_diff(got, want))


@pytest.mark.parametrize("n", [1, 2, 8191, 8192, 8193, 1000003])
def test_cumulative_sum(n, gd):
    rng = np.random.default_rng(n)
    x = _signal("depth", n, rng)
    assert bits_equal(gd.cumulative_sum(gd.DeviceVector.from_numpy(x)).numpy(), cpu.cumulative_sum(x))
    y = _signal("real", n, rng)
    got = gd.cumulative_sum(gd.DeviceVector.from_numpy(y)).numpy()
    want = cpu.cumulative_sum(y)
    assert np.abs(got - want).max() <= 2 * n * EPS * np.abs(want).max()


@pytest.mark.parametrize("W", [4, 5, 6, 7, 8, 9, 10, 11, 12, 15, 16, 17, 18, 32, 33, 48, 49, 50, 64, 65, 255, 256, 257, 2048, 2049, 3583, 3584, 3585])
def test_best_and_local_extrema_block_form_seams(W, gd):
    """Windows of 5..3584 bases run in the block form (extrema_blocks_kernel, blocks of 4, 8 or 16); its tile
    holds (256 - (W-1)//G - 1) * G outputs (2 fewer when the alignment shift is needed), so the vector lengths
    here put its seams, the vector ends and a ragged last tile under the comparison."""
    d = W - 1
    G = 16 if W >= 17 else (8 if W >= 9 else 4)
    outs = (256 - (d // G + 1)) * G if 5 <= W <= 3584 else 4096
    rng = np.random.default_rng(W)
    for n in sorted({1, W - 1, W, outs - 2, outs - 1, outs, outs + 1, 2 * outs - 3, 3 * outs + 7, 50021}):
        if n < 1:
            continue
        x = _signal("real", n, rng) if n % 2 else _signal("depth", n, rng)
        dv = gd.DeviceVector.from_numpy(x)
        for want_max in (True, False):
            got = gd.best_extrema(dv, W, want_max).numpy()
            assert bits_equal(got, cpu.best_extrema(x, W, 1 if want_max else 0)), (n, want_max, first_diff(
                got, cpu.best_extrema(x, W, 1 if want_max else 0)))
        if W % 2:
            got = gd.local_extrema(dv, W, True, -3.0).numpy()
            assert bits_equal(got, cpu.local_extrema(x, W, 1, -3.0)), n


# -------------------------------------------------------------- percentile ----

@pytest.mark.parametrize("kind", ["depth", "real", "noise"])
def test_percentile_exact_order_statistic(kind, gd):
    rng = np.random.default_rng(17)
    lens = [300007, 150001, 70000, 5]
    vecs = [_signal(kind, n, rng) if kind == "noise" else
            cpu.synth_coverage(SEED, i, 0, n, 0 if kind == "depth" else 1) for i, n in enumerate(lens)]
    dv = [gd.DeviceVector.from_numpy(v) for v in vecs]
    pts = [0, 1, 500, 25000, 50000, 90000, 99000, 99999, 100000]
    for window, lo, hi in ((1, -cpu.DBL_MAX, cpu.DBL_MAX), (1, 2.2250738585072014e-308, cpu.DBL_MAX),
                           (7, -1.0, 30.0)):
        wcnt, want = cpu.percentile(vecs, pts, window, lo, hi)
        # every route gives the reference's values: five radix passes per percentile, brackets from a
        # subsample (forced here: the population is far below the 2^24 where AUTO starts to sample; small
        # subsamples give wide brackets, 64 values give brackets that miss and fall back), the pass-level API
        for strategy, target in ((gd.SELECT_AUTO, 0), (gd.SELECT_RADIX, 0), (gd.SELECT_BRACKET, 1 << 15),
                                 (gd.SELECT_BRACKET, 4096), (gd.SELECT_BRACKET, 300)):
            cnt, got = gd.percentile(dv, pts, window, lo, hi, strategy=strategy, sample_target=target)
            assert cnt == wcnt, (strategy, target)
            assert list(got) == list(want), (window, lo, hi, strategy, target, gd.percentile_stats())
            stats = gd.percentile_stats()
            assert stats["population"] == wcnt
            if strategy == gd.SELECT_BRACKET and target >= 4096:
                assert stats["route"] == gd.SELECT_BRACKET and stats["population_passes"] <= 1 + 5 * stats["fallbacks"]
        cnt, got = gd.percentile_by_passes(dv, pts, window, lo, hi)
        assert cnt == wcnt and list(got) == list(want)
    for d, v in zip(dv, vecs):
        assert bits_equal(d.numpy(), v)                    # untouched


def test_percentile_brackets_on_awkward_populations(gd):
    rng = np.random.default_rng(23)
    n = 200000
    cases = {"constant": np.full(n, 3.25), "two values": np.where(rng.random(n) < 0.3, -1.0, 2.0),
             "sorted": np.sort(rng.standard_normal(n)), "sawtooth": (np.arange(n) % 509).astype(np.float64),
             "signed zeros": np.where(rng.random(n) < 0.5, 0.0, -0.0) * 1.0,
             "one outlier": np.concatenate([np.zeros(n - 1), [1e300]]),
             "infinities": np.where(rng.random(n) < 0.01, np.inf, np.where(rng.random(n) < 0.01, -np.inf,
                                                                           rng.standard_normal(n)))}
    pts = [0, 100, 30000, 50000, 70000, 99900, 100000]
    for name, x in cases.items():
        wcnt, want = cpu.percentile([x], pts, 1, -cpu.DBL_MAX, cpu.DBL_MAX)
        for target in (1 << 14, 1000):
            cnt, got = gd.percentile([gd.DeviceVector.from_numpy(x)], pts, strategy=gd.SELECT_BRACKET, sample_target=target)
            assert cnt == wcnt and list(got) == list(want), (name, target, gd.percentile_stats())
        for lo, hi in ((-0.5, 0.75), (2.2250738585072014e-308, cpu.DBL_MAX)):      # --min/--max: the bounded kernel
            wcnt, want = cpu.percentile([x], pts, 1, lo, hi)
            cnt, got = gd.percentile([gd.DeviceVector.from_numpy(x)], pts, 1, lo, hi, strategy=gd.SELECT_BRACKET,
                                     sample_target=1 << 14)
            assert cnt == wcnt and (cnt == 0 or list(got) == list(want)), (name, lo, hi, gd.percentile_stats())
    # NaNs pass the reference's filter; as keys the positive ones sort above every number, the negative ones
    # below.  The reference's qsort has no defined order for them, so the two routes are held to each other.
    x = rng.standard_normal(n)
    x[rng.integers(0, n, 300)] = np.nan
    x[rng.integers(0, n, 200)] = -np.nan
    x[rng.integers(0, n, 100)] = np.inf
    d = gd.DeviceVector.from_numpy(x)
    for lo, hi in ((-cpu.DBL_MAX, cpu.DBL_MAX), (-1.0, 1.5)):
        a = gd.percentile([d], [0, 10, 50000, 99990, 100000], 1, lo, hi, strategy=gd.SELECT_RADIX)
        b = gd.percentile([d], [0, 10, 50000, 99990, 100000], 1, lo, hi, strategy=gd.SELECT_BRACKET, sample_target=1 << 14)
        assert a[0] == b[0] and np.array_equal(np.array(a[1]), np.array(b[1]), equal_nan=True), (lo, hi, a, b)


def test_percentile_empty_sample(gd):
    dv = [gd.DeviceVector.from_numpy(np.zeros(1000))]
    cnt, got = gd.percentile(dv, [50000], 1, 1.0, 2.0)
    assert cnt == 0 and got == []


# --------------------------------------------------------------- intervals ----

def _random_intervals(n, count, rng, max_len=400, integer=True):
    s = rng.integers(0, n, count).astype(np.uint32)
    e = np.minimum(n, s + rng.integers(1, max_len, count)).astype(np.uint32)
    val = rng.integers(1, 5, count).astype(np.float64) if integer else rng.random(count) * 3 - 0.5
    return s, e, val


@pytest.mark.parametrize("n", [1, 1000, 1024, 1025, 250007])
def test_apply_intervals_file_order_bit_exact(n, gd):
    rng = np.random.default_rng(n)
    for integer in (True, False):
        s, e, val = _random_intervals(n, 50 + n // 20, rng, integer=integer)
        base = rng.random(n) if not integer else np.zeros(n)
        for op in (cpu.OVERLAP_SUM, cpu.OVERLAP_MIN, cpu.OVERLAP_MAX):
            for clear, missing in ((False, 0.0), (True, 0.0), (True, -1.0)):
                got = gd.apply_intervals(gd.DeviceVector.from_numpy(base), s, e, val, op, clear, missing).numpy()
                start = np.full(n, missing) if clear else base
                want = cpu.apply_intervals(start, s, e, val, op, clear, missing)
                assert bits_equal(got, want), (integer, op, clear, missing, first_diff(got, want))


def test_apply_intervals_empty(gd):
    base = np.arange(5000, dtype=np.float64)
    z = np.zeros(0, np.uint32)
    got = gd.apply_intervals(gd.DeviceVector.from_numpy(base), z, z, np.zeros(0), cpu.OVERLAP_SUM).numpy()
    assert bits_equal(got, base)
    got = gd.apply_intervals(gd.DeviceVector.from_numpy(base), z, z, np.zeros(0), cpu.OVERLAP_SUM, True, 3.0).numpy()
    assert bits_equal(got, np.full(5000, 3.0))


@pytest.mark.parametrize("n", [1, 1024, 3000, 100001])
def test_scale_intervals_bit_exact(n, gd):
    rng = np.random.default_rng(n + 1)
    cuts = np.unique(rng.integers(0, n + 1, 40))
    s, e = cuts[:-1:2].astype(np.uint32), cuts[1::2].astype(np.uint32)
    k = min(s.size, e.size)
    s, e = s[:k], e[:k]
    val = rng.random(k) * 4 + 0.25
    base = rng.standard_normal(n)
    base[::5] = 0.0
    for divide in (False, True):
        got = gd.scale_intervals(gd.DeviceVector.from_numpy(base), s, e, val, divide).numpy()
        assert bits_equal(got, cpu.scale_intervals(base, s, e, val, divide))


# ------------------------------------------------------------------ report ----

@pytest.mark.parametrize("n", [1, 2, 15, 16, 17, 4096, 4097, 100003])
def test_report_runs_bit_exact(n, gd):
    rng = np.random.default_rng(n)
    for kind in ("depth", "blocky"):
        x = cpu.synth_coverage(SEED, 1, 0, n, 0) if kind == "depth" else \
            np.repeat(rng.integers(0, 3, n // 3 + 1), 3)[:n].astype(np.float64)
        d = gd.DeviceVector.from_numpy(x)
        for collapse in (True, False):
            for uncovered in (0, 1, -1):
                gs, ge, gv = gd.report_runs(d, collapse, uncovered)
                ws, we, wv = cpu.report_runs(x, collapse, uncovered)
                assert np.array_equal(gs, ws) and np.array_equal(ge, we) and bits_equal(gv, wv), (
                    kind, collapse, uncovered)


# --------------------------------------------------------------- synthetic ----

@pytest.mark.parametrize("mode", [0, 1])
def test_synth_signal_matches_cpu_generator(mode, gd):
    for chrom, start, count in ((0, 0, 100000), (5, 123456789, 4097), (23, 4000000000 - 50, 50)):
        got = gd.synth_coverage(SEED, chrom, start, count, mode).numpy()
        assert bits_equal(got, cpu.synth_coverage(SEED, chrom, start, count, mode))


# ------------------------------------------------------------ fused chains ----

@pytest.mark.parametrize("n", [1, 2, 11, 2293, 2294, 2295, 2304, 3973, 3974, 3975, 3984, 4588, 100003])
@pytest.mark.parametrize("N", [1, 3, 4, 11, 41, 129])
def test_fused_smooth_localmax_is_the_two_operators(n, N, gd):
    rng = np.random.default_rng(n + N)
    for kind in ("depth", "real"):
        x = _signal(kind, n, rng)
        d = gd.DeviceVector.from_numpy(x)
        sm = cpu.smooth(x, 101)
        for want_max, fill in ((True, 0.0), (False, cpu.DBL_MAX), (True, -7.0)):
            got = gd.smooth_local_extrema(d, 101, N, want_max, fill).numpy()
            want = cpu.local_extrema(sm, N, 1 if want_max else 0, fill)
            assert bits_equal(got, want), (kind, want_max, first_diff(got, want))
        # fma mode: identical to the two fma-mode kernels run one after the other
        got = gd.smooth_local_extrema(d, 101, N, True, 0.0, mode=gd.FIR_FMA).numpy()
        two = gd.localmax(gd.smooth(d, 101, mode=gd.FIR_FMA), N).numpy()
        assert bits_equal(got, two)
        # hann mode is not shift invariant, so the fused kernel evaluates it as fma (ties stay ties)
        got = gd.smooth_local_extrema(d, 101, N, True, 0.0, mode=gd.FIR_HANN).numpy()
        assert bits_equal(got, two)
    assert gd.lib().gdsp_smooth_local_extrema_fusable(101, 11) == 1
    assert gd.lib().gdsp_smooth_local_extrema_fusable(21, 11) == 0
    assert gd.lib().gdsp_smooth_local_extrema_fusable(101, 131) == 0


def _filter_cases(n, rng):
    """Signals for the filtered route of the fused kernel (gdsp_peaks.hip): ties, near-ties, zero stretches, mixed signs,
    and the values that make a tile queue every base."""
    t = np.arange(n)
    cases = {}
    x = np.full(n, 30.0); x[n // 3: n // 2] = 31.0; x[n // 2: n // 2 + 7] = 0.0
    cases["flat steps"] = x                                       # long stretches of equal smoothed values
    x = _signal("depth", n, rng); x[(t // 700) % 3 == 1] = 0.0
    cases["zero islands"] = x
    cases["mixed signs"] = _signal("real", n, rng) - 40.0
    x = _signal("real", n, rng); half = n // 2; x[n - half:] = x[:half][::-1]
    cases["mirror image"] = x                                     # smoothed values either side of the centre agree to a rounding
    cases["period 7"] = (t % 7 == 0) * 5.0 + (t % 7 == 3) * 5.0
    x = np.zeros(n); x[::2] = -0.0; x[n // 4] = 2.0
    cases["signed zeros"] = x
    x = _signal("real", n, rng) * 1e-200; x[(t // 500) % 2 == 0] = 0.0
    cases["below 2^-500"] = x
    x = _signal("real", n, rng); x[n // 5] = 2.0 ** 1020; x[n // 2] = -2.0 ** 1018
    cases["huge"] = x
    x = _signal("depth", n, rng); x[n // 7] = np.inf; x[n // 3] = np.nan; x[n // 2] = -np.inf
    cases["nan and inf"] = x
    x = rng.standard_normal(n); x[np.abs(x) < 0.3] = 0.0
    cases["sparse noise"] = x
    x = _signal("real", n, rng); x[n // 3] = 2.0 ** 64; x[n // 3 + 200] = 2.0 ** 63.9; x[2 * n // 3] = 2.0 ** -64.5; x[2 * n // 3 + 300] = 2.0 ** -63.5
    cases["edges of the single-precision range"] = x
    cases["one part in 1e6 apart"] = 1000.0 + 1e-3 * np.sin(t / 40.0)   # smoothed values closer than single precision resolves
    # runs of equal inputs around the window's length (a base whose whole window lies in one run is written the run's value,
    # one chain of taps per run: the filter's form for flat stretches), steps of one in a deep signal, runs that return to
    # the value before, a run of zeros between equal runs
    x = np.empty(n); at = 0; k = 0
    lens = [60, 100, 101, 102, 103, 150, 99, 300, 101, 1000, 205, 16, 500]
    vals = [30.0, 31.0, 30.0, 29.0, 29.0, 57.0, 58.0, 0.0, 58.0, 3.0, 3.5, 3.0, 1e-3]
    while at < n:
        x[at: at + lens[k % len(lens)]] = vals[(k * 5) % len(vals)]
        at += lens[k % len(lens)]; k += 1
    cases["runs about a window long"] = x
    # the flat form takes a run's value from a table when it is a count below 256 and walks the chain of taps otherwise:
    # counts either side of the table's end, values between counts, counts past 2^31, all three in a tile
    x = np.empty(n); at = 0; k = 0
    lens = [130, 240, 101, 400, 180, 111, 350]
    vals = [255.0, 254.0, 256.0, 255.0, 257.0, 255.5, 1.0, 2.0 ** 31, 3.0, 2.0 ** 40, 12.0, 1e6, 0.5, 200.0, 4294967296.0 + 7.0]
    while at < n:
        x[at: at + lens[k % len(lens)]] = vals[(k * 4) % len(vals)]
        at += lens[k % len(lens)]; k += 1
    cases["runs of counts about the table's end"] = x
    return cases


PEAKS_ROUTES = ({}, {"GDSP_PEAKS_ROUTE": "filter"}, {"GDSP_PEAKS_ROUTE": "direct"}, {"GDSP_PEAKS_ROUTE": "filter", "GDSP_PEAKS_QUEUE_CAP": "7"},
                {"GDSP_PEAKS_FILTER": "0"}, {"GDSP_PEAKS_FLAT": "0"})


@pytest.mark.parametrize("n", [2291, 2292, 2293, 2294, 3961, 3962, 3963, 3971, 3972, 3973, 3974, 4584, 7944, 7945, 20011])
@pytest.mark.parametrize("N", [2, 3, 5, 11, 12, 15, 17, 41])
def test_filtered_smooth_extrema_is_bit_identical(n, N, gd):
    """`smooth W=101 = localmax|localmin N` evaluates tap by tap only the bases the block sums cannot rule out
    (gdsp_peaks.hip: filter tiles of 3984 - 2h - 2(h&1) outputs, undecided positions queued in HBM, 16 lanes per queued
    base in the exact kernel; neighbourhoods of 3..15 bases): the output is still that of the two reference loops run
    one after the other, on every kind of signal -- whichever route the probe picks, with the filter forced, with the
    direct kernel forced, with a queue of seven positions that overflows at once, and with the filter switched off."""
    rng = np.random.default_rng(n * 131 + N)
    for name, x in _filter_cases(n, rng).items():
        d = gd.DeviceVector.from_numpy(x)
        with np.errstate(all="ignore"):
            sm = cpu.smooth(x, 101)
        for want_max, fill in ((True, 0.0), (False, cpu.DBL_MAX)):
            want = cpu.local_extrema(sm, N, 1 if want_max else 0, fill)
            for env in PEAKS_ROUTES:
                os.environ.update(env)
                try:
                    got = gd.smooth_local_extrema(d, 101, N, want_max, fill).numpy()
                finally:
                    for key in env:
                        del os.environ[key]
                assert bits_equal(got, want), (name, want_max, env, first_diff(got, want))


def test_filtered_smooth_extrema_in_fma_arithmetic(gd):
    """--smooth=fma through the same filter (the default since round 4; GDSP_PEAKS_FILTER=exact keeps fma on the direct
    kernel): the exact evaluations then fuse each tap, and the output is that of the fma FIR followed by the neighbourhood
    test (the block sums are as close to fma's values as to the reference's)."""
    rng = np.random.default_rng(77)
    for n in (3973, 20011):
        for name, x in _filter_cases(n, rng).items():
            d = gd.DeviceVector.from_numpy(x)
            sm = gd.smooth(d, 101, mode=gd.FIR_FMA)
            for want_max, fill in ((True, 0.0), (False, cpu.DBL_MAX)):
                want = gd.local_extrema(sm, 11, want_max, fill).numpy()
                for env in ({"GDSP_PEAKS_ROUTE": "filter"}, {}, {"GDSP_PEAKS_FILTER": "exact"}, {"GDSP_PEAKS_FILTER": "0"}):
                    os.environ.update(env)
                    try:
                        got = gd.smooth_local_extrema(d, 101, 11, want_max, fill, mode=gd.FIR_FMA).numpy()
                    finally:
                        for key in env:
                            del os.environ[key]
                    assert bits_equal(got, want), (name, want_max, env, first_diff(got, want))


@pytest.mark.parametrize("n", [1, 2, 127, 129, 16384, 16385, 100000, 300007])
@pytest.mark.parametrize("L", [1, 2, 7, 64, 65, 1001, 5000])
def test_fused_dilate_erode_binarize_is_the_three_operators(n, L, gd):
    rng = np.random.default_rng(n * 3 + L)
    x = _islands(n, rng, max_gap=3 * L + 5, max_run=3 * L + 5)
    d = gd.DeviceVector.from_numpy(x)
    dl, dr = gd.split_length(L)
    for (el, er) in ((dl, dr), (3, 0), (0, 2 * L)):
        want = cpu.erode(cpu.dilate(x, dl, dr), el, er)
        assert bits_equal(gd.dilate_erode(d, dl, dr, el, er).numpy(), want), ("plain", el, er)
        wantb = cpu.binarize(want, 0.5, False, 9.0, -9.0)
        assert bits_equal(gd.dilate_erode(d, dl, dr, el, er, binarize=(0.5, False, 9.0, -9.0)).numpy(), wantb)
    # every stage with its own threshold and values, including ones that invert membership
    want = cpu.erode(cpu.dilate(x, dl, dr, 2.0, 5.0, -1.0), dl, dr, 0.0, 3.0, 4.0)
    got = gd.dilate_erode(d, dl, dr, dl, dr, d_T=2.0, d_one=5.0, d_zero=-1.0, e_T=0.0, e_one=3.0, e_zero=4.0).numpy()
    assert bits_equal(got, want)
    want = cpu.erode(cpu.dilate(x, dl, dr, 0.0, -5.0, 1.0), dl, dr)          # dilate's "one" is below erode's threshold
    got = gd.dilate_erode(d, dl, dr, dl, dr, d_one=-5.0, d_zero=1.0).numpy()
    assert bits_equal(got, want)


@pytest.mark.parametrize("n", [1, 1024, 3000, 100001])
def test_mask_or_and_intervals_bit_exact(n, gd):
    rng = np.random.default_rng(n + 5)
    base = rng.standard_normal(n)
    base[::3] = 0.0
    if n > 4:
        base[4] = -0.0
    # loose (overlapping, unsorted) intervals: mask / or
    s, e, val = _random_intervals(n, 20 + n // 50, rng, max_len=90, integer=False)
    for binarize_first in (False, True):
        got = gd.mask_intervals(gd.DeviceVector.from_numpy(base), s, e, val, True, 0.0, binarize_first).numpy()
        assert bits_equal(got, cpu.mask_intervals(base, s, e, val, True, 0.0, binarize_first)), binarize_first
    # sorted, non-overlapping intervals: masknot / and
    cuts = np.unique(rng.integers(0, n + 1, 40))
    s2, e2 = cuts[:-1:2].astype(np.uint32), cuts[1::2].astype(np.uint32)
    k = min(s2.size, e2.size)
    s2, e2 = s2[:k], e2[:k]
    for outside, binarize_first in ((9.5, False), (0.0, True)):
        got = gd.mask_intervals(gd.DeviceVector.from_numpy(base), s2, e2, np.ones(k), False, outside, binarize_first).numpy()
        assert bits_equal(got, cpu.mask_intervals(base, s2, e2, np.ones(k), False, outside, binarize_first))
    # minwith / maxwith are the ingest kernel without clearing
    for op in (cpu.OVERLAP_MIN, cpu.OVERLAP_MAX):
        got = gd.apply_intervals(gd.DeviceVector.from_numpy(base), s, e, val, op).numpy()
        assert bits_equal(got, cpu.apply_intervals(base, s, e, val, op))


@pytest.mark.parametrize("n", [1, 4095, 4096, 4097, 200003])
@pytest.mark.parametrize("nknots", [1, 2, 7, 2048, 2049, 10000])
def test_map_values_bit_exact(n, nknots, gd):
    rng = np.random.default_rng(n + nknots)
    x = rng.standard_normal(n) * 30
    kin = np.sort(rng.standard_normal(nknots) * 25)
    kin = np.unique(kin)
    kout = rng.standard_normal(kin.size) * 10
    if n > 3 and kin.size > 1:
        x[0], x[1], x[2] = kin[0], kin[-1], kin[kin.size // 2]        # exactly on knots
    got = gd.map_values(gd.DeviceVector.from_numpy(x), kin, kout).numpy()
    assert bits_equal(got, cpu.map_values(x, kin, kout))


@pytest.mark.parametrize("n", [1, 1024, 5000, 100001])
def test_minover_maxover_bit_exact(n, gd):
    rng = np.random.default_rng(n + 9)
    for kind in ("blocky", "real"):
        base = np.repeat(rng.integers(0, 4, n // 3 + 1), 3)[:n].astype(np.float64) if kind == "blocky" \
            else rng.standard_normal(n)
        cuts = np.unique(rng.integers(0, n + 1, 60))
        s, e = cuts[:-1:2].astype(np.uint32), cuts[1::2].astype(np.uint32)
        k = min(s.size, e.size)
        s, e = s[:k], e[:k]
        for want_max, fill in ((False, 99.0), (True, 0.0)):
            got = gd.extreme_in_intervals(gd.DeviceVector.from_numpy(base), s, e, want_max, fill).numpy()
            want = cpu.extreme_in_intervals(base, s, e, want_max, fill)
            assert bits_equal(got, want), (kind, want_max, first_diff(got, want))
    # one interval spanning many tiles, ties everywhere: the most central base wins
    flat = np.full(n, 2.0)
    got = gd.extreme_in_intervals(gd.DeviceVector.from_numpy(flat), [0], [n], True, 0.0).numpy()
    assert bits_equal(got, cpu.extreme_in_intervals(flat, [0], [n], True, 0.0))


@pytest.mark.parametrize("n", [1, 100, 20011, 300007])
@pytest.mark.parametrize("W", [9001, 20000, 65536, 1000001])
def test_windows_beyond_one_lds_tile(n, W, gd):
    """Windows longer than the tiled kernels hold fall back to whole-vector passes; same answers."""
    rng = np.random.default_rng(n + W)
    x = _signal("depth", n, rng)
    d = gd.DeviceVector.from_numpy(x)
    assert bits_equal(gd.best_extrema(d, W, True).numpy(), cpu.best_extrema(x, W, 1))
    assert bits_equal(gd.best_extrema(d, W, False).numpy(), cpu.best_extrema(x, W, 0))
    assert bits_equal(gd.sliding_sum(d, W).numpy(), cpu.sliding_sum(x, W))
    N = W | 1
    if n <= 20011:            # the oracle's localmax is O(n N)
        y = _signal("noise", n, rng)
        dy = gd.DeviceVector.from_numpy(y)
        assert bits_equal(gd.localmax(dy, N).numpy(), cpu.local_extrema(y, N, 1, 0.0))
        assert bits_equal(gd.localmin(dy, N).numpy(), cpu.local_extrema(y, N, 0, cpu.DBL_MAX))


# ------------------------------------------------------------------- clump ----
@pytest.mark.parametrize("n", [1, 2, 100, 4095, 4096, 4097, 100003, 5000011])
@pytest.mark.parametrize("L", [1, 7, 63, 64, 100, 5000])
def test_clump_anticlump_bit_exact(n, L, gd):
    """clump.c:494-736 by whole-vector scans; depth against a dyadic threshold keeps every running sum exact.
    n = 5000011 needs more than 1024 scan chunks (two totals per thread in the offsets pass)."""
    rng = np.random.default_rng(n + L)
    x = _signal("depth", n, rng)
    for T in (float(np.floor(np.median(x))) + 0.5, float(np.floor(x.mean())) - 0.25):
        for above in (True, False):
            got = gd.clump(gd.DeviceVector.from_numpy(x), T, L, above, 1.0, 0.0).numpy()
            want = cpu.clump(x, T, L, above)
            assert bits_equal(got, want), (T, above, first_diff(got, want))


def test_clump_degenerate_inputs(gd):
    x = np.full(5000, 3.0)
    for T, above, want in ((4.0, True, 0.0), (3.0, True, 1.0), (2.0, False, 0.0), (3.0, False, 1.0)):
        got = gd.clump(gd.DeviceVector.from_numpy(x), T, 100, above, 1.0, 0.0).numpy()
        assert np.all(got == want) and bits_equal(got, cpu.clump(x, T, 100, above))
    got = gd.clump(gd.DeviceVector.from_numpy(x), 3.0, 5001, True, 1.0, 0.0).numpy()     # longer than the vector
    assert np.all(got == 0.0)
    got = gd.clump(gd.DeviceVector.from_numpy(x), 3.0, 5000, True, 9.0, -9.0).numpy()    # exactly the vector
    assert np.all(got == 9.0)
