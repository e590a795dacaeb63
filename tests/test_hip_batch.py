"""GPU: the one-launch-per-device forms (gdsp_*_batch, include/genodsp_hip.h) give, for every vector of the table,
the bits of the single-vector call -- they run the same tile code, only the block-to-tile map differs -- and the
single-vector calls are held to the oracle elsewhere (tests/test_hip_parity.py).  41 vectors of awkward lengths (more
than one 32-entry table; empty, shorter than a window, a few tiles, many tiles), checked against both the
single-vector call and, for the BASELINE pipelines, the oracle.  Replaces the per-chromosome loop of genodsp.c:909-921.
"""
import numpy as np
import pytest

from conftest import bits_equal
from oracle import cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gd():
    import genodsp_amd
    return genodsp_amd


def lengths():
    rng = np.random.default_rng(77)
    fixed = [1, 2, 7, 49, 50, 51, 100, 101, 499, 1001, 2303, 2304, 3984, 3985, 4096, 4097, 8192, 16385, 300007, 123457]
    return fixed + [int(x) for x in rng.integers(1, 60000, 21)]


@pytest.fixture(scope="module")
def vectors(gd):
    rng = np.random.default_rng(5)
    host = []
    for i, n in enumerate(lengths()):
        if i % 3 == 0:
            x = cpu.synth_coverage(20240611, i, 0, n, 0)                     # piecewise-constant depth: ties everywhere
        elif i % 3 == 1:
            x = cpu.synth_coverage(20240611, i, 0, n, 1)
        else:
            x = rng.normal(size=n) * 10.0
        host.append(x)
    return host, [gd.DeviceVector.from_numpy(x) for x in host]


def same(gd, batch_out, single_fn, vecs):
    for i, (b, v) in enumerate(zip(batch_out, vecs)):
        want = single_fn(v).numpy()
        assert bits_equal(b.numpy(), want), (i, v.n)


@pytest.mark.parametrize("mode", ["exact", "fma", "hann"])
@pytest.mark.parametrize("W", [101, 21, 301])
def test_smooth_batch(gd, vectors, mode, W):
    host, vecs = vectors
    m = {"exact": gd.FIR_EXACT, "fma": gd.FIR_FMA, "hann": gd.FIR_HANN}[mode]
    out = gd.smooth_batch(vecs, W, mode=m)
    same(gd, out, lambda v: gd.smooth(v, W, mode=m), vecs)
    if mode == "exact" and W == 101:
        for x, o in zip(host, out):
            if x.size > 50:                                                   # (shorter than the half window: reference UB)
                assert bits_equal(o.numpy(), cpu.smooth(x, 101))


@pytest.mark.parametrize("want_max", [True, False])
@pytest.mark.parametrize("mode", ["exact", "fma"])
def test_smooth_local_extrema_batch(gd, vectors, mode, want_max):
    host, vecs = vectors
    m = {"exact": gd.FIR_EXACT, "fma": gd.FIR_FMA}[mode]
    fill = 0.0 if want_max else 1.7976931348623157e308
    out = gd.smooth_local_extrema_batch(vecs, 101, 11, want_max, fill, mode=m)
    same(gd, out, lambda v: gd.smooth_local_extrema(v, 101, 11, want_max, fill, mode=m), vecs)
    if mode == "exact":
        for x, o in zip(host, out):
            if x.size > 50:
                assert bits_equal(o.numpy(), cpu.local_extrema(cpu.smooth(x, 101), 11, int(want_max), fill))


@pytest.mark.parametrize("N", [3, 5, 11, 31, 1001])
def test_local_and_best_extrema_batch(gd, vectors, N):
    host, vecs = vectors
    same(gd, gd.local_extrema_batch(vecs, N, True, 0.0), lambda v: gd.local_extrema(v, N, True, 0.0), vecs)
    same(gd, gd.local_extrema_batch(vecs, N, False, 9e99), lambda v: gd.local_extrema(v, N, False, 9e99), vecs)
    same(gd, gd.best_extrema_batch(vecs, N + 1, True), lambda v: gd.best_extrema(v, N + 1, True), vecs)
    same(gd, gd.best_extrema_batch(vecs, N, False), lambda v: gd.best_extrema(v, N, False), vecs)


@pytest.mark.parametrize("length", [5, 40, 1001, 3000, 9000])
def test_morphology_batch(gd, vectors, length):
    host, vecs = vectors
    left, right = gd.split_length(length)
    T = 3.0
    same(gd, gd.dilate_batch(vecs, left, right, T=T), lambda v: gd.dilate(v, left, right, T=T), vecs)
    same(gd, gd.erode_batch(vecs, left, right, T=T, one=2.0, zero=-1.0), lambda v: gd.erode(v, left, right, T=T, one=2.0, zero=-1.0), vecs)
    for binarize in (None, (0.5, True, 7.0, 3.0)):
        out = gd.dilate_erode_batch(vecs, left, right, left, right, d_T=T, binarize=binarize)
        same(gd, out, lambda v: gd.dilate_erode(v, left, right, left, right, d_T=T, binarize=binarize), vecs)
    if length == 1001:
        out = gd.dilate_erode_batch(vecs, left, right, left, right, d_T=T, binarize=(0.0, False, 1.0, 0.0))
        for x, o in zip(host, out):
            want = cpu.binarize(cpu.erode(cpu.dilate(x, left, right, T=T), left, right))
            assert bits_equal(o.numpy(), want)


def test_pointwise_batch(gd, vectors):
    host, vecs = vectors

    def both(batch_fn, single_fn):
        a = [v.copy() for v in vecs]
        b = [v.copy() for v in vecs]
        gd.sync()
        batch_fn(a)
        for x, y in zip(a, b):
            single_fn(y)
            assert bits_equal(x.numpy(), y.numpy())

    both(lambda vs: gd.binarize_batch(vs, 4.0), lambda v: gd.binarize(v, 4.0))
    both(lambda vs: gd.binarize_batch(vs, 4.0, True, 3.0, -2.0), lambda v: gd.binarize(v, 4.0, True, 3.0, -2.0))
    both(lambda vs: gd.clip_batch(vs, lo=1.0, hi=9.5), lambda v: gd.clip(v, lo=1.0, hi=9.5))
    both(lambda vs: gd.clip_batch(vs, hi=9.5), lambda v: gd.clip(v, hi=9.5))
    both(lambda vs: gd.erase_batch(vs, lo=2.0, hi=20.0, keep_inside=True, zero=-1.0), lambda v: gd.erase(v, lo=2.0, hi=20.0, keep_inside=True, zero=-1.0))
    both(lambda vs: gd.erase_batch(vs, lo=2.0), lambda v: gd.erase(v, lo=2.0))
    both(lambda vs: gd.add_constant_batch(vs, 0.1), lambda v: gd.add_constant(v, 0.1))
    both(lambda vs: gd.abs_batch(vs), lambda v: gd.abs_(v))


def test_batch_refuses_what_the_single_call_refuses(gd, vectors):
    host, vecs = vectors
    with pytest.raises(gd.GdspError):
        gd.smooth_batch(vecs, 100)                                            # even window
    with pytest.raises(gd.GdspError):
        gd.smooth_batch(vecs[:3], 101, outs=vecs[:3])                         # out aliases in
    assert gd.smooth_batch([], 101) == []
