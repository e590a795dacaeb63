"""GPU: one chromosome of 2^32 - 1 bases -- the longest vector the reference's u32 lengths allow
(/root/reference genodsp_interface.h:42-45) -- through the kernels of BASELINE configs[1..4] and through the C driver.

34.4 GB per vector; the oracle recomputes windows from the regenerated synthetic signal at both ends of the
chromosome, either side of base 2^31 (where a signed 32-bit index turns negative), at the last seams of every kernel's
tiling below 2^32 and at a few random places, bit for bit (smooth --smooth=hann within its stated bound).
Operators: smooth W=101 (exact, hann), smooth=localmax fused, dilate/erode 1001 and the fused chain, binarize,
interval ingest at the very top, report.  Then `genodsp_hip chrU:4294967295` end to end: reads near both ends and across
2^31, `= smooth W=101`, stdout compared with text built from oracle windows (positions print through %d like
genodsp.c:1640-1668, so those beyond 2^31 come out negative in both programs).
"""
import ctypes
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, bits_equal
from oracle import cpu

pytestmark = pytest.mark.gpu

SEED = 20240611
N = 2 ** 32 - 1
CHROM = 3
M = 6000
BIN = os.path.join(ROOT, "genodsp_amd", "genodsp_hip")


@pytest.fixture(scope="module")
def gd():
    import genodsp_amd
    return genodsp_amd


@pytest.fixture(scope="module")
def real(gd):
    v = gd.synth_coverage(SEED, CHROM, 0, N, 1)
    gd.sync()
    return v


def places():
    top = []
    for tile in (2304, 2304 - 10, 3984, 4096, 16384, 32768, 1 << 18):      # last seam of each tiling below 2^32
        top.append(min(N - M, (N // tile) * tile - M // 2))
    rng = np.random.default_rng(5)
    return [0, 2 ** 31 - M // 2, 2 ** 31, N - M] + top + [int(s) for s in rng.integers(0, N - M, 6)]


def fetch(vec, start, count):
    return vec.buf.download(np.float64, count, vec.offset + 8 * start)


def regenerate(start, count, halo, mode=1):
    lo, hi = max(0, start - halo), min(N, start + count + halo)
    return cpu.synth_coverage(SEED, CHROM, lo, hi - lo, mode), start - lo


def check(vec, oracle_fn, halo, bound_fn=None):
    for s in places():
        x, left = regenerate(s, M, halo)
        want = oracle_fn(x)[left:left + M]
        got = fetch(vec, s, M)
        if bound_fn is None:
            assert bits_equal(got, want), (s, int(np.flatnonzero(got != want)[0]))
        else:
            assert np.all(np.abs(got - want) <= bound_fn(x)[left:left + M]), s


def test_signal_regenerates(gd, real):
    check(real, lambda x: x, 0)


def test_smooth_and_peaks(gd, real):
    taps = cpu.hann_window(101)
    sm = gd.smooth(real, 101, mode=gd.FIR_EXACT)
    check(sm, lambda x: cpu.smooth(x, 101), 50)
    pk = gd.localmax(sm, 11)
    check(pk, lambda x: cpu.local_extrema(cpu.smooth(x, 101), 11, 1, 0.0), 55)
    del sm
    fused = gd.smooth_local_extrema(real, 101, 11, True, 0.0)
    check(fused, lambda x: cpu.local_extrema(cpu.smooth(x, 101), 11, 1, 0.0), 55)
    del fused, pk
    hn = gd.smooth(real, 101, mode=gd.FIR_HANN)
    check(hn, lambda x: cpu.smooth(x, 101), 50, bound_fn=lambda x: 101 * 2.0 ** -52 * cpu.fir(np.abs(x), taps))


def test_morphology_and_binarize(gd, real):
    left, right = gd.split_length(1001)
    T = 60.0                                                           # a sparse set: depth x factor above 60
    dl = gd.dilate(real, left, right, T=T)
    check(dl, lambda x: cpu.dilate(x, left, right, T=T), 1002)
    er = gd.erode(dl, left, right)
    check(er, lambda x: cpu.erode(cpu.dilate(x, left, right, T=T), left, right), 2004)
    del dl
    fused = gd.dilate_erode(real, left, right, left, right, d_T=T, binarize=(0.0, False, 1.0, 0.0))
    check(fused, lambda x: cpu.binarize(cpu.erode(cpu.dilate(x, left, right, T=T), left, right)), 2004)
    del fused
    b = gd.binarize(er, 0.0)
    check(b, lambda x: cpu.binarize(cpu.erode(cpu.dilate(x, left, right, T=T), left, right)), 2004)
    # report: runs of the eroded set; every run the oracle finds in a window (clear of the window's ends) must be there
    s, e, v = gd.report_runs(b)
    assert s.size > 1000 and np.all(s < e) and np.all(e[:-1] <= s[1:]) and int(e[-1]) <= N and np.all(v == 1.0)
    assert int(s[-1]) > 2 ** 31                                        # (runs beyond the sign bit of a 32-bit index)
    for p in places():
        x, lft = regenerate(p, M, 2004)
        w = cpu.binarize(cpu.erode(cpu.dilate(x, left, right, T=T), left, right))[lft:lft + M]
        ws, we, wv = cpu.report_runs(w)
        for a, z in zip(ws.tolist(), we.tolist()):
            if a == 0 or z == M:
                continue                                               # may continue outside the window
            i = int(np.searchsorted(s, p + a))
            assert i < s.size and int(s[i]) == p + a and int(e[i]) == p + z, (p, a, z)


def test_interval_ingest_at_the_top(gd):
    v = gd.DeviceVector(N)
    gd.fill(v, 0.0)
    start = np.array([0, 5, 2 ** 31 - 10, N - 300, N - 200, N - 1], np.uint32)
    end = np.array([7, 9, 2 ** 31 + 10, N - 100, N, N], np.uint32)
    val = np.array([1.5, 2.0, 3.0, 4.0, 0.25, 8.0])
    gd.apply_intervals(v, start, end, val)
    for p, m in ((0, 64), (2 ** 31 - 32, 64), (N - 400, 400)):
        want = np.zeros(m)
        for a, z, x in zip(start.tolist(), end.tolist(), val.tolist()):
            lo, hi = max(a, p), min(z, p + m)
            if lo < hi:
                want[lo - p:hi - p] += x
        assert bits_equal(fetch(v, p, m), want), p
    s, e, x = gd.report_runs(v)
    assert (int(s[0]), int(e[0]), float(x[0])) == (0, 5, 1.5)
    assert (int(s[-1]), int(e[-1]), float(x[-1])) == (N - 1, N, 8.25)


def test_cumulative_sum_over_u32max_bases(gd):
    """cumulativesum in one pass (gdsp_sums.hip: cumsum_lookback_kernel) at the longest vector there is: 1 048 576 chunks,
    256 super-groups (the sum over super-groups runs in four batches of 64 lanes), a ragged last chunk.  A vector of
    ones -- every partial sum an integer below 2^53, so the bits are the reference's (sum.c:776-792): out[i] = i + 1 --
    and one of read depth, held to the oracle's running sum restarted from the exact prefix at each window."""
    v = gd.DeviceVector(N)
    gd.fill(v, 1.0)
    gd.cumulative_sum(v)
    for p in places():
        assert bits_equal(fetch(v, p, M), np.arange(p + 1, p + M + 1, dtype=np.float64)), p
    assert fetch(v, N - 1, 1)[0] == float(N)
    del v
    d = gd.synth_coverage(SEED, CHROM, 0, N, 0)                       # integer depth: sums stay exact (< 2^53)
    gd.cumulative_sum(d)
    for p in places():
        got = fetch(d, p, M)
        x = cpu.synth_coverage(SEED, CHROM, p, M, 0)
        # the prefix in front of the window is whatever the device says at p-1 (checked against itself at the seams by
        # the increments): inside the window every increment must be the signal, exactly
        first = got[0] - x[0]
        assert bits_equal(got, first + np.cumsum(x)), p
        if p > 0:
            assert fetch(d, p - 1, 1)[0] == first
    total = fetch(d, N - 1, 1)[0]
    lo, hi, cnt = gd.genome_minmax([d], 1, -gd.DBL_MAX, gd.DBL_MAX)
    assert hi == total and lo >= 0.0                                   # non-decreasing: the last value is the largest


def test_driver_on_a_chromosome_of_u32max_bases(tmp_path):
    """genodsp_hip end to end: u32 parsing of coordinates up to 4294967295, the ingest binning, smooth W=101 and the
    report over 4.29 G bases.  Expected text from oracle windows."""
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "genodsp_amd", "host")])
    rng = np.random.default_rng(12)
    spots = [(0, 3000), (2 ** 31 - 1500, 2 ** 31 + 1500), (N - 3000, N)]
    reads = []
    for lo, hi in spots:
        for _ in range(400):
            a = int(rng.integers(lo, hi - 1))
            z = min(hi, a + int(rng.integers(1, 200)))
            reads.append((a, z, int(rng.integers(1, 9))))
    reads.append((N - 1, N, 7))
    text = "".join("chrU\t%d\t%d\t%d\n" % r for r in reads)
    p = subprocess.run([BIN, "chrU:%d" % N, "--precision=6", "=", "smooth", "W=101"], input=text, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    want = []
    for lo, hi in spots:
        a, z = max(0, lo - 200), min(N, hi + 200)
        x = np.zeros(z - a)
        for s, e, v in reads:
            if s >= a and e <= z:
                x[s - a:e - a] += v
        y = cpu.smooth(x, 101)
        rs, re_, rv = cpu.report_runs(y)
        for s, e, v in zip(rs.tolist(), re_.tolist(), rv.tolist()):
            want.append("chrU\t%d\t%d\t%.6f" % (ctypes.c_int32(a + s).value, ctypes.c_int32(a + e).value, v))
    got = p.stdout.splitlines()
    assert len(got) == len(want) and len(got) > 6000
    assert got == want
