"""GPU: seeded random pipelines of operators through the HIP path and through the oracle, bit for bit.

Every fixture in tests/golden pins one operator (or one short chain) with the reference's own output;
this test is about what happens between operators: 160 random chains of up to seven operators with
random arguments over three chromosomes of awkward lengths, each operator fed by the previous one's
output.  Only operators whose GPU form is bit-identical on any input are chained freely; the running
sums (slidingsum, cumulativesum, clump), which are bit-identical on exactly-summable signals, are
drawn only while the signal is still integer valued."""
import numpy as np
import pytest

from backends import GpuBackend, OracleBackend
from conftest import bits_equal, first_diff
from oracle import cpu
from pipeline import Runner

pytestmark = pytest.mark.gpu

CHROMS = [("cA", 70001), ("cB", 12289), ("cC", 4097)]


def _draw(rng, integral):
    """one operator as command-line text, and whether its output is still integer valued"""
    pick = rng.integers(0, 18 if integral else 14)
    W = int(rng.choice([3, 5, 11, 21, 101, 257]))
    N = int(rng.choice([3, 7, 11, 33, 101, 1001]))
    L = int(rng.choice([1, 2, 5, 30, 200, 1001]))
    T = float(rng.integers(0, 12))
    if pick == 0:
        return "= smooth W=%d" % W, False
    if pick == 1:
        return "= localmax N=%d" % N, integral
    if pick == 2:
        return "= localmin N=%d --infinity=%g" % (N, float(rng.integers(20, 99))), integral
    if pick == 3:
        return "= bestmax W=%d" % int(rng.choice([3, 10, 100, 1000, 3000])), integral
    if pick == 4:
        return "= bestmin W=%d" % int(rng.choice([4, 33, 640, 2049])), integral
    if pick == 5:
        return "= dilate %d --threshold=%g" % (L, T), True
    if pick == 6:
        return "= erode %d --threshold=%g --one=%d" % (L, T, int(rng.integers(1, 9))), True
    if pick == 7:
        return "= close %d --threshold=%g" % (L, T), True
    if pick == 8:
        return "= open %d --threshold=%g --zero=-1" % (L, T), True
    if pick == 9:
        return "= binarize %g%s" % (T, " --ties:above" if rng.random() < 0.5 else ""), True
    if pick == 10:
        return "= clip --min=%g --max=%g" % (T / 2, T + 5), integral and (T % 2 == 0)
    if pick == 11:
        return "= erase --min=%g --max=%g%s" % (T, T + 6, " --keep:inside" if rng.random() < 0.5 else ""), integral
    if pick == 12:
        c = float(rng.integers(-5, 6)) if integral else float(rng.standard_normal())
        return "= addconst %r" % c, integral
    if pick == 13:
        return "= sum W=%d" % int(rng.choice([3, 64, 100, 1000])), integral
    if pick == 14:
        return "= slidingsum W=%d" % int(rng.choice([4, 101, 1000])), True
    if pick == 15:
        return "= cumulativesum", True
    if pick == 16:
        return "= clump %g --length=%d" % (T + 0.5, int(rng.choice([5, 100, 2000]))), True
    return "= anticlump %g --length=%d --one=2" % (T + 0.25, int(rng.choice([7, 300]))), True


@pytest.mark.parametrize("seed", range(160))
def test_random_pipeline_is_the_oracle_bit_for_bit(seed):
    rng = np.random.default_rng(1000 + seed)
    vectors = {}
    for i, (c, n) in enumerate(CHROMS):
        v = cpu.synth_coverage(777 + seed, i, 0, n, 1 if seed % 4 == 3 else 0)      # every fourth: real valued
        if seed % 3 == 1:
            v[(np.arange(n) // 1500) % 2 == 1] = 0.0                # islands
        vectors[c] = v
    ops, integral = [], (seed % 4 != 3)
    for _ in range(int(rng.integers(2, 8))):
        text, integral = _draw(rng, integral)
        ops.append(text)
    pipeline = " ".join(ops)
    want = Runner(OracleBackend(), CHROMS, vectors).run(pipeline)
    got = Runner(GpuBackend(), CHROMS, vectors).run(pipeline)
    for c, _ in CHROMS:
        a, b = got.result(c), want.result(c)
        assert bits_equal(a, b), "%s | %s: first difference at %s" % (pipeline, c, first_diff(a, b))
