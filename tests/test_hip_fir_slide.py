"""GPU: `smooth W=101` in the reference's arithmetic through the sliding-accumulator kernel (gdsp_fir_slide.hip: every
rounded product w[m] x[j] computed once and added to the two outputs it belongs to, each output still receiving its 101
terms in the reference's order, sum.c:651-663) against the CPU oracle, bit for bit: lengths around the strips' seams and
the 16-input blocks, strips of several lengths, vectors shorter than the window, a table of vectors in one launch, NaN,
infinities, signed zeros and huge magnitudes."""
import numpy as np
import pytest

from conftest import bits_equal, first_diff
from oracle import cpu

pytestmark = pytest.mark.gpu
SEED = 20240611


@pytest.fixture(scope="module")
def gd():
    import genodsp_amd
    assert genodsp_amd.device_count() >= 1
    return genodsp_amd


def _signal(kind, n, rng):
    if kind == "depth":
        return cpu.synth_coverage(SEED, 3, 0, n, 0)
    if kind == "real":
        return cpu.synth_coverage(SEED, 3, 0, n, 1)
    x = rng.standard_normal(n) * 5
    if kind == "odd" and n > 40:
        k = rng.integers(0, n, size=max(4, n // 300))
        x[k[0::4]] = np.nan
        x[k[1::4]] = np.inf
        x[k[2::4]] = -np.inf
        x[k[3::4]] = -0.0
        x[rng.integers(0, n, size=4)] = 1e300
        x[rng.integers(0, n, size=4)] = 5e-324
    return x


SIZES = [1, 2, 13, 14, 15, 49, 50, 51, 101, 498, 512, 526, 1010, 1024, 1038, 8191, 65522, 65536, 65550, 300007]
CASES = [(n, kind, 0) for n in SIZES for kind in ("depth", "real", "noise", "odd")] \
      + [(n, kind, strip) for strip in (512, 1024, 4096) for n in (526, 1038, 4110, 65550, 300007) for kind in ("real", "odd")]


# GDSP_FIR_SLIDE=1: every lane fetches its own strip; =2: blocks of inputs staged by LDS-DMA a block ahead, outputs paired in LDS
FORMS = ["1", "2"]


@pytest.mark.parametrize("form", FORMS)
@pytest.mark.parametrize("n,kind,strip", CASES)
def test_sliding_accumulators_give_the_reference_bits(n, kind, strip, form, gd, monkeypatch):
    monkeypatch.setenv("GDSP_FIR_SLIDE", form)
    if strip:
        monkeypatch.setenv("GDSP_FIR_SLIDE_STRIP", str(strip))
    rng = np.random.default_rng(n * 7 + strip)
    x = _signal(kind, n, rng)
    got = gd.smooth(gd.DeviceVector.from_numpy(x), 101, mode=gd.FIR_EXACT).numpy()
    want = cpu.smooth(x, 101)
    assert bits_equal(got, want), first_diff(got, want)


@pytest.mark.parametrize("form", FORMS)
def test_a_table_of_vectors_in_one_launch(form, gd, monkeypatch):
    monkeypatch.setenv("GDSP_FIR_SLIDE", form)
    monkeypatch.setenv("GDSP_FIR_SLIDE_STRIP", "512")
    rng = np.random.default_rng(3)
    lens = [70001, 1, 0, 513, 40000, 14, 498, 99999, 2, 1024] + [3000 + 17 * i for i in range(30)]      # more than one table of 32
    xs = [_signal("odd" if i % 3 == 0 else "real", n, rng) if n else np.zeros(0) for i, n in enumerate(lens)]
    ins = [gd.DeviceVector.from_numpy(x) for x in xs]
    outs = gd.smooth_batch(ins, 101, mode=gd.FIR_EXACT)
    for x, o in zip(xs, outs):
        want = cpu.smooth(x, 101) if x.size else x
        assert bits_equal(o.numpy(), want), (x.size, first_diff(o.numpy(), want))


@pytest.mark.parametrize("form", FORMS)
def test_same_bits_as_the_direct_kernel_on_a_whole_chromosome(form, gd, monkeypatch):
    """20 Mbp of real-valued coverage: every output equal to the direct kernel's (which the golden vectors and the
    in-run check of bench.py hold to the reference binary)"""
    n = 20_000_003
    v = gd.synth_coverage(SEED, 5, 0, n, 1)
    monkeypatch.setenv("GDSP_FIR_SLIDE", "0")
    direct = gd.smooth(v, 101, mode=gd.FIR_EXACT).numpy()
    monkeypatch.setenv("GDSP_FIR_SLIDE", form)
    slide = gd.smooth(v, 101, mode=gd.FIR_EXACT).numpy()
    assert bits_equal(slide, direct), first_diff(slide, direct)
