"""GPU: dilate / erode / close / open beyond what one LDS tile can stage (262 k bases of reach): the set as bits in HBM
workspace (gdsp_morph.hip, morph_any).  The reference accepts any length (morphology.c:696-866, :1163-1315).
Bit-exact against the oracle: first the workspace form forced onto the small cases the tile kernels are tested on
(every seam of the 64-bit words, the 256-word groups, vector ends), then real long reaches on a 20 Mbp vector."""
import os

import numpy as np
import pytest

from conftest import bits_equal
from oracle import cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gd():
    import genodsp_amd
    assert genodsp_amd.device_count() >= 1
    return genodsp_amd


@pytest.fixture()
def forced(monkeypatch):
    monkeypatch.setenv("GDSP_MORPH_FORCE_BITS", "1")


def islands(n, rng, max_gap=40, max_run=25):
    v = np.zeros(n)
    pos = 0
    while pos < n:
        gap = int(rng.integers(1, max_gap + 1))
        run = int(rng.integers(1, max_run + 1))
        v[pos + gap:pos + gap + run] = float(rng.integers(1, 9))
        pos += gap + run
    return v


@pytest.mark.parametrize("n", [1, 2, 63, 64, 65, 127, 128, 129, 16383, 16384, 16385, 100000, 300007])
@pytest.mark.parametrize("L", [1, 2, 3, 10, 63, 64, 65, 1001, 5000])
def test_workspace_form_on_the_tile_kernels_cases(n, L, gd, forced):
    rng = np.random.default_rng(n + L)
    x = islands(n, rng, max_gap=3 * L + 5, max_run=3 * L + 5)
    if n > 3:
        x[rng.integers(0, n, 3)] = 0.5
    d = gd.DeviceVector.from_numpy(x)
    left, right = gd.split_length(L)
    for T in (0.0, 0.75):
        assert bits_equal(gd.morph_bits("dilate", d, left, right, T, 1.0, 0.0).numpy(), cpu.dilate(x, left, right, T)), ("dilate", T)
        assert bits_equal(gd.morph_bits("erode", d, left, right, T, 1.0, 0.0).numpy(), cpu.erode(x, left, right, T)), ("erode", T)
        assert bits_equal(gd.morph_bits("close", d, float(L), T, 1.0, 0.0).numpy(), cpu.close(x, L, T)), ("close", T)
        assert bits_equal(gd.morph_bits("open", d, float(L), T, 1.0, 0.0).numpy(), cpu.open_(x, L, T)), ("open", T)
    assert bits_equal(gd.morph_bits("dilate", d, 0, L, 0.0, 7.0, -1.0).numpy(), cpu.dilate(x, 0, L, 0.0, 7.0, -1.0))
    assert bits_equal(gd.morph_bits("erode", d, L, 0, 0.0, 7.0, -1.0).numpy(), cpu.erode(x, L, 0, 0.0, 7.0, -1.0))


def test_workspace_form_nan_and_degenerate_inputs(gd, forced):
    rng = np.random.default_rng(77)
    n = 20000
    x = islands(n, rng, max_gap=300, max_run=200)
    x[0] = np.nan
    x[rng.integers(1, n, 40)] = np.nan
    d = gd.DeviceVector.from_numpy(x)
    for left, right in ((5, 5), (0, 300), (700, 1), (2000, 2000), (30000, 30000)):
        assert bits_equal(gd.morph_bits("dilate", d, left, right, 0.0, 1.0, 0.0).numpy(), cpu.dilate(x, left, right, 0.0))
        assert bits_equal(gd.morph_bits("erode", d, left, right, 0.0, 1.0, 0.0).numpy(), cpu.erode(x, left, right, 0.0))
    for m in (1, 500, 40000):
        for y in (np.ones(m), np.zeros(m)):
            dy = gd.DeviceVector.from_numpy(y)
            assert bits_equal(gd.morph_bits("dilate", dy, 5, 6, 0.0, 1.0, 0.0).numpy(), cpu.dilate(y, 5, 6))
            assert bits_equal(gd.morph_bits("erode", dy, 5, 6, 0.0, 1.0, 0.0).numpy(), cpu.erode(y, 5, 6))
            assert bits_equal(gd.morph_bits("close", dy, 10.0, 0.0, 1.0, 0.0).numpy(), cpu.close(y, 10))
            assert bits_equal(gd.morph_bits("open", dy, 10.0, 0.0, 1.0, 0.0).numpy(), cpu.open_(y, 10))
    z = islands(50000, rng)
    dz = gd.DeviceVector.from_numpy(z)
    for L in (0.0, 0.5, 7.5, 39.999, 1e9):
        assert bits_equal(gd.morph_bits("close", dz, L, 0.0, 1.0, 0.0).numpy(), cpu.close(z, L)), L
        assert bits_equal(gd.morph_bits("open", dz, L, 0.0, 1.0, 0.0).numpy(), cpu.open_(z, L)), L


@pytest.mark.parametrize("reach", [300000, 5000000])
def test_long_reach_on_20_mbp(reach, gd):
    """beyond the tile kernels (GDSP_EINVAL there): the public entry points fall through to the workspace form"""
    n = 20000003
    rng = np.random.default_rng(reach)
    x = np.zeros(n)
    pos = 0
    while pos < n:                                                    # runs and gaps from tiny to several times the reach
        gap = int(rng.choice([7, 900, reach // 3, reach, reach + 1, 2 * reach + 5]))
        run = int(rng.choice([1, 50, reach // 2, reach - 1, reach, 3 * reach]))
        x[pos + gap:pos + gap + run] = 3.0
        pos += gap + run
    x[:reach + 10] = 0.0                                              # (keeps the reference's erode off its u32 underflow)
    d = gd.DeviceVector.from_numpy(x)
    left, right = gd.split_length(reach)
    assert gd.lib().gdsp_dilate(d.ptr, d.like().ptr, n, left, right, 0.0, 1.0, 0.0, None) != 0       # the tile kernel declines
    assert bits_equal(gd.dilate(d, left, right).numpy(), cpu.dilate(x, left, right)), "dilate"
    assert bits_equal(gd.erode(d, left, right).numpy(), cpu.erode(x, left, right)), "erode"
    assert bits_equal(gd.dilate(d, 0, reach).numpy(), cpu.dilate(x, 0, reach)), "dilate one-sided"
    assert bits_equal(gd.erode(d, reach, 0).numpy(), cpu.erode(x, reach, 0)), "erode one-sided"
    assert bits_equal(gd.close(d, reach).numpy(), cpu.close(x, reach)), "close"
    assert bits_equal(gd.open_(d, reach).numpy(), cpu.open_(x, reach)), "open"
