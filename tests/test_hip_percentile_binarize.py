"""GPU: `= percentile P = binarize --threshold=percentileP` in one read of the signal (gdsp_percentiles_binarize):
the percentile values are gdsp_percentiles' and the outputs are gdsp_binarize's against that value, bit for bit --
whether the counting pass settled every base (one_pass), some sources fell to a pass of their own, or the whole call
took the radix route.  Reference: percentile.c:392-751 feeding logical.c:216-268."""
import numpy as np
import pytest

from conftest import bits_equal
from oracle import cpu

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gd():
    import genodsp_amd
    return genodsp_amd


def signals(n, seed):
    rng = np.random.default_rng(seed)
    out = {"real": cpu.synth_coverage(20240611, 3, 0, n, 1), "depth": cpu.synth_coverage(20240611, 4, 0, n, 0)}
    x = rng.standard_normal(n) * 5.0
    x[::97] = np.nan
    x[5::1013] = np.inf
    x[7::1019] = -np.inf
    out["noise with nan and inf"] = x
    return out


@pytest.mark.parametrize("n", [5000, 1 << 20, 3_000_001])
def test_fused_equals_percentile_then_binarize(n, gd):
    for name, x in signals(n, n).items():
        parts = [x[: n // 3], x[n // 3: n // 3 + 16 * ((n // 2) // 16)], x[n // 3 + 16 * ((n // 2) // 16):]]
        vecs = [gd.DeviceVector.from_numpy(p) for p in parts]
        for pts, which in (([99000], 0), ([50000, 90000, 99900], 1), ([100], 0)):
            for kw in ({}, {"ties_above": True, "one": 7.0, "zero": -2.0}, {"lo": 1.0, "hi": 40.0}, {"window": 3}):
                cnt, vals = gd.percentile(vecs, pts, **{k: v for k, v in kw.items() if k in ("lo", "hi", "window")})
                c2, v2, outs, one_pass = gd.percentile_binarize(vecs, pts, which=which, **kw)
                assert c2 == cnt and bits_equal(np.array(v2), np.array(vals)), (name, pts, kw)
                if cnt == 0:
                    continue
                T = vals[which]
                for p, o in zip(parts, outs):
                    want = cpu.binarize(p, T, kw.get("ties_above", False), kw.get("one", 1.0), kw.get("zero", 0.0))
                    assert bits_equal(o.numpy(), want), (name, pts, kw, one_pass)
                if n >= (1 << 21) and name == "real" and "window" not in kw and pts != [100]:
                    assert one_pass, (pts, kw)                     # the counting pass settled it (the route under test)
                # (the 0.1th percentile of this signal is the zero a third of its bases hold: more ties inside the bracket
                #  than a strip takes, so those sources are binarized by a pass of their own -- same output)
                if "window" in kw or n < 10000:
                    assert not one_pass


def test_fused_leaves_sources_intact_and_reports_nothing_when_nothing_qualifies(gd):
    x = cpu.synth_coverage(20240611, 1, 0, 2_500_000, 1)
    v = gd.DeviceVector.from_numpy(x)
    cnt, vals, outs, one_pass = gd.percentile_binarize([v], [99000])
    assert bits_equal(v.numpy(), x) and one_pass
    cnt, vals, outs, one_pass = gd.percentile_binarize([v], [99000], lo=1e30, hi=1e31)
    assert cnt == 0 and vals == [] and not one_pass


# resident: decided on the device, its selects in one workgroup's LDS (pc_ls_*); resident_digits: the same with a launch
# per radix digit (what a call across ranks takes); chained / host: the digits with five read-backs / one per digit
# resident_gives_up: the LDS select of the candidates declares its cell too big (a hook): the answer then comes from digit
# passes over the candidate list, asked for by the host after the read-back
ROUTES = {"resident": {}, "resident_digits": {"GDSP_PERCENTILE_LDS_SELECT": "0"}, "resident_gives_up": {"GDSP_PERCENTILE_LDS_GIVEUP": "1"},
          "chained": {"GDSP_PERCENTILE_RESIDENT_OFF": "1"},
          "host": {"GDSP_PERCENTILE_RESIDENT_OFF": "1", "GDSP_PERCENTILE_CHAIN_OFF": "1"}}


@pytest.mark.parametrize("n", [70_000, 2_100_001, 9_000_003])
def test_the_routes_of_a_one_device_call_agree(n, gd, monkeypatch):
    """A call on one device with nothing to reduce is decided on the device (pc_resident: one read-back); the same call
    with the digits chained (five read-backs) and with a read-back per digit must give the same population, the same
    values bit for bit and the same binarized outputs -- on coverage, read depth (ties on the pivots), NaN / infinities /
    negative values (the integer compares of the counting pass give way to FP64 ones in those waves), a constant vector
    (one distinct value: every select ends at its first digit), bounds, a stride, several percentiles at once."""
    sig = signals(n, n + 1)
    sig["constant"] = np.full(n, 3.25)
    sig["negative"] = -np.abs(sig["real"]) - 1.0
    for name, x in sig.items():
        vecs = [gd.DeviceVector.from_numpy(x[: n // 2 + 1]), gd.DeviceVector.from_numpy(x[n // 2 + 1:])]
        for pts, kw in (([99000], {}), ([500, 50000, 99990], {}), ([75000], {"lo": 1.0, "hi": 60.0}), ([90000], {"window": 3})):
            seen = {}
            for route, env in ROUTES.items():
                for k in ("GDSP_PERCENTILE_RESIDENT_OFF", "GDSP_PERCENTILE_CHAIN_OFF", "GDSP_PERCENTILE_LDS_SELECT", "GDSP_PERCENTILE_LDS_GIVEUP"):
                    monkeypatch.delenv(k, raising=False)
                for k, v in env.items():
                    monkeypatch.setenv(k, v)
                cnt, vals = gd.percentile(vecs, pts, **kw)
                st = gd.percentile_stats()
                if route.startswith("resident") and st["route"] == gd.SELECT_BRACKET and len(pts) <= 8:
                    assert st["resident"] == 1 or st["sample"] < 256, (name, pts, kw, st)
                if not route.startswith("resident"):
                    assert st["resident"] == 0
                fused = gd.percentile_binarize(vecs, pts, which=len(pts) - 1, **kw)
                seen[route] = (cnt, np.array(vals), fused[0], np.array(fused[1]), [o.numpy() for o in fused[2]])
            ref = seen["host"]
            for route in ("resident", "resident_digits", "resident_gives_up", "chained"):
                got = seen[route]
                assert got[0] == ref[0] and got[2] == ref[2] and bits_equal(got[1], ref[1]) and bits_equal(got[3], ref[3]), (name, pts, kw, route)
                for a, b in zip(got[4], ref[4]):
                    assert bits_equal(a, b), (name, pts, kw, route)


def test_resident_route_gives_way_when_it_cannot_settle_a_call(gd):
    """What the resident route cannot settle raises its status word before anything observable is written, and the call
    runs again the old way: a subsample too small to place pivots (an explicit sample target of 100 values), a pivot that
    is a NaN (a signal whose top two per cent are NaNs), more percentiles than its state holds (nine)."""
    x = cpu.synth_coverage(20240611, 5, 0, 1_500_000, 1)
    v = gd.DeviceVector.from_numpy(x)
    want = gd.percentile([v], [99000], strategy=gd.SELECT_RADIX)
    got = gd.percentile([v], [99000], strategy=gd.SELECT_BRACKET, sample_target=100)
    assert got[0] == want[0] and bits_equal(np.array(got[1]), np.array(want[1]))
    c, vals, outs, one_pass = gd.percentile_binarize([v], [99000], strategy=gd.SELECT_BRACKET, sample_target=100)
    assert c == want[0] and bits_equal(np.array(vals), np.array(want[1])) and bits_equal(outs[0].numpy(), cpu.binarize(x, vals[0], False, 1.0, 0.0))
    y = x.copy()
    y[::50] = np.nan
    w = gd.DeviceVector.from_numpy(y)
    want = gd.percentile([w], [99500], strategy=gd.SELECT_RADIX)
    got = gd.percentile([w], [99500])
    assert got[0] == want[0] and bits_equal(np.array(got[1]), np.array(want[1]))
    nine = [10000 * k for k in range(1, 10)]
    want = gd.percentile([v], nine, strategy=gd.SELECT_RADIX)
    got = gd.percentile([v], nine)
    assert gd.percentile_stats()["resident"] == 0
    assert got[0] == want[0] and bits_equal(np.array(got[1]), np.array(want[1]))


def test_eight_percentiles_in_one_resident_call_equal_the_oracle(gd, monkeypatch):
    """The most percentiles the resident route's state holds, in one call: one workgroup lays eight grids (a wave each)
    over its sorted share of the subsample, one pass counts the subsample into all eight, and the candidates of every
    percentile that landed in a bracket go through their own two launches (pc_ls_*).  Values are the oracle's exact order
    statistics (percentile.c:587-589 for the rank), with and without bounds, on coverage and read depth; on noise with NaNs
    and infinities the plain radix route's."""
    for k in ("GDSP_PERCENTILE_RESIDENT_OFF", "GDSP_PERCENTILE_CHAIN_OFF", "GDSP_PERCENTILE_LDS_SELECT", "GDSP_PERCENTILE_LDS_GIVEUP"):
        monkeypatch.delenv(k, raising=False)
    n = 2_600_003
    pts = [100, 2500, 25000, 50000, 75000, 97500, 99900, 99999]
    for name, x in signals(n, 11).items():
        vecs = [gd.DeviceVector.from_numpy(x[: n // 2]), gd.DeviceVector.from_numpy(x[n // 2:])]
        for kw in ({}, {"lo": 0.5, "hi": 55.0}):
            cnt, vals = gd.percentile(vecs, pts, **kw)
            st = gd.percentile_stats()
            if name.startswith("noise"):
                # NaNs in the population: the reference's qsort comparator (genodsp.c:2262-2270) is not an order on them and
                # the oracle's values are whatever its sort leaves; the plain radix route on the keys' order is the yardstick
                # (and a NaN among the pivots sends the call the old way: no claim on the route)
                wcnt, wvals = gd.percentile(vecs, pts, strategy=gd.SELECT_RADIX, **kw)
            else:
                wcnt, wvals = cpu.percentile([x[: n // 2], x[n // 2:]], pts, **kw)
                assert st["resident"] == 1, (name, kw, st)
            assert cnt == wcnt and bits_equal(np.array(vals), np.array(wvals)), (name, kw, vals, wvals)


def test_infinities_and_signed_zeros_without_nans_are_held_to_the_oracle_on_every_route(gd, monkeypatch):
    """A NaN-free population on which the reference's comparator (genodsp.c:2262-2270) IS an order: +-inf (inside the
    population only when the bounds let them in: the defaults are -DBL_MAX / DBL_MAX, percentile.c:611-651 skips what lies
    outside), +0 and -0 (equal under that order: a zero is compared as a value, its sign is the sort's leftover and not
    pinned) and negative values.  Every route of a one-device call -- resident, resident with a launch per digit, the LDS
    select giving up, chained, a read-back per digit -- gives the oracle's population and order statistics, and the fused
    binarize the oracle's binarize at that value."""
    n = 2_100_001
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n) * 5.0
    x[3::1013] = np.inf
    x[7::1019] = -np.inf
    x[11::29] = 0.0
    x[13::31] = -0.0
    assert not np.isnan(x).any()
    parts = [x[: n // 2 + 3], x[n // 2 + 3:]]
    vecs = [gd.DeviceVector.from_numpy(p) for p in parts]
    pts = [30, 10000, 50000, 90000, 99950]           # -inf, a negative value, a zero, a positive value, +inf (bounds open)
    for kw in ({"lo": -np.inf, "hi": np.inf}, {}, {"lo": -np.inf, "hi": 2.5}, {"lo": 0.0, "hi": np.inf}, {"lo": -np.inf, "hi": np.inf, "window": 3}):
        wcnt, wvals = cpu.percentile(parts, pts, **kw)
        if kw.get("lo") == -np.inf and kw.get("hi") == np.inf and "window" not in kw:
            assert wvals[0] == -np.inf and wvals[2] == 0.0 and wvals[4] == np.inf
        for route, env in ROUTES.items():
            for k in ("GDSP_PERCENTILE_RESIDENT_OFF", "GDSP_PERCENTILE_CHAIN_OFF", "GDSP_PERCENTILE_LDS_SELECT", "GDSP_PERCENTILE_LDS_GIVEUP"):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            cnt, vals = gd.percentile(vecs, pts, **kw)
            assert cnt == wcnt, (route, kw)
            for got, want in zip(vals, wvals):
                assert got == want and (want == 0.0 or bits_equal(np.array([got]), np.array([want]))), (route, kw, vals, wvals)
            for which in (0, 2, 4):
                c2, v2, outs, one_pass = gd.percentile_binarize(vecs, pts, which=which, **kw)
                assert c2 == wcnt and all(a == b for a, b in zip(v2, wvals)), (route, kw)
                for p, o in zip(parts, outs):
                    assert bits_equal(o.numpy(), cpu.binarize(p, wvals[which], False, 1.0, 0.0)), (route, kw, which)
