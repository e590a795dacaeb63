"""CPU: the C-ABI library loads and exports every symbol include/genodsp_hip.h declares
(no compute calls: there is no GPU here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "genodsp_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(gdsp_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def built_lib():
    import genodsp_amd
    if not os.path.exists(genodsp_amd.SO_PATH):
        genodsp_amd.build()
    return ctypes.CDLL(genodsp_amd.SO_PATH)


def test_header_declares_the_hot_path():
    syms = declared_symbols()
    for must in ("gdsp_smooth", "gdsp_fir_apply", "gdsp_local_extrema", "gdsp_best_extrema", "gdsp_dilate",
                 "gdsp_erode", "gdsp_close", "gdsp_open", "gdsp_binarize", "gdsp_clip", "gdsp_erase",
                 "gdsp_add_constant", "gdsp_abs", "gdsp_invert", "gdsp_sliding_sum", "gdsp_window_sum",
                 "gdsp_cumulative_sum", "gdsp_select_histogram", "gdsp_apply_intervals",
                 "gdsp_scale_intervals", "gdsp_report_runs"):
        assert must in syms


def test_library_exports_every_declared_symbol(built_lib):
    missing = [s for s in declared_symbols() if not hasattr(built_lib, s)]
    assert not missing, missing


def test_python_binding_covers_the_header():
    import genodsp_amd
    assert sorted(genodsp_amd.SIGNATURES) == declared_symbols()


def test_host_side_helpers_run_without_a_gpu(built_lib):
    """Entry points that are pure host code can be exercised here."""
    import numpy as np
    import genodsp_amd as gd
    from oracle import cpu
    for W in (3, 5, 101, 1001):
        assert gd.hann_taps(W).tobytes() == cpu.hann_window(W).tobytes()
    L = gd.lib()
    for x in (0.0, -0.0, 1.5, -1.5, 1e300, -1e300, 5e-324):
        assert L.gdsp_key_to_double(L.gdsp_double_to_key(x)) == x
    keys = [L.gdsp_double_to_key(x) for x in (-1e9, -2.0, -1e-300, 0.0, 1e-300, 2.0, 1e9)]
    assert keys == sorted(keys)
    assert L.gdsp_double_to_key(-0.0) == L.gdsp_double_to_key(0.0)
    # rank formula, percentile.c:587-589
    assert L.gdsp_percentile_rank(1000, 99000) == 990
    assert L.gdsp_percentile_rank(1000, 100000) == 999
    assert L.gdsp_percentile_rank(7, 50000) == 3
    # interval binning keeps file order inside a tile
    tile = L.gdsp_interval_tile()
    start = np.array([10, 5, tile - 3, 2 * tile + 1, 7], np.uint32)
    end = np.array([20, 9, tile + 4, 2 * tile + 2, 8], np.uint32)
    n = 3 * tile
    off = np.zeros(4, np.uint32)
    ln = ctypes.c_uint64()
    vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)
    gd.call("gdsp_bin_intervals", n, vp(start), vp(end), 5, vp(off), None, ctypes.byref(ln))
    assert ln.value == 6
    lst = np.zeros(6, np.uint32)
    gd.call("gdsp_bin_intervals", n, vp(start), vp(end), 5, vp(off), vp(lst), ctypes.byref(ln))
    assert off.tolist() == [0, 4, 5, 6]
    assert lst.tolist() == [0, 1, 2, 4, 2, 3]


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    import genodsp_amd._lib as L
    monkeypatch.setattr(L, "_lib", None)
    monkeypatch.setattr(L, "SO_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(L.GdspError):
        L.lib()


def test_driver_output_formatter_prints_what_printf_prints():
    """genodsp_hip formats its output lines by hand (the report is bound by fprintf otherwise); the function is
    pulled out of the driver's source and held to snprintf("%.*f") on 2 M values: integers, decimal-looking
    values near ties, dyadic fractions, random bit patterns, precisions 0..9."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    p = subprocess.run(["bash", "tools/check_output_format.sh", "2000000"], cwd=root, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    assert p.stdout.strip().splitlines()[-1] == "bad=0", p.stdout[-2000:]
