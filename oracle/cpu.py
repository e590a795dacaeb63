"""ctypes binding of oracle/liboracle.so (TEST INFRASTRUCTURE ONLY).

Every function takes/returns numpy float64 arrays and mirrors one orc_* entry of
oracle/gdsp_oracle.h.  In-place reference ops return a modified copy.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "liboracle.so")

OVERLAP_SUM, OVERLAP_MIN, OVERLAP_MAX = 0, 1, 2
DBL_MAX = float(np.finfo(np.float64).max)

_lib = None


def build():
    """Compile the restatement (gcc, a second or two)."""
    subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])


def lib():
    global _lib
    if _lib is None:
        src = os.path.join(_HERE, "gdsp_oracle.c")
        if (not os.path.exists(_SO)) or os.path.getmtime(_SO) < os.path.getmtime(src):
            build()
        _lib = C.CDLL(_SO)
        _declare(_lib)
    return _lib


_pd = C.POINTER(C.c_double)
_pu = C.POINTER(C.c_uint32)
_u32, _f64, _int, _u64 = C.c_uint32, C.c_double, C.c_int, C.c_uint64


def _declare(L):
    def sig(name, res, *args):
        f = getattr(L, name)
        f.restype = res
        f.argtypes = list(args)

    sig("orc_hann_window", None, _u32, _pd)
    sig("orc_fir", None, _pd, _u32, _pd, _u32, _pd)
    sig("orc_smooth", None, _pd, _u32, _u32, _pd)
    sig("orc_sliding_sum", None, _pd, _u32, _u32, _f64, _pd)
    sig("orc_window_sum", None, _pd, _u32, _u32, _f64, _int, _f64)
    sig("orc_cumulative_sum", None, _pd, _u32)
    sig("orc_local_extrema", None, _pd, _u32, _u32, _int, _f64, _pd)
    sig("orc_best_extrema", None, _pd, _u32, _u32, _int, _pd)
    for nm in ("orc_dilate", "orc_erode"):
        sig(nm, None, _pd, _u32, _u32, _u32, _f64, _f64, _f64)
    for nm in ("orc_close", "orc_open"):
        sig(nm, None, _pd, _u32, _f64, _f64, _f64, _f64)
    sig("orc_binarize", None, _pd, _u32, _f64, _int, _f64, _f64)
    sig("orc_clip", None, _pd, _u32, _int, _f64, _int, _f64)
    sig("orc_erase", None, _pd, _u32, _int, _f64, _int, _f64, _int, _f64)
    sig("orc_add_constant", None, _pd, _u32, _f64)
    sig("orc_abs", None, _pd, _u32)
    sig("orc_genome_minmax", None, C.POINTER(_pd), _pu, _int, _pd, _pd)
    sig("orc_invert", None, _pd, _u32, _f64)
    sig("orc_map", None, _pd, _u32, _pd, _pd, _u32)
    sig("orc_clump", None, _pd, _u32, _f64, _u32, _int, _f64, _f64)
    sig("orc_percentile", _u32, C.POINTER(_pd), _pu, _int, _u32, _f64, _f64, _pu, _int, _pd)
    sig("orc_fill", None, _pd, _u32, _f64)
    sig("orc_apply_intervals", None, _pd, _u32, _pu, _pu, _pd, _u32, _int, _int, _f64)
    sig("orc_scale_intervals", None, _pd, _u32, _pu, _pu, _pd, _u32, _int, _f64)
    sig("orc_mask_intervals", None, _pd, _u32, _pu, _pu, _pd, _u32, _int, _f64, _int)
    sig("orc_extreme_in_intervals", None, _pd, _u32, _pu, _pu, _u32, _int, _f64)
    sig("orc_report_runs", _u32, _pd, _u32, _int, _int, _pu, _pu, _pd, _u32)
    sig("orc_synth_coverage", None, _u64, _u32, _u32, _u32, _int, _pd)


def _in(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_pd)


def _copy(a):
    a = np.array(a, dtype=np.float64, order="C", copy=True)
    return a, a.ctypes.data_as(_pd)


def _u(a):
    a = np.ascontiguousarray(a, dtype=np.uint32)
    return a, a.ctypes.data_as(_pu)


def hann_window(W):
    w = np.empty(W, np.float64)
    lib().orc_hann_window(W, w.ctypes.data_as(_pd))
    return w


def fir(v, w):
    v, pv = _in(v)
    w, pw = _in(w)
    out = np.empty_like(v)
    lib().orc_fir(pv, v.size, pw, w.size, out.ctypes.data_as(_pd))
    return out


def smooth(v, W):
    v, pv = _in(v)
    out = np.empty_like(v)
    lib().orc_smooth(pv, v.size, W, out.ctypes.data_as(_pd))
    return out


def smooth_threads(vecs, W, threads, outs=None):
    """orc_smooth of every vector over `threads` pthreads inside the library; outs: preallocated (and touched) arrays"""
    vecs = [np.ascontiguousarray(v, np.float64) for v in vecs]
    outs = outs if outs is not None else [np.zeros_like(v) for v in vecs]
    n = len(vecs)
    pv = (C.c_void_p * n)(*[v.ctypes.data for v in vecs])
    po = (C.c_void_p * n)(*[o.ctypes.data for o in outs])
    lens = (C.c_uint32 * n)(*[v.size for v in vecs])
    L = lib()
    L.orc_smooth_threads.restype = C.c_int
    L.orc_smooth_threads.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_uint32, C.c_int]
    started = L.orc_smooth_threads(pv, lens, po, n, W, threads)
    return outs, started


def sliding_sum(v, W, denom=1.0):
    v, pv = _in(v)
    out = np.empty_like(v)
    lib().orc_sliding_sum(pv, v.size, W, denom, out.ctypes.data_as(_pd))
    return out


def window_sum(v, W, denom=1.0, use_actual=False, zero=0.0):
    v, pv = _copy(v)
    lib().orc_window_sum(pv, v.size, W, denom, int(use_actual), zero)
    return v


def cumulative_sum(v):
    v, pv = _copy(v)
    lib().orc_cumulative_sum(pv, v.size)
    return v


def local_extrema(v, N, want_max, fill):
    v, pv = _in(v)
    out = np.empty_like(v)
    lib().orc_local_extrema(pv, v.size, N, int(want_max), fill, out.ctypes.data_as(_pd))
    return out


def best_extrema(v, W, want_max):
    v, pv = _in(v)
    out = np.empty_like(v)
    lib().orc_best_extrema(pv, v.size, W, int(want_max), out.ctypes.data_as(_pd))
    return out


def dilate(v, left, right, T=0.0, one=1.0, zero=0.0):
    v, pv = _copy(v)
    lib().orc_dilate(pv, v.size, left, right, T, one, zero)
    return v


def erode(v, left, right, T=0.0, one=1.0, zero=0.0):
    v, pv = _copy(v)
    lib().orc_erode(pv, v.size, left, right, T, one, zero)
    return v


def close(v, length, T=0.0, one=1.0, zero=0.0):
    v, pv = _copy(v)
    lib().orc_close(pv, v.size, float(length), T, one, zero)
    return v


def open_(v, length, T=0.0, one=1.0, zero=0.0):
    v, pv = _copy(v)
    lib().orc_open(pv, v.size, float(length), T, one, zero)
    return v


def binarize(v, T=0.0, ties_above=False, one=1.0, zero=0.0):
    v, pv = _copy(v)
    lib().orc_binarize(pv, v.size, T, int(ties_above), one, zero)
    return v


def clip(v, lo=None, hi=None):
    v, pv = _copy(v)
    lib().orc_clip(pv, v.size, lo is not None, 0.0 if lo is None else lo,
                   hi is not None, 0.0 if hi is None else hi)
    return v


def erase(v, lo=None, hi=None, keep_inside=False, zero=0.0):
    v, pv = _copy(v)
    lib().orc_erase(pv, v.size, lo is not None, 0.0 if lo is None else lo,
                    hi is not None, 0.0 if hi is None else hi, int(keep_inside), zero)
    return v


def add_constant(v, c):
    v, pv = _copy(v)
    lib().orc_add_constant(pv, v.size, c)
    return v


def abs_(v):
    v, pv = _copy(v)
    lib().orc_abs(pv, v.size)
    return v


def _genome(vecs):
    vecs = [np.ascontiguousarray(x, dtype=np.float64) for x in vecs]
    ptrs = (_pd * len(vecs))(*[x.ctypes.data_as(_pd) for x in vecs])
    lens = np.array([x.size for x in vecs], np.uint32)
    return vecs, ptrs, lens


def genome_minmax(vecs):
    vecs, ptrs, lens = _genome(vecs)
    lo, hi = C.c_double(), C.c_double()
    lib().orc_genome_minmax(ptrs, lens.ctypes.data_as(_pu), len(vecs), C.byref(lo), C.byref(hi))
    return lo.value, hi.value


def invert(v, mid):
    v, pv = _copy(v)
    lib().orc_invert(pv, v.size, mid)
    return v


def map_values(v, knots_in, knots_out):
    v, pv = _copy(v)
    a, pa = _in(knots_in)
    b, pb = _in(knots_out)
    lib().orc_map(pv, v.size, pa, pb, a.size)
    return v


def clump(v, average, min_length, above=True, one=1.0, zero=0.0):
    v, pv = _copy(v)
    lib().orc_clump(pv, v.size, average, min_length, 1 if above else 0, one, zero)
    return v


def percentile(vecs, p_thousandths, window=1, lo=-DBL_MAX, hi=DBL_MAX):
    """vecs in the reference's processing order.  Returns (count, values)."""
    vecs, ptrs, lens = _genome(vecs)
    pt, ppt = _u(p_thousandths)
    out = np.zeros(pt.size, np.float64)
    count = lib().orc_percentile(ptrs, lens.ctypes.data_as(_pu), len(vecs), window, lo, hi,
                                 ppt, pt.size, out.ctypes.data_as(_pd))
    return count, out


def apply_intervals(v, start, end, val, overlap=OVERLAP_SUM, clear=False, missing=0.0):
    v, pv = _copy(v)
    s, ps = _u(start)
    e, pe = _u(end)
    x, px = _in(val)
    lib().orc_apply_intervals(pv, v.size, ps, pe, px, s.size, overlap, int(clear), missing)
    return v


def scale_intervals(v, start, end, val, divide=False, infinity=DBL_MAX):
    v, pv = _copy(v)
    s, ps = _u(start)
    e, pe = _u(end)
    x, px = _in(val)
    lib().orc_scale_intervals(pv, v.size, ps, pe, px, s.size, int(divide), infinity)
    return v


def mask_intervals(v, start, end, val, inside=True, outside_val=0.0, binarize_first=False):
    v, pv = _copy(v)
    s, ps = _u(start)
    e, pe = _u(end)
    x, px = _in(val)
    lib().orc_mask_intervals(pv, v.size, ps, pe, px, s.size, int(inside), outside_val, int(binarize_first))
    return v


def extreme_in_intervals(v, start, end, want_max, fill):
    v, pv = _copy(v)
    s, ps = _u(start)
    e, pe = _u(end)
    lib().orc_extreme_in_intervals(pv, v.size, ps, pe, s.size, int(want_max), fill)
    return v


def report_runs(v, collapse=True, uncovered=0):
    v, pv = _in(v)
    cap = v.size + 1
    s = np.empty(cap, np.uint32)
    e = np.empty(cap, np.uint32)
    x = np.empty(cap, np.float64)
    n = lib().orc_report_runs(pv, v.size, int(collapse), uncovered, s.ctypes.data_as(_pu),
                              e.ctypes.data_as(_pu), x.ctypes.data_as(_pd), cap)
    return s[:n].copy(), e[:n].copy(), x[:n].copy()


def synth_coverage(seed, chrom_index, start, count, mode=0):
    out = np.empty(count, np.float64)
    lib().orc_synth_coverage(seed, chrom_index, start, count, mode, out.ctypes.data_as(_pd))
    return out
