"""ctypes loader for genodsp_amd/libgenodsp_hip.so (the C ABI in include/genodsp_hip.h).

There is no CPU fallback: if the shared library is missing, or a call fails, this
raises.  `build()` compiles it in-tree with hipcc for gfx950.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(_HERE, "libgenodsp_hip.so")
CSRC = os.path.join(_HERE, "csrc")

_lib = None


class GdspError(RuntimeError):
    pass


def build(jobs=8):
    """Compile every HIP source for gfx950 into genodsp_amd/libgenodsp_hip.so."""
    subprocess.check_call(["make", "-s", "-j", str(jobs), "-C", CSRC])
    if not os.path.exists(SO_PATH):
        raise GdspError("build did not produce " + SO_PATH)


_vp, _u32, _u64, _f64, _int, _sz = C.c_void_p, C.c_uint32, C.c_uint64, C.c_double, C.c_int, C.c_size_t
_pf = C.POINTER(C.c_float)

# name -> (restype, argtypes); mirrors include/genodsp_hip.h one to one
SIGNATURES = {
    "gdsp_last_error": (C.c_char_p, []),
    "gdsp_version": (C.c_char_p, []),
    "gdsp_poison": (_int, [C.POINTER(_f64)]),
    "gdsp_device_count": (_int, [C.POINTER(_int)]),
    "gdsp_get_device": (_int, [_vp]),
    "gdsp_set_device": (_int, [_int]),
    "gdsp_malloc": (_int, [C.POINTER(_vp), _sz]),
    "gdsp_free": (_int, [_vp]),
    "gdsp_host_alloc": (_int, [C.POINTER(_vp), _sz]),
    "gdsp_host_free": (_int, [_vp]),
    "gdsp_memcpy_h2d": (_int, [_vp, _vp, _sz, _vp]),
    "gdsp_memcpy_d2h": (_int, [_vp, _vp, _sz, _vp]),
    "gdsp_memcpy_d2d": (_int, [_vp, _vp, _sz, _vp]),
    "gdsp_memcpy_peer": (_int, [_vp, _int, _vp, _int, _sz, _vp]),
    "gdsp_memset": (_int, [_vp, _int, _sz, _vp]),
    "gdsp_stream_create": (_int, [C.POINTER(_vp)]),
    "gdsp_stream_destroy": (_int, [_vp]),
    "gdsp_stream_sync": (_int, [_vp]),
    "gdsp_device_sync": (_int, []),
    "gdsp_event_create": (_int, [C.POINTER(_vp)]),
    "gdsp_event_destroy": (_int, [_vp]),
    "gdsp_event_record": (_int, [_vp, _vp]),
    "gdsp_stream_wait_event": (_int, [_vp, _vp]),
    "gdsp_event_elapsed_ms": (_int, [_vp, _vp, _pf]),
    "gdsp_fill": (_int, [_vp, _u32, _f64, _vp]),
    "gdsp_hann_taps": (_int, [_u32, _vp]),
    "gdsp_fir_plan_create": (_int, [C.POINTER(_vp), _vp, _u32]),
    "gdsp_fir_plan_destroy": (_int, [_vp]),
    "gdsp_fir_apply": (_int, [_vp, _vp, _vp, _u32, _int, _vp]),
    "gdsp_smooth": (_int, [_vp, _vp, _u32, _u32, _int, _vp]),
    "gdsp_smooth_local_extrema_fusable": (_int, [_u32, _u32]),
    "gdsp_smooth_local_extrema": (_int, [_vp, _vp, _u32, _u32, _int, _u32, _int, _f64, _vp]),
    "gdsp_sliding_sum": (_int, [_vp, _vp, _u32, _u32, _f64, _vp]),
    "gdsp_window_sum": (_int, [_vp, _u32, _u32, _f64, _int, _f64, _vp]),
    "gdsp_cumulative_sum_work": (_sz, [_u32]),
    "gdsp_cumulative_sum": (_int, [_vp, _u32, _vp, _vp]),
    "gdsp_local_extrema": (_int, [_vp, _vp, _u32, _u32, _int, _f64, _vp]),
    "gdsp_best_extrema": (_int, [_vp, _vp, _u32, _u32, _int, _vp]),
    "gdsp_long_window_work": (_sz, [_u32]),
    "gdsp_best_extrema_any": (_int, [_vp, _vp, _u32, _u32, _int, _vp, _sz, _vp]),
    "gdsp_local_extrema_any": (_int, [_vp, _vp, _u32, _u32, _int, _f64, _vp, _sz, _vp]),
    "gdsp_sliding_sum_any": (_int, [_vp, _vp, _u32, _u32, _f64, _vp, _sz, _vp]),
    "gdsp_dilate": (_int, [_vp, _vp, _u32, _u32, _u32, _f64, _f64, _f64, _vp]),
    "gdsp_erode": (_int, [_vp, _vp, _u32, _u32, _u32, _f64, _f64, _f64, _vp]),
    "gdsp_dilate_erode": (_int, [_vp, _vp, _u32, _u32, _u32, _f64, _f64, _f64, _u32, _u32, _f64, _f64, _f64,
                                 _int, _f64, _int, _f64, _f64, _vp]),
    "gdsp_dilate_erode_fusable": (_int, [_u32, _u32, _u32, _u32]),
    "gdsp_close": (_int, [_vp, _vp, _u32, _f64, _f64, _f64, _f64, _vp]),
    "gdsp_open": (_int, [_vp, _vp, _u32, _f64, _f64, _f64, _f64, _vp]),
    "gdsp_dilate_any": (_int, [_vp, _vp, _u32, _u32, _u32, _f64, _f64, _f64, _vp, _sz, _vp]),
    "gdsp_erode_any": (_int, [_vp, _vp, _u32, _u32, _u32, _f64, _f64, _f64, _vp, _sz, _vp]),
    "gdsp_close_any": (_int, [_vp, _vp, _u32, _f64, _f64, _f64, _f64, _vp, _sz, _vp]),
    "gdsp_open_any": (_int, [_vp, _vp, _u32, _f64, _f64, _f64, _f64, _vp, _sz, _vp]),
    "gdsp_binarize": (_int, [_vp, _u32, _f64, _int, _f64, _f64, _vp]),
    "gdsp_clip": (_int, [_vp, _u32, _int, _f64, _int, _f64, _vp]),
    "gdsp_erase": (_int, [_vp, _u32, _int, _f64, _int, _f64, _int, _f64, _vp]),
    "gdsp_add_constant": (_int, [_vp, _u32, _f64, _vp]),
    "gdsp_abs": (_int, [_vp, _u32, _vp]),
    "gdsp_invert": (_int, [_vp, _u32, _f64, _vp]),
    "gdsp_map": (_int, [_vp, _u32, _vp, _vp, _u32, _vp]),
    "gdsp_clump_work": (_sz, [_u32]),
    "gdsp_clump": (_int, [_vp, _u32, _f64, _u32, _int, _f64, _f64, _vp, _vp]),
    "gdsp_minmax_init": (_int, [_vp, _vp]),
    "gdsp_minmax_update": (_int, [_vp, _u32, _u32, _f64, _f64, _vp, _vp]),
    "gdsp_select_hist_init": (_int, [_vp, _int, _vp]),
    "gdsp_select_histogram": (_int, [_vp, _u32, _u32, _f64, _f64, _int, _int, _u64, _vp, _vp]),
    "gdsp_select_pick": (_int, [_vp, _int, _u64, C.POINTER(_u32), C.POINTER(_u64)]),
    "gdsp_key_to_double": (_f64, [_u64]),
    "gdsp_double_to_key": (_u64, [_f64]),
    "gdsp_percentile_rank": (_u32, [_u32, _u32]),
    "gdsp_percentiles": (_int, [_vp, _int, _u32, _f64, _f64, _vp, _int, _int, _u32, _vp, _vp, _vp, _vp]),
    "gdsp_percentiles_binarize": (_int, [_vp, _int, _u32, _f64, _f64, _vp, _int, _int, _u32, _vp, _vp, _vp, _vp, _vp, C.POINTER(_int)]),
    "gdsp_percentiles_stats": (None, [_vp]),
    "gdsp_comm_create": (_int, [C.POINTER(_vp), C.POINTER(_int), _int]),
    "gdsp_comm_destroy": (_int, [_vp]),
    "gdsp_comm_size": (_int, [_vp]),
    "gdsp_comm_device": (_int, [_vp, _int]),
    "gdsp_comm_rccl_version": (_int, [C.POINTER(_int)]),
    "gdsp_comm_allreduce_u64": (_int, [_vp, C.POINTER(_vp), _sz, _int, C.POINTER(_vp)]),
    "gdsp_comm_allreduce_f64": (_int, [_vp, C.POINTER(_vp), _sz, _int, C.POINTER(_vp)]),
    "gdsp_percentiles_use_comm": (_int, [_vp]),
    "gdsp_percentiles_use_device_reduce": (_int, [_vp, _vp]),
    "gdsp_interval_tile": (_u32, []),
    "gdsp_bin_intervals": (_int, [_u32, _vp, _vp, _u32, _vp, _vp, C.POINTER(_u64)]),
    "gdsp_apply_intervals": (_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _int, _int, _f64, _vp]),
    "gdsp_scale_intervals": (_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _int, _f64, _vp]),
    "gdsp_mask_intervals": (_int, [_vp, _u32, _vp, _vp, _vp, _vp, _vp, _int, _f64, _int, _vp]),
    "gdsp_extreme_in_intervals_work": (_sz, [_u32]),
    "gdsp_extreme_in_intervals": (_int, [_vp, _u32, _vp, _vp, _u32, _vp, _vp, _int, _f64, _vp, _vp]),
    "gdsp_report_runs_work": (_sz, [_u32]),
    "gdsp_report_runs": (_int, [_vp, _u32, _int, _int, _vp, _vp, _vp, _u32, _vp, _vp, _vp]),
    "gdsp_synth_coverage": (_int, [_vp, _u64, _u32, _u32, _u32, _int, _vp]),
    # one launch per operator per device: (items, nitems, <the single-vector call's parameters>, stream)
    "gdsp_smooth_batch": (_int, [_vp, _int, _u32, _int, _vp]),
    "gdsp_smooth_local_extrema_batch": (_int, [_vp, _int, _u32, _int, _u32, _int, _f64, _vp]),
    "gdsp_local_extrema_batch": (_int, [_vp, _int, _u32, _int, _f64, _vp]),
    "gdsp_best_extrema_batch": (_int, [_vp, _int, _u32, _int, _vp]),
    "gdsp_dilate_batch": (_int, [_vp, _int, _u32, _u32, _f64, _f64, _f64, _vp]),
    "gdsp_erode_batch": (_int, [_vp, _int, _u32, _u32, _f64, _f64, _f64, _vp]),
    "gdsp_dilate_erode_batch": (_int, [_vp, _int, _u32, _u32, _f64, _f64, _f64, _u32, _u32, _f64, _f64, _f64,
                                       _int, _f64, _int, _f64, _f64, _vp]),
    "gdsp_binarize_batch": (_int, [_vp, _int, _f64, _int, _f64, _f64, _vp]),
    "gdsp_clip_batch": (_int, [_vp, _int, _int, _f64, _int, _f64, _vp]),
    "gdsp_erase_batch": (_int, [_vp, _int, _int, _f64, _int, _f64, _int, _f64, _vp]),
    "gdsp_add_constant_batch": (_int, [_vp, _int, _f64, _vp]),
    "gdsp_abs_batch": (_int, [_vp, _int, _vp]),
}

# functions whose int return is a status code
_STATUS = {k for k, (r, _) in SIGNATURES.items() if r is _int} - {"gdsp_smooth_local_extrema_fusable", "gdsp_dilate_erode_fusable", "gdsp_comm_size",
                                                                  "gdsp_comm_device"}


def lib():
    """The loaded library.  Raises GdspError when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(SO_PATH):
            raise GdspError(
                "genodsp_amd/libgenodsp_hip.so is missing: the HIP extension is the only "
                "compute path (no CPU fallback). Build it with genodsp_amd.build() or "
                "`make -C genodsp_amd/csrc`.")
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)          # AttributeError here = header and library disagree
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def call(name, *args):
    """Invoke a status-returning entry point; raise on a non-zero status."""
    L = lib()
    rc = getattr(L, name)(*args)
    if name in _STATUS and rc != 0:
        raise GdspError("%s failed (%d): %s" % (name, rc, L.gdsp_last_error().decode()))
    return rc
