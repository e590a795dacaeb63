"""genodsp_amd -- MI355X-native genomic-signal DSP (genodsp-compatible hot path).

This package is the thin Python face of the C ABI in include/genodsp_hip.h
(genodsp_amd/libgenodsp_hip.so: hand-written HIP kernels for gfx950).  It exists
for tests and bench.py; the drop-in product is the C library plus the C host
driver under genodsp_amd/host/.  There is no CPU fallback anywhere in here.

    import genodsp_amd as gd
    v = gd.DeviceVector.from_numpy(x)        # f64 chromosome vector in HBM
    out = gd.smooth(v, 101)                  # op_smooth_apply, sum.c:616-676
    out.numpy()
"""
import ctypes as C

import numpy as np

from ._lib import GdspError, SIGNATURES, SO_PATH, build, call, lib

FIR_EXACT, FIR_FMA, FIR_HANN = 0, 1, 2
OVERLAP_SUM, OVERLAP_MIN, OVERLAP_MAX = 0, 1, 2
DBL_MAX = float(np.finfo(np.float64).max)
DBL_MIN = float(np.finfo(np.float64).tiny)


def device_count():
    n = C.c_int(0)
    call("gdsp_device_count", C.byref(n))
    return n.value


def current_device():
    d = C.c_int(0)
    call("gdsp_get_device", C.byref(d))
    return d.value


def set_device(i):
    call("gdsp_set_device", int(i))


def _sp(stream):
    return C.c_void_p(stream) if stream else None


class Stream:
    def __init__(self):
        h = C.c_void_p()
        call("gdsp_stream_create", C.byref(h))
        self.handle = h.value

    def sync(self):
        call("gdsp_stream_sync", C.c_void_p(self.handle))

    def wait_event(self, event):
        """what is queued on this stream from now on starts after `event`"""
        call("gdsp_stream_wait_event", C.c_void_p(self.handle), C.c_void_p(event.handle))

    def close(self):
        if self.handle:
            call("gdsp_stream_destroy", C.c_void_p(self.handle))
            self.handle = None


def sync(stream=None):
    """Wait for `stream`; with no stream, for EVERY stream of the current device (the library's streams are
    non-blocking, so the NULL stream alone would not order a copy behind a kernel launched on one of them)."""
    if stream:
        call("gdsp_stream_sync", _sp(stream))
    else:
        call("gdsp_device_sync")


class Event:
    def __init__(self):
        h = C.c_void_p()
        call("gdsp_event_create", C.byref(h))
        self.handle = h.value

    def record(self, stream=None):
        call("gdsp_event_record", C.c_void_p(self.handle), _sp(stream))

    def elapsed_ms(self, later):
        ms = C.c_float()
        call("gdsp_event_elapsed_ms", C.c_void_p(self.handle), C.c_void_p(later.handle), C.byref(ms))
        return ms.value


class DeviceBuffer:
    """Raw HBM allocation (hipMalloc), 256-byte aligned."""

    def __init__(self, nbytes):
        p = C.c_void_p()
        call("gdsp_malloc", C.byref(p), int(nbytes))
        self.ptr = p.value
        self.nbytes = int(nbytes)

    def free(self):
        if self.ptr:
            call("gdsp_free", C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass

    def upload(self, arr, offset=0, stream=None):
        arr = np.ascontiguousarray(arr)
        assert offset + arr.nbytes <= self.nbytes
        sync(stream)                     # nothing launched earlier, on whatever stream, may still be using the bytes
        call("gdsp_memcpy_h2d", C.c_void_p(self.ptr + offset), arr.ctypes.data_as(C.c_void_p), arr.nbytes, _sp(stream))
        sync(stream)

    def download(self, dtype, count, offset=0, stream=None):
        out = np.empty(count, dtype)
        assert offset + out.nbytes <= self.nbytes
        sync(stream)                     # producers on other (non-blocking) streams included when stream is None
        call("gdsp_memcpy_d2h", out.ctypes.data_as(C.c_void_p), C.c_void_p(self.ptr + offset), out.nbytes, _sp(stream))
        sync(stream)
        return out


class DeviceVector:
    """A chromosome's worth of f64 values in HBM (the reference's spec.valVector)."""

    def __init__(self, n, buf=None, offset=0):
        self.n = int(n)
        self.buf = buf if buf is not None else DeviceBuffer(max(self.n, 2) * 8)
        self.offset = offset
        assert offset % 16 == 0

    @property
    def ptr(self):
        return C.c_void_p(self.buf.ptr + self.offset)

    @classmethod
    def from_numpy(cls, arr):
        arr = np.ascontiguousarray(arr, dtype=np.float64)
        v = cls(arr.size)
        if arr.size:
            v.buf.upload(arr, v.offset)
        return v

    def numpy(self):
        return self.buf.download(np.float64, self.n, self.offset)

    def like(self):
        return DeviceVector(self.n)

    def copy(self, stream=None):
        out = self.like()
        if self.n:
            sync(stream)
            call("gdsp_memcpy_d2d", out.ptr, self.ptr, self.n * 8, _sp(stream))
        return out


def _dev_array(arr, dtype):
    arr = np.ascontiguousarray(arr, dtype=dtype)
    buf = DeviceBuffer(max(arr.nbytes, 16))
    if arr.nbytes:
        buf.upload(arr)
    return buf


# ----------------------------------------------------------------- sum.c ----

def hann_taps(W):
    w = np.empty(W, np.float64)
    call("gdsp_hann_taps", W, w.ctypes.data_as(C.c_void_p))
    return w


class FirPlan:
    def __init__(self, taps):
        taps = np.ascontiguousarray(taps, dtype=np.float64)
        h = C.c_void_p()
        call("gdsp_fir_plan_create", C.byref(h), taps.ctypes.data_as(C.c_void_p), taps.size)
        self.handle = h.value
        self.W = taps.size

    def apply(self, v, out=None, mode=FIR_EXACT, stream=None):
        out = out if out is not None else v.like()
        call("gdsp_fir_apply", C.c_void_p(self.handle), v.ptr, out.ptr, v.n, mode, _sp(stream))
        return out

    def close(self):
        if self.handle:
            call("gdsp_fir_plan_destroy", C.c_void_p(self.handle))
            self.handle = None


def smooth(v, W=101, out=None, mode=FIR_EXACT, stream=None):
    out = out if out is not None else v.like()
    call("gdsp_smooth", v.ptr, out.ptr, v.n, W, mode, _sp(stream))
    return out


def smooth_local_extrema(v, W, N, want_max, fill, out=None, mode=FIR_EXACT, stream=None):
    """`= smooth W = localmax|localmin N` fused (bit-identical to the two calls in sequence)."""
    out = out if out is not None else v.like()
    call("gdsp_smooth_local_extrema", v.ptr, out.ptr, v.n, W, mode, N, int(want_max), float(fill), _sp(stream))
    return out


def _long_work(v):
    nbytes = lib().gdsp_long_window_work(v.n)
    return DeviceBuffer(nbytes), nbytes


def sliding_sum(v, W, denom=1.0, out=None, stream=None):
    out = out if out is not None else v.like()
    if W <= 8192:
        call("gdsp_sliding_sum", v.ptr, out.ptr, v.n, W, float(denom), _sp(stream))
        return out
    work, nbytes = _long_work(v)
    call("gdsp_sliding_sum_any", v.ptr, out.ptr, v.n, W, float(denom), C.c_void_p(work.ptr), nbytes, _sp(stream))
    sync(stream)
    return out


def window_sum(v, W, denom=1.0, use_actual=False, zero=0.0, stream=None):
    call("gdsp_window_sum", v.ptr, v.n, W, float(denom), int(use_actual), float(zero), _sp(stream))
    return v


def cumulative_sum(v, stream=None):
    work = DeviceBuffer(lib().gdsp_cumulative_sum_work(v.n))
    call("gdsp_cumulative_sum", v.ptr, v.n, C.c_void_p(work.ptr), _sp(stream))
    sync(stream)
    return v


# -------------------------------------------------------------- minmax.c ----

def local_extrema(v, N, want_max, fill, out=None, stream=None):
    out = out if out is not None else v.like()
    if N <= 4096:
        call("gdsp_local_extrema", v.ptr, out.ptr, v.n, N, int(want_max), float(fill), _sp(stream))
        return out
    work, nbytes = _long_work(v)
    call("gdsp_local_extrema_any", v.ptr, out.ptr, v.n, N, int(want_max), float(fill), C.c_void_p(work.ptr), nbytes, _sp(stream))
    sync(stream)
    return out


def localmax(v, N=3, zero=0.0, **kw):
    return local_extrema(v, N, True, zero, **kw)


def localmin(v, N=3, infinity=DBL_MAX, **kw):
    return local_extrema(v, N, False, infinity, **kw)


def best_extrema(v, W, want_max, out=None, stream=None):
    out = out if out is not None else v.like()
    if W <= 4096:
        call("gdsp_best_extrema", v.ptr, out.ptr, v.n, W, int(want_max), _sp(stream))
        return out
    work, nbytes = _long_work(v)
    call("gdsp_best_extrema_any", v.ptr, out.ptr, v.n, W, int(want_max), C.c_void_p(work.ptr), nbytes, _sp(stream))
    sync(stream)
    return out


# ---------------------------------------------------------- morphology.c ----

def split_length(length):
    """dilate/erode <length> -> (left, right), morphology.c:917-920 / :1369-1372."""
    left = int(float(length) / 2)
    return left, int(length) - left


def dilate(v, left, right, T=0.0, one=1.0, zero=0.0, out=None, stream=None):
    out = out if out is not None else v.like()
    work, nbytes = _long_work(v) if left + right > 200000 else (None, 0)
    call("gdsp_dilate_any", v.ptr, out.ptr, v.n, left, right, float(T), float(one), float(zero),
         C.c_void_p(work.ptr) if work else None, nbytes, _sp(stream))
    if work:
        sync(stream)
    return out


def erode(v, left, right, T=0.0, one=1.0, zero=0.0, out=None, stream=None):
    out = out if out is not None else v.like()
    work, nbytes = _long_work(v) if left + right > 200000 else (None, 0)
    call("gdsp_erode_any", v.ptr, out.ptr, v.n, left, right, float(T), float(one), float(zero),
         C.c_void_p(work.ptr) if work else None, nbytes, _sp(stream))
    if work:
        sync(stream)
    return out


def dilate_erode(v, d_left, d_right, e_left, e_right, d_T=0.0, d_one=1.0, d_zero=0.0, e_T=0.0, e_one=1.0,
                 e_zero=0.0, binarize=None, out=None, stream=None):
    """`= dilate = erode [= binarize]` fused; binarize = (T, ties_above, one, zero) or None."""
    out = out if out is not None else v.like()
    b = binarize if binarize is not None else (0.0, False, 1.0, 0.0)
    call("gdsp_dilate_erode", v.ptr, out.ptr, v.n, d_left, d_right, float(d_T), float(d_one), float(d_zero),
         e_left, e_right, float(e_T), float(e_one), float(e_zero), int(binarize is not None),
         float(b[0]), int(b[1]), float(b[2]), float(b[3]), _sp(stream))
    return out


def close(v, length, T=0.0, one=1.0, zero=0.0, out=None, stream=None):
    out = out if out is not None else v.like()
    work, nbytes = _long_work(v) if length > 100000 else (None, 0)
    call("gdsp_close_any", v.ptr, out.ptr, v.n, float(length), float(T), float(one), float(zero),
         C.c_void_p(work.ptr) if work else None, nbytes, _sp(stream))
    if work:
        sync(stream)
    return out


def open_(v, length, T=0.0, one=1.0, zero=0.0, out=None, stream=None):
    out = out if out is not None else v.like()
    work, nbytes = _long_work(v) if length > 100000 else (None, 0)
    call("gdsp_open_any", v.ptr, out.ptr, v.n, float(length), float(T), float(one), float(zero),
         C.c_void_p(work.ptr) if work else None, nbytes, _sp(stream))
    if work:
        sync(stream)
    return out


def morph_bits(op, v, *args, **kw):
    """the workspace form of dilate / erode / close / open whatever the length (GDSP_MORPH_FORCE_BITS is read once per
    process by the library, so tests call the entry points with workspace and the variable set before the first call)"""
    out = v.like()
    work, nbytes = _long_work(v)
    a = [float(x) if isinstance(x, float) else x for x in args]
    call("gdsp_%s_any" % op, v.ptr, out.ptr, v.n, *a, C.c_void_p(work.ptr), nbytes, _sp(kw.get("stream")))
    sync(kw.get("stream"))
    return out


# --------------------------------------------- logical.c, mask.c, add.c ----

def binarize(v, T=0.0, ties_above=False, one=1.0, zero=0.0, stream=None):
    call("gdsp_binarize", v.ptr, v.n, float(T), int(ties_above), float(one), float(zero), _sp(stream))
    return v


def clip(v, lo=None, hi=None, stream=None):
    call("gdsp_clip", v.ptr, v.n, lo is not None, 0.0 if lo is None else float(lo),
         hi is not None, 0.0 if hi is None else float(hi), _sp(stream))
    return v


def erase(v, lo=None, hi=None, keep_inside=False, zero=0.0, stream=None):
    call("gdsp_erase", v.ptr, v.n, lo is not None, 0.0 if lo is None else float(lo),
         hi is not None, 0.0 if hi is None else float(hi), int(keep_inside), float(zero), _sp(stream))
    return v


def add_constant(v, c, stream=None):
    call("gdsp_add_constant", v.ptr, v.n, float(c), _sp(stream))
    return v


def abs_(v, stream=None):
    call("gdsp_abs", v.ptr, v.n, _sp(stream))
    return v


def invert(v, mid, stream=None):
    call("gdsp_invert", v.ptr, v.n, float(mid), _sp(stream))
    return v


def map_values(v, knots_in, knots_out, stream=None):
    """op_map_apply: piecewise-linear mapping through knots sorted by knots_in."""
    a = _dev_array(knots_in, np.float64)
    b = _dev_array(knots_out, np.float64)
    call("gdsp_map", v.ptr, v.n, C.c_void_p(a.ptr), C.c_void_p(b.ptr), len(knots_in), _sp(stream))
    sync(stream)
    return v


def clump(v, average, min_length, above=True, one=1.0, zero=0.0, stream=None):
    """op_clump_apply (above) / op_skimp_apply: in place."""
    work = DeviceBuffer(lib().gdsp_clump_work(v.n))
    call("gdsp_clump", v.ptr, v.n, float(average), int(min_length), int(bool(above)), float(one), float(zero),
         C.c_void_p(work.ptr), _sp(stream))
    sync(stream)
    return v


def fill(v, val, stream=None):
    call("gdsp_fill", v.ptr, v.n, float(val), _sp(stream))
    return v


def genome_minmax(vecs, window=1, lo=-DBL_MAX, hi=DBL_MAX, stream=None):
    """(min, max, count) over the sampled values of all vectors."""
    acc = DeviceBuffer(32)
    call("gdsp_minmax_init", C.c_void_p(acc.ptr), _sp(stream))
    for v in vecs:
        call("gdsp_minmax_update", v.ptr, v.n, window, float(lo), float(hi), C.c_void_p(acc.ptr), _sp(stream))
    r = acc.download(np.float64, 3, stream=stream)
    return float(r[0]), float(r[1]), int(r[2])


# ---------------------------------------------------------- percentile.c ----

# digit schedule over the 64-bit key: sign+exponent first, then the mantissa
SELECT_DIGITS = [(52, 12), (39, 13), (26, 13), (13, 13), (0, 13)]


def radix_select(histogram, p_thousandths, allreduce=None):
    """Host side of the exact order statistic (percentile.c:587-710), shared by every backend.

    histogram(shift, bits, prefix) -> np.uint64 array of (1<<bits)+2 words for THIS rank's
    chromosomes: the digit counts, then the smallest and the largest matching key.
    allreduce(array, op) -> the "sum" / "min" / "max" of the array over ranks (None: one rank).
    Returns (count, [values])."""
    L = lib()
    results = []
    count = None
    first = None           # the prefix-free first pass is shared by every requested percentile

    def reduced(shift, bits, prefix):
        nb = 1 << bits
        h = np.array(histogram(shift, bits, prefix), dtype=np.uint64)
        if allreduce is not None:
            h[:nb] = allreduce(h[:nb].copy(), "sum")
            h[nb:nb + 1] = allreduce(h[nb:nb + 1].copy(), "min")
            h[nb + 1:] = allreduce(h[nb + 1:].copy(), "max")
        return h

    for pt in p_thousandths:
        prefix, value, k = 0, None, None
        for di, (shift, bits) in enumerate(SELECT_DIGITS):
            nb = 1 << bits
            if di == 0:
                if first is None:
                    first = reduced(shift, bits, 0)
                    count = int(first[:nb].sum())
                if count == 0:
                    return 0, []
                hist = first
                k = L.gdsp_percentile_rank(count, int(pt))
            else:
                hist = reduced(shift, bits, prefix)
            if hist[nb] == hist[nb + 1]:           # one distinct candidate left
                value = L.gdsp_key_to_double(int(hist[nb]))
                break
            b, kw = C.c_uint32(), C.c_uint64()
            call("gdsp_select_pick", hist.ctypes.data_as(C.c_void_p), bits, C.c_uint64(k), C.byref(b), C.byref(kw))
            prefix |= b.value << shift
            k = kw.value
        if value is None:
            value = L.gdsp_key_to_double(prefix)
        results.append(value)
    return count, results


class SelectSource(C.Structure):
    """gdsp_select_source of include/genodsp_hip.h"""
    _fields_ = [("d_v", C.c_void_p), ("n", C.c_uint32), ("device", C.c_int), ("stream", C.c_void_p)]


REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.POINTER(C.c_uint64), C.c_size_t, C.c_int)
DEVICE_REDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p)
SELECT_AUTO, SELECT_RADIX, SELECT_BRACKET = 0, 1, 2


import threading as _threading
_PERCENTILE_HOOK = _threading.RLock()


def percentile(vecs, p_thousandths, window=1, lo=-DBL_MAX, hi=DBL_MAX, allreduce=None, stream=None,
               strategy=SELECT_AUTO, sample_target=0, device_allreduce=None):
    """Exact percentiles of the sampled genome (percentile.c:392-751), non-destructive: gdsp_percentiles.
    vecs: this rank's chromosome vectors (current device).  Across ranks, one of
      device_allreduce(ptr, count, "sum"|"min"|"max", stream): all-reduce `count` u64 words at DEVICE address `ptr`
        in place, ordered on `stream` (RCCL through torch.distributed in bench.py: nothing leaves HBM);
      allreduce(np.uint64 array, op) -> the reduced array (host words; the gloo tests).
    Returns (count, [values])."""
    device = current_device()
    src = (SelectSource * max(1, len(vecs)))()
    for i, v in enumerate(vecs):
        src[i].d_v, src[i].n, src[i].device, src[i].stream = v.ptr, v.n, device, stream
    pts = (C.c_uint32 * len(p_thousandths))(*[int(p) for p in p_thousandths])
    vals = (C.c_double * len(p_thousandths))()
    count = C.c_uint64(0)
    failure = []

    def reduce(_ctx, words, n, op):
        try:
            arr = np.ctypeslib.as_array(words, shape=(n,))
            arr[:] = allreduce(arr.copy(), ("sum", "min", "max")[op])
            return 0
        except Exception as e:           # an exception must not unwind through the C frames
            failure.append(e)
            return 1

    def device_reduce(_ctx, d_words, n, op, s):
        try:
            device_allreduce(d_words, n, ("sum", "min", "max")[op], s)
            return 0
        except Exception as e:
            failure.append(e)
            return 1

    assert allreduce is None or device_allreduce is None
    cb = REDUCE_FN(reduce) if allreduce is not None else C.cast(None, REDUCE_FN)
    dcb = DEVICE_REDUCE_FN(device_reduce) if device_allreduce is not None else None
    # the device hook is state of the library (one per process): installed, used and cleared under one lock, so that
    # two threads calling percentile() cannot swap or clear each other's hook in mid-call
    with _PERCENTILE_HOOK:
        try:
            if dcb is not None:
                call("gdsp_percentiles_use_device_reduce", dcb, None)
            call("gdsp_percentiles", src, len(vecs), int(window), float(lo), float(hi), pts, len(p_thousandths),
                 int(strategy), int(sample_target), cb, None, vals, C.byref(count))
        except GdspError:
            if failure:
                raise failure[0]
            raise
        finally:
            if dcb is not None:
                call("gdsp_percentiles_use_device_reduce", None, None)
    if count.value == 0:
        return 0, []
    return int(count.value), [float(x) for x in vals]


class Comm:
    """RCCL communicator over devices of this process (gdsp_comm_create: ncclCommInitAll)."""

    def __init__(self, devices):
        h = C.c_void_p()
        arr = (C.c_int * len(devices))(*devices)
        call("gdsp_comm_create", C.byref(h), arr, len(devices))
        self.handle = h.value
        self.devices = list(devices)

    def allreduce(self, bufs, count, op, dtype="u64", streams=None):
        """bufs: one device address per rank, reduced in place; op 'sum' | 'min' | 'max'."""
        ptrs = (C.c_void_p * len(bufs))(*bufs)
        st = (C.c_void_p * len(bufs))(*(streams or [None] * len(bufs)))
        call("gdsp_comm_allreduce_" + dtype, C.c_void_p(self.handle), ptrs, int(count), ("sum", "min", "max").index(op), st)

    def close(self):
        if self.handle:
            call("gdsp_comm_destroy", C.c_void_p(self.handle))
            self.handle = None


def rccl_version():
    v = C.c_int(0)
    call("gdsp_comm_rccl_version", C.byref(v))
    return v.value


def percentile_stats():
    """What the last percentile() did: route (SELECT_RADIX / SELECT_BRACKET), population, subsample size,
    candidates kept on this rank, percentiles that fell back, histogram passes over the population, whether a fused
    binarize was settled in the counting pass, whether the call was decided on the device (one read-back)."""
    out = (C.c_uint64 * 8)()
    lib().gdsp_percentiles_stats(out)
    keys = ("route", "population", "sample", "candidates", "fallbacks", "population_passes", "binarize_in_one_pass", "resident")
    return dict(zip(keys, [int(x) for x in out]))


class PercentileBinarize(C.Structure):
    _fields_ = [("which", C.c_int), ("tiesAbove", C.c_int), ("one", C.c_double), ("zero", C.c_double), ("d_out", C.c_void_p)]


def percentile_binarize(vecs, p_thousandths, which=0, outs=None, ties_above=False, one=1.0, zero=0.0, window=1, lo=-DBL_MAX,
                        hi=DBL_MAX, stream=None, strategy=SELECT_AUTO, sample_target=0, device_allreduce=None):
    """`= percentile P = binarize --threshold=percentileP` in one read of the signal (gdsp_percentiles_binarize):
    -> (count, [values], outs, one_pass); outs[i] = binarize(vecs[i], values[which]), the vectors are left intact."""
    device = current_device()
    outs = outs if outs is not None else [v.like() for v in vecs]
    src = (SelectSource * max(1, len(vecs)))()
    for i, v in enumerate(vecs):
        src[i].d_v, src[i].n, src[i].device, src[i].stream = v.ptr, v.n, device, stream
    optrs = (C.c_void_p * max(1, len(vecs)))(*[o.ptr.value for o in outs])
    fuse = PercentileBinarize(int(which), int(ties_above), float(one), float(zero), C.cast(optrs, C.c_void_p))
    pts = (C.c_uint32 * len(p_thousandths))(*[int(p) for p in p_thousandths])
    vals = (C.c_double * len(p_thousandths))()
    count, one_pass = C.c_uint64(0), C.c_int(0)
    failure = []

    def device_reduce(_ctx, d_words, n, op, s):
        try:
            device_allreduce(d_words, n, ("sum", "min", "max")[op], s)
            return 0
        except Exception as e:
            failure.append(e)
            return 1

    dcb = DEVICE_REDUCE_FN(device_reduce) if device_allreduce is not None else None
    _PERCENTILE_HOOK.acquire()
    try:
        if dcb is not None:
            call("gdsp_percentiles_use_device_reduce", dcb, None)
        call("gdsp_percentiles_binarize", src, len(vecs), int(window), float(lo), float(hi), pts, len(p_thousandths),
             int(strategy), int(sample_target), C.cast(None, REDUCE_FN), None, vals, C.byref(count), C.byref(fuse), C.byref(one_pass))
    except GdspError:
        if failure:
            raise failure[0]
        raise
    finally:
        if dcb is not None:
            call("gdsp_percentiles_use_device_reduce", None, None)
        _PERCENTILE_HOOK.release()
    if count.value == 0:
        return 0, [], outs, False
    return int(count.value), [float(x) for x in vals], outs, bool(one_pass.value)


def percentile_by_passes(vecs, p_thousandths, window=1, lo=-DBL_MAX, hi=DBL_MAX, allreduce=None, stream=None):
    """The same result through the pass-level entry points (gdsp_select_histogram + host walk), kept for
    callers that drive the passes themselves."""
    def histogram(shift, bits, prefix):
        nb = 1 << bits
        dh = DeviceBuffer((nb + 2) * 8)
        call("gdsp_select_hist_init", C.c_void_p(dh.ptr), bits, _sp(stream))
        for v in vecs:
            call("gdsp_select_histogram", v.ptr, v.n, window, float(lo), float(hi), shift, bits,
                 C.c_uint64(prefix), C.c_void_p(dh.ptr), _sp(stream))
        return dh.download(np.uint64, nb + 2, stream=stream)

    return radix_select(histogram, p_thousandths, allreduce)


def lpt_shards(lengths, nranks):
    """Deal chromosomes longest-first onto the least loaded rank (what genodsp_hip --gpus=N and
    bench.py do); returns a list of chromosome-index lists, one per rank."""
    order = sorted(range(len(lengths)), key=lambda i: -lengths[i])
    load = [0] * nranks
    shards = [[] for _ in range(nranks)]
    for i in order:
        r = min(range(nranks), key=lambda k: load[k])
        shards[r].append(i)
        load[r] += lengths[i]
    return shards


# ------------------------------------- genodsp.c / add.c / multiply.c ------

class BinnedIntervals:
    """Intervals of one chromosome, staged on the device with their tile CSR."""

    def __init__(self, n, start, end, val):
        start = np.ascontiguousarray(start, np.uint32)
        end = np.ascontiguousarray(end, np.uint32)
        val = np.ascontiguousarray(val, np.float64)
        tile = lib().gdsp_interval_tile()
        ntiles = (n + tile - 1) // tile
        offsets = np.zeros(ntiles + 1, np.uint32)
        length = C.c_uint64()
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        call("gdsp_bin_intervals", n, vp(start), vp(end), start.size, vp(offsets), None, C.byref(length))
        tlist = np.zeros(max(length.value, 1), np.uint32)
        call("gdsp_bin_intervals", n, vp(start), vp(end), start.size, vp(offsets), vp(tlist), C.byref(length))
        self.n = n
        self.d_start = _dev_array(start, np.uint32)
        self.d_end = _dev_array(end, np.uint32)
        self.d_val = _dev_array(val, np.float64)
        self.d_offsets = _dev_array(offsets, np.uint32)
        self.d_list = _dev_array(tlist, np.uint32)

    def _args(self):
        return [C.c_void_p(b.ptr) for b in (self.d_start, self.d_end, self.d_val, self.d_offsets, self.d_list)]


def apply_intervals(v, start, end, val, overlap=OVERLAP_SUM, clear=False, missing=0.0, stream=None):
    b = BinnedIntervals(v.n, start, end, val)
    call("gdsp_apply_intervals", v.ptr, v.n, *b._args(), overlap, 3 if clear else 0, float(missing), _sp(stream))
    sync(stream)
    return v


def scale_intervals(v, start, end, val, divide=False, infinity=DBL_MAX, stream=None):
    b = BinnedIntervals(v.n, start, end, val)
    call("gdsp_scale_intervals", v.ptr, v.n, *b._args(), int(divide), float(infinity), _sp(stream))
    sync(stream)
    return v


def mask_intervals(v, start, end, val, inside=True, outside_val=0.0, binarize_first=False, stream=None):
    """mask / or (inside=True) and masknot / and (inside=False); see gdsp_mask_intervals."""
    b = BinnedIntervals(v.n, start, end, val)
    call("gdsp_mask_intervals", v.ptr, v.n, *b._args(), int(inside), float(outside_val), int(binarize_first), _sp(stream))
    sync(stream)
    return v


def extreme_in_intervals(v, start, end, want_max, fill, stream=None):
    """minover / maxover over sorted, non-overlapping intervals."""
    b = BinnedIntervals(v.n, start, end, np.ones(len(start)))
    count = len(start)
    work = DeviceBuffer(lib().gdsp_extreme_in_intervals_work(count))
    call("gdsp_extreme_in_intervals", v.ptr, v.n, C.c_void_p(b.d_start.ptr), C.c_void_p(b.d_end.ptr), count,
         C.c_void_p(b.d_offsets.ptr), C.c_void_p(b.d_list.ptr), int(want_max), float(fill), C.c_void_p(work.ptr), _sp(stream))
    sync(stream)
    return v


def report_runs(v, collapse=True, uncovered=0, stream=None):
    """(start, end, value) arrays of the runs report_intervals would print."""
    work = DeviceBuffer(lib().gdsp_report_runs_work(v.n))
    cnt = DeviceBuffer(16)
    call("gdsp_report_runs", v.ptr, v.n, int(collapse), uncovered, None, None, None, 0,
         C.c_void_p(cnt.ptr), C.c_void_p(work.ptr), _sp(stream))
    n = int(cnt.download(np.uint32, 1, stream=stream)[0])
    if n == 0:
        return np.empty(0, np.uint32), np.empty(0, np.uint32), np.empty(0, np.float64)
    s, e, x = DeviceBuffer(n * 4), DeviceBuffer(n * 4), DeviceBuffer(n * 8)
    call("gdsp_report_runs", v.ptr, v.n, int(collapse), uncovered, C.c_void_p(s.ptr), C.c_void_p(e.ptr),
         C.c_void_p(x.ptr), n, C.c_void_p(cnt.ptr), C.c_void_p(work.ptr), _sp(stream))
    return (s.download(np.uint32, n, stream=stream), e.download(np.uint32, n, stream=stream),
            x.download(np.float64, n, stream=stream))


# ------------------------- one launch per operator per device (gdsp_*_batch) ------

class BatchItem(C.Structure):
    _fields_ = [("d_in", C.c_void_p), ("d_out", C.c_void_p), ("n", C.c_uint32)]


def batch_items(ins, outs):
    """Table for the gdsp_*_batch calls: vector i is read from ins[i] and written to outs[i] (in-place operators use
    outs only; pass ins=None)."""
    items = (BatchItem * max(len(outs), 1))()
    for i, o in enumerate(outs):
        items[i].d_in = ins[i].ptr.value if ins is not None else None
        items[i].d_out = o.ptr.value
        items[i].n = o.n
    return items


def _batch(name, vecs, outs, *params, stream=None, in_place=False):
    outs = outs if outs is not None else ([v for v in vecs] if in_place else [v.like() for v in vecs])
    items = batch_items(None if in_place else vecs, outs)
    call(name, items, len(outs), *params, _sp(stream))
    return outs


def smooth_batch(vecs, W=101, outs=None, mode=FIR_EXACT, stream=None):
    return _batch("gdsp_smooth_batch", vecs, outs, int(W), int(mode), stream=stream)


def smooth_local_extrema_batch(vecs, W, N, want_max, fill, outs=None, mode=FIR_EXACT, stream=None):
    return _batch("gdsp_smooth_local_extrema_batch", vecs, outs, int(W), int(mode), int(N), int(want_max), float(fill),
                  stream=stream)


def local_extrema_batch(vecs, N, want_max, fill, outs=None, stream=None):
    return _batch("gdsp_local_extrema_batch", vecs, outs, int(N), int(want_max), float(fill), stream=stream)


def best_extrema_batch(vecs, W, want_max, outs=None, stream=None):
    return _batch("gdsp_best_extrema_batch", vecs, outs, int(W), int(want_max), stream=stream)


def dilate_batch(vecs, left, right, T=0.0, one=1.0, zero=0.0, outs=None, stream=None):
    return _batch("gdsp_dilate_batch", vecs, outs, left, right, float(T), float(one), float(zero), stream=stream)


def erode_batch(vecs, left, right, T=0.0, one=1.0, zero=0.0, outs=None, stream=None):
    return _batch("gdsp_erode_batch", vecs, outs, left, right, float(T), float(one), float(zero), stream=stream)


def dilate_erode_batch(vecs, d_left, d_right, e_left, e_right, d_T=0.0, d_one=1.0, d_zero=0.0, e_T=0.0, e_one=1.0,
                       e_zero=0.0, binarize=None, outs=None, stream=None):
    b = binarize if binarize is not None else (0.0, False, 1.0, 0.0)
    return _batch("gdsp_dilate_erode_batch", vecs, outs, d_left, d_right, float(d_T), float(d_one), float(d_zero),
                  e_left, e_right, float(e_T), float(e_one), float(e_zero), int(binarize is not None),
                  float(b[0]), int(b[1]), float(b[2]), float(b[3]), stream=stream)


def binarize_batch(vecs, T=0.0, ties_above=False, one=1.0, zero=0.0, stream=None):
    return _batch("gdsp_binarize_batch", vecs, None, float(T), int(ties_above), float(one), float(zero), stream=stream,
                  in_place=True)


def clip_batch(vecs, lo=None, hi=None, stream=None):
    return _batch("gdsp_clip_batch", vecs, None, int(lo is not None), float(lo or 0.0), int(hi is not None), float(hi or 0.0),
                  stream=stream, in_place=True)


def erase_batch(vecs, lo=None, hi=None, keep_inside=False, zero=0.0, stream=None):
    return _batch("gdsp_erase_batch", vecs, None, int(lo is not None), float(lo or 0.0), int(hi is not None), float(hi or 0.0),
                  int(keep_inside), float(zero), stream=stream, in_place=True)


def add_constant_batch(vecs, c, stream=None):
    return _batch("gdsp_add_constant_batch", vecs, None, float(c), stream=stream, in_place=True)


def abs_batch(vecs, stream=None):
    return _batch("gdsp_abs_batch", vecs, None, stream=stream, in_place=True)


def synth_coverage(seed, chrom_index, start, count, mode=0, out=None, stream=None):
    out = out if out is not None else DeviceVector(count)
    call("gdsp_synth_coverage", out.ptr, C.c_uint64(seed), chrom_index, start, count, mode, _sp(stream))
    return out
