#!/usr/bin/env python3
"""Small driver for rocprofv3: a few launches of one operator on one chromosome-sized vector.
usage: python3 tools/prof_op.py <op> [launches] [n]
ops: clump cumsum sum1000 sum100 slidingsum close open dilate erode localmax bestmax binarize smooth_exact smooth_fma
     smooth_hann smooth_hann1001 smooth_hann2001 peaks_exact peaks_fma morph_fused percentile report select
     smooth_hann_batch smooth_exact_batch smooth_fma_batch morph_fused_batch binarize_batch peaks_exact_batch percentile_binarize   (three vectors of n/2, n/3, n/6
     bases in one launch: the gdsp_*_batch forms, what genodsp_hip and bench.py launch by default)"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402

op = sys.argv[1]
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = int(sys.argv[3]) if len(sys.argv) > 3 else 145138636
gd.set_device(0)
depth = gd.synth_coverage(20240611, 7, 0, n, 0)
real = gd.synth_coverage(20240611, 7, 0, n, 1)
a = gd.DeviceVector(n)
b = gd.DeviceVector(n)
l, r = gd.split_length(1001)


def copy(dst, src):
    gd.call("gdsp_memcpy_d2d", dst.ptr, src.ptr, n * 8, None)


# three vectors that share the two big buffers: the one-launch-per-device forms
cuts = [0, (n // 2) & ~1, (n // 2 + n // 3) & ~1, n]
parts_in = lambda src: [gd.DeviceVector(cuts[i+1] - cuts[i], src.buf, 8 * cuts[i]) for i in range(3)]
parts_b = parts_in(b)
work = None
if op == "clump":
    work = gd.DeviceBuffer(gd.lib().gdsp_clump_work(n))
elif op == "cumsum":
    work = gd.DeviceBuffer(gd.lib().gdsp_cumulative_sum_work(n))
RUN = {
    "clump": lambda: (copy(a, depth), gd.call("gdsp_clump", a.ptr, n, 30.5, 1000, 1, 1.0, 0.0, C.c_void_p(work.ptr), None)),
    "cumsum": lambda: (copy(a, depth), gd.call("gdsp_cumulative_sum", a.ptr, n, C.c_void_p(work.ptr), None)),
    "sum1000": lambda: (copy(a, depth), gd.window_sum(a, 1000)),
    "sum100": lambda: (copy(a, depth), gd.window_sum(a, 100)),
    "sum2000": lambda: (copy(a, depth), gd.window_sum(a, 2000)),
    "slidingsum": lambda: gd.sliding_sum(depth, 101, out=b),
    "close": lambda: gd.close(depth, 1001, out=b),
    "open": lambda: gd.open_(depth, 1001, out=b),
    "dilate": lambda: gd.dilate(depth, l, r, out=b),
    "erode": lambda: gd.erode(depth, l, r, out=b),
    "localmax": lambda: gd.localmax(real, 11, out=b),
    "bestmax": lambda: gd.best_extrema(real, 1001, True, out=b),
    "binarize": lambda: (copy(a, depth), gd.binarize(a, 10.0)),
    "smooth_exact": lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_EXACT),
    "smooth_fma": lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_FMA),
    "smooth_hann": lambda: gd.smooth(real, 101, out=b, mode=gd.FIR_HANN),
    "smooth_hann1001": lambda: gd.smooth(real, 1001, out=b, mode=gd.FIR_HANN),
    "smooth_hann2001": lambda: gd.smooth(real, 2001, out=b, mode=gd.FIR_HANN),
    "smooth_hann5001": lambda: gd.smooth(real, 5001, out=b, mode=gd.FIR_HANN),
    "smooth_hann50001": lambda: gd.smooth(real, 50001, out=b, mode=gd.FIR_HANN),
    "peaks_exact": lambda: gd.smooth_local_extrema(real, 101, 11, True, 0.0, out=b, mode=gd.FIR_EXACT),
    "peaks_fma": lambda: gd.smooth_local_extrema(real, 101, 11, True, 0.0, out=b, mode=gd.FIR_FMA),
    "peaks_exact_depth": lambda: gd.smooth_local_extrema(depth, 101, 11, True, 0.0, out=b, mode=gd.FIR_EXACT),
    "morph_fused": lambda: gd.dilate_erode(depth, l, r, l, r, binarize=(0.0, False, 1.0, 0.0), out=b),
    "smooth_hann_batch": lambda: gd.smooth_batch(parts_in(real), 101, outs=parts_b, mode=gd.FIR_HANN),
    "smooth_exact_batch": lambda: gd.smooth_batch(parts_in(real), 101, outs=parts_b, mode=gd.FIR_EXACT),
    "smooth_fma_batch": lambda: gd.smooth_batch(parts_in(real), 101, outs=parts_b, mode=gd.FIR_FMA),
    "percentile_binarize": lambda: gd.percentile_binarize(parts_in(real), [99000], outs=parts_b),
    "peaks_exact_batch": lambda: gd.smooth_local_extrema_batch(parts_in(real), 101, 11, True, 0.0, outs=parts_b, mode=gd.FIR_EXACT),
    "morph_fused_batch": lambda: gd.dilate_erode_batch(parts_in(depth), l, r, l, r, binarize=(0.0, False, 1.0, 0.0), outs=parts_b),
    "binarize_batch": lambda: (copy(a, depth), gd.binarize_batch(parts_in(a), 10.0)),
    "percentile": lambda: gd.percentile([real], [99000]),
    "report": lambda: gd.report_runs(depth),
}
if op in ("percentile_genome", "percentile_binarize_genome"):
    # the genome-wide call of BASELINE configs[4]: 24 sources, 3.1 Gbp (n is ignored)
    import bench
    del depth, real, a, b
    vecs = [gd.synth_coverage(20240611, c, 0, n_c, 1) for c, (_, n_c) in enumerate(bench.GENOME)]
    outs = [gd.DeviceVector(v.n) for v in vecs] if op == "percentile_binarize_genome" else None
    RUN[op] = (lambda: gd.percentile(vecs, [99000])) if outs is None else (lambda: gd.percentile_binarize(vecs, [99000], outs=outs))
for _ in range(launches):
    RUN[op]()
gd.sync()
print("done", op, launches, n)
