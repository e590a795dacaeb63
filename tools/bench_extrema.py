"""bestmax / localmax timing for a few windows on one chromosome-sized vector."""
import sys
sys.path.insert(0, ".")
import genodsp_amd as gd  # noqa: E402

n = 248956422
S = gd.Stream()
real = gd.synth_coverage(20240611, 0, 0, n, 1)
out = real.like()
for name, fn in (("bestmax W=301", lambda: gd.best_extrema(real, 301, True, out=out, stream=S.handle)),
                 ("bestmax W=1001", lambda: gd.best_extrema(real, 1001, True, out=out, stream=S.handle)),
                 ("bestmax W=3001", lambda: gd.best_extrema(real, 3001, True, out=out, stream=S.handle)),
                 ("bestmax W=2049", lambda: gd.best_extrema(real, 2049, True, out=out, stream=S.handle)),
                 ("bestmax W=3601", lambda: gd.best_extrema(real, 3601, True, out=out, stream=S.handle)),
                 ("bestmax W=33", lambda: gd.best_extrema(real, 33, True, out=out, stream=S.handle)),
                 ("bestmax W=100", lambda: gd.best_extrema(real, 100, True, out=out, stream=S.handle)),
                 ("localmax N=11", lambda: gd.local_extrema(real, 11, True, 0.0, out=out, stream=S.handle)),
                 ("localmax N=33", lambda: gd.local_extrema(real, 33, True, 0.0, out=out, stream=S.handle)),
                 ("localmax N=101", lambda: gd.local_extrema(real, 101, True, 0.0, out=out, stream=S.handle)),
                 ("localmax N=1001", lambda: gd.local_extrema(real, 1001, True, 0.0, out=out, stream=S.handle))):
    best = 1e30
    for _ in range(5):
        gd.sync(S.handle)
        e0, e1 = gd.Event(), gd.Event()
        e0.record(S.handle)
        fn()
        e1.record(S.handle)
        best = min(best, e0.elapsed_ms(e1))
    print("%-18s %8.3f ms %7.1f Gbases/s %6.2f TB/s" % (name, best, n / best / 1e6, 16 * n / best / 1e9))
