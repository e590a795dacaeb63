#!/usr/bin/env python3
"""percentile 99 of one chromosome-sized vector: wall time of the whole call and rocprof-free per-call device time.
usage: python3 tools/bench_percentile.py [n] [reps]"""
import os
import sys
import time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import genodsp_amd as gd  # noqa: E402
n = int(sys.argv[1]) if len(sys.argv) > 1 else 248956422
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
gd.set_device(0)
real = gd.synth_coverage(20240611, 0, 0, n, 1)
depth = gd.synth_coverage(20240611, 0, 0, n, 0)
targets = [int(t) for t in os.environ.get("TARGETS", "0").split(",")]
ROUTES = (("resident", {}), ("chained", {"GDSP_PERCENTILE_RESIDENT_OFF": "1"}),
          ("host", {"GDSP_PERCENTILE_RESIDENT_OFF": "1", "GDSP_PERCENTILE_CHAIN_OFF": "1"}))
only = os.environ.get("ROUTES")
if only:
    ROUTES = tuple(r for r in ROUTES if r[0] in only.split(","))
for name, v in (("real", real), ("depth", depth)):
  for route, env in ROUTES:
    for k in ("GDSP_PERCENTILE_RESIDENT_OFF", "GDSP_PERCENTILE_CHAIN_OFF"):
        os.environ.pop(k, None)
    os.environ.update(env)
    os.environ["TAG"] = route
    for target in targets:
        best = 1e30
        for _ in range(reps):
            gd.sync()
            t0 = time.perf_counter()
            vals = gd.percentile([v], [99000], sample_target=target)
            gd.sync()
            best = min(best, time.perf_counter() - t0)
        print("percentile 99 on %-5s %s sample_target %9d  %8.3f ms  %7.1f Gbases/s  %5.1f%% of 8 TB/s at 8 B/base  -> %r" % (
            name, os.environ.get("TAG", ""), target, best * 1e3, n / best / 1e9, 100 * 8 * n / best / 8e12, vals))

if os.environ.get("GENOME"):
    # the genome-wide call: 24 sources, 3.1 Gbp, one percentile (what `= percentile 99` costs by itself)
    import bench
    del real, depth
    vecs = [gd.synth_coverage(20240611, c, 0, n_c, 1) for c, (_, n_c) in enumerate(bench.GENOME)]
    total = sum(n_c for _, n_c in bench.GENOME)
    for route, env in ROUTES:
        for k in ("GDSP_PERCENTILE_RESIDENT_OFF", "GDSP_PERCENTILE_CHAIN_OFF"):
            os.environ.pop(k, None)
        os.environ.update(env)
        best = 1e30
        for _ in range(reps):
            gd.sync()
            t0 = time.perf_counter()
            vals = gd.percentile(vecs, [99000])
            gd.sync()
            best = min(best, time.perf_counter() - t0)
        print("percentile 99 on the 3.1 Gbp genome (24 sources) %-8s %8.3f ms  %7.1f Gbases/s  %5.1f%% of 8 TB/s at 8 B/base  -> %r" % (
            route, best * 1e3, total / best / 1e9, 100 * 8 * total / best / 8e12, vals))
