#!/usr/bin/env python3
"""Generate tests/golden/golden.npz + golden.json from the UNMODIFIED reference.

Runs only where oracle/_ref has been built from /root/reference (`make -C oracle ref`).
Each case pushes seeded f64 vectors through the reference's own parse + apply
functions (oracle/ref_harness.c) and records inputs and outputs at full precision;
CLI cases record the reference binary's stdout/stderr for small text inputs.
The fixtures are data only: no reference source is stored.

    python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
from oracle import cpu, ref  # noqa: E402

SEED = 20240611
rng = np.random.default_rng(SEED)
arrays = {}
cases = []


def signal(kind, n, chrom_index=0):
    if kind == "depth":      # integer read depth, piecewise constant (our synthetic generator)
        return cpu.synth_coverage(SEED, chrom_index, 0, n, 0)
    if kind == "real":       # depth * U(0.5,1.5)
        return cpu.synth_coverage(SEED, chrom_index, 0, n, 1)
    if kind == "noise":      # signed reals
        return rng.standard_normal(n) * 7.0
    if kind == "blocky":     # small integers in short runs, with zeros
        return np.repeat(rng.integers(0, 4, n // 5 + 1), 5)[:n].astype(np.float64)
    if kind == "islands":    # sparse islands of depth separated by longer gaps
        v = np.zeros(n)
        pos = 0
        while pos < n:
            gap = int(rng.integers(1, 40))
            run = int(rng.integers(1, 25))
            v[pos + gap:pos + gap + run] = float(rng.integers(1, 9))
            pos += gap + run
        return v
    raise ValueError(kind)


def vector_case(name, chroms, pipeline, inputs, want_globals=(), files=None):
    """chroms: [(name, length)], inputs: {chrom: vector}; files: {placeholder: text}, "@placeholder@"
    in the pipeline stands for that file's path."""
    g = ref.Genome(chroms)
    for c, v in inputs.items():
        g.set(c, v)
    real = pipeline
    for key, text in (files or {}).items():
        path = "/tmp/golden_%s_%s" % (name, key)
        with open(path, "w") as f:
            f.write(text)
        real = real.replace("@%s@" % key, path)
    g.run(real)
    rec = {"name": name, "kind": "vector", "chroms": chroms, "pipeline": pipeline, "globals": {}, "files": files or {}}
    for c, _ in chroms:
        arrays["%s/in/%s" % (name, c)] = np.asarray(inputs[c], np.float64)
        arrays["%s/out/%s" % (name, c)] = g.get(c)
    for gl in want_globals:
        val = g.get_global(gl)
        rec["globals"][gl] = None if val is None else float(val).hex()
    rec["sorted"] = g.sorted_names()
    g.close()
    cases.append(rec)


def one(name, kind, n, pipeline):
    vector_case(name, [("chrA", n)], pipeline, {"chrA": signal(kind, n)})


# ---- Hann taps (sum.c:632-645) read back through an impulse at full precision
for W in (3, 5, 11, 101, 1001):
    n = 3 * W + 7
    v = np.zeros(n)
    v[n // 2] = 1.0
    one("hann_impulse_W%d" % W, "depth", n, "= smooth W=%d" % W)
    arrays["hann_impulse_W%d/in/chrA" % W] = v
    g = ref.Genome([("chrA", n)])
    g.set("chrA", v)
    g.run("= smooth W=%d" % W)
    arrays["hann_impulse_W%d/out/chrA" % W] = g.get("chrA")
    g.close()

# ---- smooth
for kind in ("depth", "real", "noise"):
    for W, n in ((5, 400), (21, 700), (101, 2600), (101, 5003), (301, 1500)):
        one("smooth_%s_W%d_n%d" % (kind, W, n), kind, n, "= smooth W=%d" % W)
one("smooth_default_window", "real", 900, "= smooth")
one("smooth_even_window_raised", "real", 500, "= smooth W=10")
one("smooth_tiny_vector", "real", 60, "= smooth W=101")          # n > hOff=50: still defined

# ---- local extrema / sliding extrema
for kind in ("depth", "real", "blocky"):
    for N in (3, 11, 41, 201):
        one("localmax_%s_N%d" % (kind, N), kind, 1500, "= localmax N=%d" % N)
        one("localmin_%s_N%d" % (kind, N), kind, 1500, "= localmin N=%d" % N)
    for W in (3, 10, 100, 1001):
        one("bestmax_%s_W%d" % (kind, W), kind, 2500, "= bestmax W=%d" % W)
        one("bestmin_%s_W%d" % (kind, W), kind, 2500, "= bestmin W=%d" % W)
one("localmax_fill", "real", 700, "= localmax N=5 --zero=-1")
one("localmin_fill", "real", 700, "= localmin N=5 --infinity=99")
one("smooth_then_localmax", "depth", 4000, "= smooth W=101 = localmax N=11")

# ---- window sums
for kind in ("depth", "blocky", "real"):
    for W in (4, 101, 1000):
        one("slidingsum_%s_W%d" % (kind, W), kind, 3000, "= slidingsum W=%d" % W)
    one("slidingsum_%s_denom" % kind, kind, 1000, "= slidingsum W=25 --denom=W")
    one("sum_%s_W100" % kind, kind, 2350, "= sum W=100")
    one("sum_%s_actual" % kind, kind, 2350, "= sum W=64 --denom=actual --zero=-2")
    one("sum_%s_chrom" % kind, kind, 1234, "= sum --window=chromosome")
    one("cumsum_%s" % kind, kind, 3000, "= cumulativesum")

# ---- morphology (islands keeps erode clear of the reference's u32 underflow: no run
#      ends left of the erosion length because every vector starts with a gap > left)
for L in (1, 2, 7, 20, 61):
    v = signal("islands", 3000)
    v[:L + 2] = 0.0
    for op in ("dilate", "erode", "close", "open"):
        vector_case("%s_islands_L%d" % (op, L), [("chrA", 3000)], "= %s %d" % (op, L), {"chrA": v})
v = signal("depth", 4000)
v[:600] = 0.0
vector_case("dilate_erode_binarize", [("chrA", 4000)], "= dilate 1001 = erode 1001 = binarize", {"chrA": v})
vector_case("dilate_threshold_vals", [("chrA", 4000)], "= dilate 30 --threshold=20 --one=5 --zero=-1", {"chrA": v})
vector_case("dilate_left_right", [("chrA", 4000)], "= dilate --left=7 --right=2", {"chrA": v})
vector_case("erode_left_right", [("chrA", 4000)], "= erode --left=3 --right=9 --threshold=10", {"chrA": v})
vector_case("close_threshold", [("chrA", 4000)], "= close 150 --threshold=30", {"chrA": v})
vector_case("open_threshold", [("chrA", 4000)], "= open 150 --threshold=10 --one=2", {"chrA": v})

# ---- elementwise
one("binarize_default", "noise", 999, "= binarize")
one("binarize_T", "depth", 999, "= binarize 20")
one("binarize_above", "depth", 999, "= binarize 20 --ties:above --one=3 --zero=-3")
one("clip_both", "noise", 999, "= clip --min=-2.5 --max=4")
one("clip_min", "noise", 999, "= clip --min=-2.5")
one("clip_max", "noise", 999, "= clip --max=4")
for flags in ("--min=-1 --max=3", "--min=-1", "--max=3", "--min=-1 --max=3 --keep:inside --zero=7",
              "--min=-1 --keep:inside", "--max=3 --keep:inside"):
    one("erase_" + flags.replace(" ", "_").replace("-", "").replace("=", "").replace(":", ""),
        "noise", 999, "= erase " + flags)
one("addconst", "real", 999, "= addconst 1.75")
one("addconst_zero", "real", 999, "= addconst 0")
one("abs", "noise", 999, "= abs")
vector_case("invert_genome", [("c1", 500), ("c2", 800), ("c3", 300)], "= invert",
            {"c1": signal("noise", 500), "c2": signal("noise", 800), "c3": signal("noise", 300)})
vector_case("invert_mid", [("c1", 500)], "= invert 2.5", {"c1": signal("noise", 500)})

# ---- map (map.c): strictly increasing knots, given out of order (the reference sorts them)
MAPFILE = "# in out\n10 100\n0 0\n2.5 -4\n\n40 41.5\n25 3e2\n61 7\n"
for kind in ("depth", "real", "noise"):
    vector_case("map_" + kind, [("chrA", 1500)], "= map @m@", {"chrA": signal(kind, 1500)}, files={"m": MAPFILE})
vector_case("map_single_knot", [("chrA", 300)], "= map @m@", {"chrA": signal("noise", 300)}, files={"m": "3 9\n"})

# ---- percentile: named globals; --preserve is not needed because outputs are not compared
chroms3 = [("c1", 2000), ("c2", 3500), ("c3", 1200)]
sig3 = {"c1": signal("depth", 2000, 0), "c2": signal("depth", 3500, 1), "c3": signal("depth", 1200, 2)}
sig3r = {"c1": signal("real", 2000, 0), "c2": signal("real", 3500, 1), "c3": signal("real", 1200, 2)}
vector_case("percentile99_depth", chroms3, "= percentile 99 --quiet", sig3, ["percentile99"])
vector_case("percentile99_nz_depth", chroms3, "= percentile 99 --min=1/inf --quiet", sig3, ["percentile99"])
vector_case("percentile_range_depth", chroms3, "= percentile 50..100by10 --min=1/inf --quiet", sig3,
            ["percentile%d" % p for p in (50, 60, 70, 80, 90, 100)])
vector_case("percentile_real", chroms3, "= percentile 0.5..99.5by9 --quiet", sig3r,
            ["percentile%s" % p for p in ("0.5", "9.5", "18.5", "27.5", "36.5", "45.5", "54.5", "63.5",
                                          "72.5", "81.5", "90.5", "99.5")])
vector_case("percentile_window", chroms3, "= percentile 75 --window=7 --max=30 --quiet", sig3r, ["percentile75"])
vector_case("percentile_0_100", chroms3, "= percentile 0,100 --quiet", sig3r, ["percentile0", "percentile100"])

# ---- ingest + report through text (genodsp.c:1187-1350, :1561-1691), reference CLI
def cli_case(name, chrom_text, args, stdin_text, files=None):
    """files: {placeholder: text}; an argument "@placeholder@" stands for the path of that file."""
    chrom_path = "/tmp/golden_%s.chroms" % name
    with open(chrom_path, "w") as f:
        f.write(chrom_text)
    real_args = []
    for a in args:
        for key, text in (files or {}).items():
            path = "/tmp/golden_%s_%s" % (name, key)
            with open(path, "w") as f:
                f.write(text)
            a = a.replace("@%s@" % key, path)
        real_args.append(a)
    rc, out, err = ref.run_cli(["--chromosomes=" + chrom_path] + real_args, stdin_text)
    cases.append({"name": name, "kind": "cli", "chroms_text": chrom_text, "args": args, "files": files or {},
                  "stdin": stdin_text, "returncode": rc, "stdout": out, "stderr": err})


APPENDIX_C_IV = "chr1 10 20\nchr1 15 30\nchr2 0 5\nchrX 1 2\n# comment\ntrack foo\nchr1 15 18 extra\n"
APPENDIX_C_CH = "chr1 100\nchr2 50\n"
cli_case("cli_coverage", APPENDIX_C_CH, ["--novalue"], APPENDIX_C_IV)
cli_case("cli_smooth5", APPENDIX_C_CH, ["--novalue", "--precision=6", "=", "smooth", "--window=5"], APPENDIX_C_IV)
cli_case("cli_smooth_localmax", APPENDIX_C_CH,
         ["--novalue", "--precision=6", "=", "smooth", "W=5", "=", "localmax", "N=11"], APPENDIX_C_IV)
cli_case("cli_dilate6", APPENDIX_C_CH, ["--novalue", "=", "dilate", "6"], APPENDIX_C_IV)
cli_case("cli_erode6", APPENDIX_C_CH, ["--novalue", "=", "erode", "6"], APPENDIX_C_IV)
cli_case("cli_dilate_erode_binarize", APPENDIX_C_CH,
         ["--novalue", "=", "dilate", "7", "=", "erode", "7", "=", "binarize", "0.5"], APPENDIX_C_IV)
cli_case("cli_percentile99", APPENDIX_C_CH, ["--novalue", "=", "percentile", "99"], APPENDIX_C_IV)
cli_case("cli_show_uncovered", APPENDIX_C_CH, ["--novalue", "--uncovered:show"], APPENDIX_C_IV)
cli_case("cli_NA_uncovered", APPENDIX_C_CH, ["--novalue", "--uncovered:NA"], APPENDIX_C_IV)
cli_case("cli_nocollapse_origin1", APPENDIX_C_CH, ["--novalue", "--nocollapse", "--origin=one"], APPENDIX_C_IV)

# config 1 of BASELINE.json: coverage depth (--novalue) on 1 chromosome x 1 Mbp
def coverage_intervals(n, seed):
    r = np.random.default_rng(seed)
    lines, pos = [], 0
    while True:
        pos += int(r.integers(1, 200))
        length = int(r.integers(20, 300))
        start = max(0, pos - int(r.integers(0, 120)))
        if start + length > n:
            break
        lines.append("chrS\t%d\t%d" % (start, start + length))
    return "\n".join(lines) + "\n"


cli_case("cli_config1_coverage_1Mbp", "chrS 1000000\n", ["--novalue"], coverage_intervals(1000000, 1))

# valued intervals with overlaps, non-integer values: file-order accumulation
def valued_intervals(n, seed, count):
    r = np.random.default_rng(seed)
    lines = []
    for _ in range(count):
        s = int(r.integers(0, n - 50))
        e = s + int(r.integers(1, 50))
        lines.append("chrV %d %d %.3f" % (s, e, r.random() * 3))
    return "\n".join(lines) + "\n"


cli_case("cli_valued_overlaps", "chrV 5000\n", ["--precision=12", "--nocollapse"], valued_intervals(5000, 2, 800))

# interval-file operators (add.c, multiply.c, mask.c, logical.c, minmax.c, opio.c)
GENOME3 = "chrA 300\nchrB 120\nchrC 40\n"
SIGNAL3 = "".join("chrA %d %d %s\n" % (s, e, v) for s, e, v in
                  [(5, 60, "2.5"), (40, 90, "1.25"), (100, 101, "7"), (150, 260, "0.5"), (255, 300, "3")]) + \
          "chrB 0 50 4\nchrB 60 61 -2\nchrB 100 120 1.5\n"
SORTED_IV = "chrA 10 50 2\nchrA 50 55 0\nchrA 80 120 0.5\nchrA 250 300 4\nchrB 30 70 3\n"
LOOSE_IV = "chrB 10 30 1.5\nchrA 20 200 0.75\nchrA 0 30 6\nchrC 5 6 9\nchrA 190 290 0\n"
for op, ivs, extra in (("add", LOOSE_IV, []), ("subtract", LOOSE_IV, []), ("multiply", SORTED_IV, []),
                       ("divide", SORTED_IV, []), ("divide", SORTED_IV, ["--infinity=1000"]),
                       ("mask", LOOSE_IV, ["--mask=-1"]), ("mask", LOOSE_IV, []), ("masknot", SORTED_IV, ["--mask=9"]),
                       ("or", LOOSE_IV, []), ("and", SORTED_IV, []), ("minwith", LOOSE_IV, []),
                       ("maxwith", LOOSE_IV, []), ("or", LOOSE_IV, ["--novalue"]),
                       ("minover", SORTED_IV, ["--infinity=99"]), ("maxover", SORTED_IV, []),
                       ("maxover", SORTED_IV, ["--zero=-1"])):
    tag = "cli_file_%s%s" % (op, "".join(e.strip("-").replace("=", "") for e in extra))
    cli_case(tag, GENOME3, ["--precision=4", "--uncovered:show", "=", op, "@iv@"] + extra, SIGNAL3, {"iv": ivs})
cli_case("cli_file_input_output", GENOME3,
         ["--precision=3", "=", "output", "@mid@", "=", "addconst", "1", "=", "input", "@iv@", "--missing=2", "--overlap=max"],
         SIGNAL3, {"iv": LOOSE_IV, "mid": ""})
cli_case("cli_file_mask_variable", GENOME3,
         ["--precision=3", "=", "percentile", "50", "--quiet", "=", "input", "@sig@", "=", "mask", "@iv@", "--mask=percentile50"],
         SIGNAL3, {"iv": SORTED_IV, "sig": SIGNAL3})

# ---- clump / anticlump (clump.c); appended last so the seeded generator leaves earlier cases unchanged.
#      Depth against dyadic thresholds keeps the running sums exact (see gdsp_clump.hip).
for kind in ("depth", "blocky", "islands"):
    v = signal(kind, 3000)
    for op in ("clump", "anticlump"):
        for T, L in (("2", 20), ("1.5", 100), ("10", 7), ("0.25", 1)):
            vector_case("%s_%s_T%s_L%d" % (op, kind, T.replace(".", "p"), L), [("chrA", 3000)],
                        "= %s %s --length=%d" % (op, T, L), {"chrA": v})
vector_case("clump_default_length", [("chrA", 2000)], "= clump 12", {"chrA": signal("depth", 2000)})
vector_case("clump_one_zero", [("chrA", 2000)], "= clump 12 L=50 --one=7 --zero=-1", {"chrA": signal("depth", 2000)})
vector_case("skimp_alias", [("chrA", 2000)], "= skimp 12 L=50", {"chrA": signal("depth", 2000)})
vector_case("clump_all_below", [("chrA", 500)], "= clump 1000 L=5", {"chrA": signal("depth", 500)})
vector_case("clump_all_above", [("chrA", 500)], "= clump -1 L=5", {"chrA": signal("depth", 500)})
vector_case("clump_relative", [("c1", 2000), ("c2", 900)], "= clump 12 --length=CL/40",
            {"c1": signal("depth", 2000, 0), "c2": signal("depth", 900, 1)})
vector_case("clump_relative_max", [("c1", 2000), ("c2", 900)], "= anticlump 12 --length=max(0.02*CL,30)",
            {"c1": signal("depth", 2000, 0), "c2": signal("depth", 900, 1)})
vector_case("clump_longer_than_vector", [("chrA", 300)], "= clump 1 L=1K", {"chrA": signal("depth", 300)})
vector_case("clump_whole_vector", [("chrA", 300)], "= clump 1 L=CL", {"chrA": signal("depth", 300) + 1})
vector_case("clump_percentile_variable", [("c1", 2000), ("c2", 900)],       # --preserve: the signal survives percentile
            "= percentile 75 --quiet --preserve=@keep@ = clump --average=percentile75 L=40",
            {"c1": signal("depth", 2000, 0), "c2": signal("depth", 900, 1)}, ["percentile75"], files={"keep": ""})
cli_case("cli_clump", APPENDIX_C_CH, ["--novalue", "=", "clump", "1.5", "L=8"], APPENDIX_C_IV)
cli_case("cli_anticlump_show", APPENDIX_C_CH, ["--novalue", "--uncovered:show", "=", "anticlump", "0.5", "L=12", "--one=2"],
         APPENDIX_C_IV)

# ---- random pipelines through the reference BINARY (whole driver: ingest -> operators -> report), stored as a
#      digest of its stdout: sha256, line count, first and last lines.  Signal kept away from the chromosome ends
#      and windows kept short of them, so that none of the reference's out-of-bounds reads (SURVEY Appendix B
#      #1-2) is provoked; percentile only with --preserve, which leaves the signal usable afterwards.
import hashlib  # noqa: E402


def random_cli_case(k, scale=1, into=None, keep_stdout=False):
    """scale > 1: chromosomes and interval counts that many times larger, so that the kernels' tiles, the ingest
    batches and the report's chunks all meet their seams (golden_seams.json.gz holds 40 such cases, golden.json the
    scale-1 ones); keep_stdout: the whole text the reference printed is kept beside the digest"""
    r = np.random.default_rng(SEED + 7000 + k)
    lens = [int(r.integers(6000 * scale, 9000 * scale)), int(r.integers(4000 * scale, 6000 * scale))]
    chroms_text = "".join("chr%s %d\n" % ("RQ"[i], n) for i, n in enumerate(lens))
    lines = []
    for i, n in enumerate(lens):
        for _ in range(int(r.integers(30 * scale, 90 * scale))):
            a = int(r.integers(1500, n - 1800))
            b = a + int(r.integers(1, 260))
            lines.append("chr%s\t%d\t%d\t%d" % ("RQ"[i], a, b, int(r.integers(1, 7))))
    order = r.permutation(len(lines))
    stdin = "\n".join(lines[j] for j in order) + "\n"
    args = ["--precision=%d" % int(r.integers(0, 7))]
    if r.random() < 0.3:
        args.append("--novalue")
    if r.random() < 0.3:
        args.append("--uncovered:%s" % r.choice(["show", "NA"]))
    if r.random() < 0.2:
        args.append("--nocollapse")
    if r.random() < 0.2:
        args.append("--origin=one")
    menu = [lambda: ["smooth", "W=%d" % int(r.choice([5, 11, 51, 101]))],
            lambda: ["localmax", "N=%d" % int(r.choice([3, 11, 41]))],
            lambda: ["localmin", "N=%d" % int(r.choice([5, 21])), "--infinity=50"],
            lambda: ["bestmax", "W=%d" % int(r.choice([4, 30, 300]))],
            lambda: ["bestmin", "W=%d" % int(r.choice([9, 100]))],
            lambda: ["dilate", "%d" % int(r.integers(1, 300))],
            lambda: ["erode", "%d" % int(r.integers(1, 120))],
            lambda: ["close", "%d" % int(r.integers(2, 400)), "--threshold=%d" % int(r.integers(0, 3))],
            lambda: ["open", "%d" % int(r.integers(2, 100))],
            lambda: ["binarize", "%d" % int(r.integers(0, 5))],
            lambda: ["clip", "--min=1", "--max=%d" % int(r.integers(3, 9))],
            lambda: ["erase", "--max=%d" % int(r.integers(1, 4))],
            lambda: ["addconst", "%g" % float(r.integers(-2, 3))],
            lambda: ["abs"],
            lambda: ["invert", "2.5"],
            lambda: ["slidingsum", "W=%d" % int(r.choice([4, 50, 501]))],
            lambda: ["sum", "W=%d" % int(r.choice([10, 100]))],
            lambda: ["cumulativesum"],
            lambda: ["clump", "%g" % (float(r.integers(1, 6)) + 0.5), "--length=%d" % int(r.choice([10, 150]))],
            lambda: ["anticlump", "0.5", "--length=%d" % int(r.choice([20, 400]))],
            lambda: ["percentile", "%d" % int(r.choice([50, 90, 99])), "--min=1/inf", "--quiet", "--preserve=@keep@"]]
    pct = None
    for _ in range(int(r.integers(1, 6))):
        op = menu[int(r.integers(0, len(menu)))]()
        if op[0] == "percentile":
            pct = op[1]
        args += ["="] + op
        if pct is not None and op[0] != "percentile" and r.random() < 0.5:
            args += ["=", "binarize", "--threshold=percentile%s" % pct]
            pct = None
    name = "cli_random_%02d" % k
    files = {"keep": ""} if "--preserve=@keep@" in args else {}
    chrom_path = "/tmp/golden_%s.chroms" % name
    with open(chrom_path, "w") as f:
        f.write(chroms_text)
    real = [a.replace("@keep@", "/tmp/golden_%s_keep" % name) for a in args]
    rc, out, err = ref.run_cli(["--chromosomes=" + chrom_path] + real, stdin)
    body = out.splitlines()
    case = {"name": name, "kind": "cli_digest", "chroms_text": chroms_text, "args": args, "files": files,
            "stdin": stdin, "returncode": rc, "sha256": hashlib.sha256(out.encode()).hexdigest(),
            "lines": len(body), "head": body[:5], "tail": body[-3:],
            "stderr_percentile": [l for l in err.splitlines() if l.startswith("percentile ")]}
    if keep_stdout:
        case["stdout"] = out
    (cases if into is None else into).append(case)


for k in range(48):
    random_cli_case(k)


def random_file_case(k):
    """as above, with interval-file operators: loose (overlapping, any order) files for the operators that take
    them, sorted disjoint ones for the operators that walk the gaps between intervals"""
    r = np.random.default_rng(SEED + 9000 + k)
    lens = [int(r.integers(3000, 5000)), int(r.integers(1500, 2500)), 400]
    names = ["chrF", "chrG", "chrH"]
    chroms_text = "".join("%s %d\n" % (c, n) for c, n in zip(names, lens))

    def loose(count):
        out = []
        for _ in range(count):
            i = int(r.integers(0, 3))
            a = int(r.integers(0, lens[i] - 120))
            out.append("%s %d %d %s" % (names[i], a, a + int(r.integers(1, 120)), "%.2f" % (r.random() * 5 - 1)))
        return "\n".join(out) + "\n"

    def disjoint():
        out = []
        for i in range(3):
            pos = int(r.integers(0, 60))
            while pos < lens[i] - 150:
                e = pos + int(r.integers(1, 140))
                out.append("%s %d %d %s" % (names[i], pos, e, "%.2f" % (r.random() * 4 + 0.25)))
                pos = e + int(r.integers(0, 160))
        return "\n".join(out) + "\n"

    files, args = {}, ["--precision=%d" % int(r.integers(0, 6))]
    if r.random() < 0.4:
        args.append("--uncovered:show")
    for j in range(int(r.integers(1, 4))):
        key = "f%d" % j
        op = str(r.choice(["add", "subtract", "mask", "or", "minwith", "maxwith", "multiply", "divide", "masknot", "and",
                           "minover", "maxover", "input", "smooth", "binarize", "dilate"]))
        if op in ("smooth", "binarize", "dilate"):
            args += ["=", op] + {"smooth": ["W=11"], "binarize": ["1"], "dilate": ["15"]}[op]
            continue
        files[key] = disjoint() if op in ("multiply", "divide", "masknot", "and", "minover", "maxover") else loose(int(r.integers(5, 60)))
        extra = {"mask": ["--mask=%d" % int(r.integers(-2, 3))], "masknot": ["--mask=7"], "divide": ["--infinity=99"],
                 "minover": ["--infinity=50"], "maxover": ["--zero=-1"],
                 "input": ["--missing=%d" % int(r.integers(0, 3)), "--overlap=%s" % r.choice(["sum", "min", "max"])]}.get(op, [])
        if op == "or" and r.random() < 0.5:
            extra = ["--novalue"]
        args += ["=", op, "@%s@" % key] + extra
    name = "cli_random_files_%02d" % k
    stdin = loose(int(r.integers(20, 120)))
    chrom_path = "/tmp/golden_%s.chroms" % name
    with open(chrom_path, "w") as f:
        f.write(chroms_text)
    real = []
    for a in args:
        for key, text in files.items():
            path = "/tmp/golden_%s_%s" % (name, key)
            with open(path, "w") as f:
                f.write(text)
            a = a.replace("@%s@" % key, path)
        real.append(a)
    rc, out, err = ref.run_cli(["--chromosomes=" + chrom_path] + real, stdin)
    body = out.splitlines()
    cases.append({"name": name, "kind": "cli_digest", "chroms_text": chroms_text, "args": args, "files": files,
                  "stdin": stdin, "returncode": rc, "sha256": hashlib.sha256(out.encode()).hexdigest(),
                  "lines": len(body), "head": body[:5], "tail": body[-3:], "stderr_percentile": []})


for k in range(40):
    random_file_case(k)

# ---- round 2 (appended; earlier cases keep their random streams) -------------------------------------------
# percentile: the global --window= is its default window (percentile.c:153); --preserve really writes the signal
# with ten decimals and reads it back (:532-535, :716-724), so values return rounded; `0`, `100` and any range
# from 0 to 100 are answered from the extremes alone, silently (:432-530)
def long_decimals(n, seed, count):
    r = np.random.default_rng(seed)
    out = []
    for _ in range(count):
        a = int(r.integers(0, n - 80))
        out.append("chrV %d %d %.15f" % (a, a + int(r.integers(1, 80)), r.random() * 3))
    return "\n".join(out) + "\n"


cli_case("cli_percentile_global_window_preserve", "chrV 4000\nchrW 900\n",
         ["--window=7", "--precision=14", "=", "percentile", "75", "--preserve=@keep@"],
         long_decimals(4000, 11, 300) + "chrW 10 500 0.00000000004\nchrW 100 300 1.00000000005\n", {"keep": ""})
cli_case("cli_percentile_opt_window_beats_global", "chrV 4000\n",
         ["--window=7", "--precision=3", "=", "percentile", "40..60by10", "--window=3", "--preserve=@keep@", "=", "variables"],
         long_decimals(4000, 12, 200), {"keep": ""})
for tag, what in (("0", ["0"]), ("100", ["100"]), ("0_to_100by10", ["0..100by10"]), ("0_100_bash", ["0,100", "--report:bash"]),
                  ("0_map", ["0", "--map=@m@"])):
    cli_case("cli_percentile_extremes_" + tag, APPENDIX_C_CH,
             ["--novalue", "=", "percentile"] + what + ["--min=1/inf", "=", "variables"], APPENDIX_C_IV,
             {"m": ""} if "--map=@m@" in what else None)

# the three diagnostics of genodsp.c:553-561: argument scanning, the named-variable channel, every input line echoed
cli_case("cli_debug_pipe_globals_input", APPENDIX_C_CH,
         ["--novalue", "--debug=pipe", "--debug=globals", "--precision=2", "--debug=input", "--window=5", "=", "addconst", "1",
          "=", "slidingsum", "=", "mask", "@m@", "=", "variables"], APPENDIX_C_IV, {"m": "chr1 12 40\n# kept\n"})

# ---- seam-crossing command lines and the running-sum cases, in a file of their own (golden_seams.json.gz):
#      40 of the random pipelines above at thirty times the size (chromosomes of 120-270 kbp, 1800-5400 intervals), as
#      digests; and six scale-1 pipelines in which a running sum (slidingsum / cumulativesum) follows `smooth` on real
#      values -- the one place where a parallel evaluation cannot reproduce the reference's single accumulator to the
#      last bit -- with the reference's whole output, so that the test can hold every base to a stated bound.
import gzip  # noqa: E402

seams = []
k = 5000
while len(seams) < 40:                                    # (a pipeline the reference itself stops at is no fixture)
    random_cli_case(k, 30, into=seams)
    if seams[-1]["returncode"] != 0:
        seams.pop()
    k += 1
RUNNING_SUM_SEEDS = (1044, 1125, 1380, 1411, 1455, 1522)
for k in RUNNING_SUM_SEEDS:
    random_cli_case(k, 1, into=seams, keep_stdout=True)
    seams[-1]["kind"] = "cli_running_sum"


def long_window_case(k, into):
    """windows beyond what one LDS tile of the HIP kernels holds (8190 bases for the extrema, 4001 taps for the
    block-sum smooth, the tiled sums' 8192): the whole-vector routes (gdsp_*_any), which the driver must reach from
    its batched and from its per-chromosome order alike.  Islands of coverage 3-14 kbp long and 3-14 kbp apart, so
    that reaches of 8-20 kbp join some and not others; signal kept 15 000 bases away from the ends and `erode` only
    behind a longer `dilate`, so that none of the reference's out-of-bounds reads is provoked."""
    r = np.random.default_rng(SEED + 9000 + k)
    lens = [int(r.integers(120000, 150000)), int(r.integers(90000, 110000))]
    chroms_text = "".join("chr%s %d\n" % ("LM"[i], n) for i, n in enumerate(lens))
    lines = []
    for i, n in enumerate(lens):
        a = 15000
        while True:
            z = a + int(r.integers(3000, 14000))
            if z > n - 15600:
                break
            lines.append("chr%s\t%d\t%d\t1" % ("LM"[i], a, z))
            for _ in range(int(r.integers(20, 80))):
                b = int(r.integers(a, z - 1))
                lines.append("chr%s\t%d\t%d\t%d" % ("LM"[i], b, min(z, b + int(r.integers(1, 900))), int(r.integers(1, 7))))
            a = z + int(r.integers(3000, 14000))
    stdin = "\n".join(lines[j] for j in r.permutation(len(lines))) + "\n"
    menu = [lambda: [["bestmax", "W=%d" % int(r.choice([8193, 20001, 30000]))]],
            lambda: [["bestmin", "W=%d" % int(r.choice([9001, 16384]))]],
            lambda: [["localmax", "N=%d" % int(r.choice([8200, 20001]))]],
            lambda: [["localmin", "N=12001", "--infinity=50"]],
            lambda: [["dilate", "%d" % int(r.choice([9000, 20001]))]],
            lambda: [["dilate", "20001"], ["erode", "%d" % int(r.choice([8200, 15001]))]],
            lambda: [["close", "9000"]],
            lambda: [["open", "8500", "--threshold=1"]],
            lambda: [["smooth", "W=%d" % int(r.choice([4801, 10001]))]],
            lambda: [["slidingsum", "W=20001"]],
            lambda: [["sum", "W=9000"]]]
    args = ["--precision=%d" % int(r.integers(0, 5))] + (["--uncovered:show"] if r.random() < 0.3 else [])
    for _ in range(int(r.integers(1, 3))):
        for op in menu[int(r.integers(0, len(menu)))]():
            args += ["="] + op
    name = "cli_long_window_%02d" % k
    chrom_path = "/tmp/golden_%s.chroms" % name
    with open(chrom_path, "w") as f:
        f.write(chroms_text)
    rc, out, err = ref.run_cli(["--chromosomes=" + chrom_path] + args, stdin)
    body = out.splitlines()
    if (rc != 0) or (len(body) < 3):                  # (stopped by the reference itself, or everything erased: says nothing)
        return False
    into.append({"name": name, "kind": "cli_digest", "chroms_text": chroms_text, "args": args, "files": {},
                 "stdin": stdin, "returncode": rc, "sha256": hashlib.sha256(out.encode()).hexdigest(),
                 "lines": len(body), "head": body[:5], "tail": body[-3:], "stderr_percentile": []})
    return True


k, kept = 0, 0
while kept < 16:
    kept += long_window_case(k, seams)
    k += 1
with gzip.GzipFile(os.path.join(HERE, "golden_seams.json.gz"), "wb", mtime=0) as f:
    f.write(json.dumps({"seed": SEED, "cases": seams}).encode())

np.savez_compressed(os.path.join(HERE, "golden.npz"), **arrays)
with open(os.path.join(HERE, "golden.json"), "w") as f:
    json.dump({"seed": SEED, "cases": cases}, f, indent=1)
print("wrote %d cases, %d arrays" % (len(cases), len(arrays)))
