"""GPU: the C host driver (genodsp_amd/genodsp_hip) against the reference CLI's recorded output.

Each golden "cli" case holds a command line, the text fed on stdin and what the unmodified
reference binary printed (tests/golden/make_golden.py).  stdout must match byte for byte:
this covers text ingest (read_intervals), every operator in the pipeline and report_intervals.
"""
import os
import subprocess

import pytest

import cli_compare
from conftest import ROOT, golden

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "genodsp_amd", "genodsp_hip")
CLI_CASES = golden().cli_cases()


def run(args, stdin_text="", chroms_text=None, tmp_path=None, files=None):
    if files:
        real = []
        for a in args:
            for key, text in files.items():
                path = os.path.join(str(tmp_path), key + ".dat")
                with open(path, "w") as f:
                    f.write(text)
                a = a.replace("@%s@" % key, path)
            real.append(a)
        args = real
    if chroms_text is not None:
        path = os.path.join(str(tmp_path), "genome.chroms")
        with open(path, "w") as f:
            f.write(chroms_text)
        args = ["--chromosomes=" + path] + list(args)
    p = subprocess.run([BIN] + list(args), input=stdin_text, capture_output=True, text=True, timeout=300)
    cli_compare.remember([BIN] + list(args), None, stdin_text, p.returncode, p.stdout, p.stderr, files)
    return p.returncode, p.stdout, p.stderr


@pytest.fixture(scope="module", autouse=True)
def built():
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "genodsp_amd", "host")])


# the driver runs batchable stretches of a pipeline operator-major, one launch per operator per device; --nobatch keeps
# the reference's order (every operator of the run on one chromosome, then the next: genodsp.c:909-921)
ORDERS = {"batch": [], "nobatch": ["--nobatch"]}


@pytest.mark.parametrize("order", list(ORDERS))
@pytest.mark.parametrize("case", CLI_CASES, ids=[c["name"] for c in CLI_CASES])
def test_cli_stdout_matches_reference(case, order, tmp_path):
    if case["returncode"] != 0:
        pytest.skip("reference itself failed on this input")
    if order == "nobatch" and "--debug=pipe" in case["args"]:
        pytest.skip("--debug=pipe echoes the arguments")
    rc, out, err = run(ORDERS[order] + case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    if case["name"] in ("cli_percentile99", "cli_percentile_extremes_0_map"):
        # the reference prints its sort-scrambled signal after percentile (percentile.c:34-36);
        # here the signal is untouched, i.e. the output of the same pipeline without the operator
        assert out == golden().cases["cli_coverage"]["stdout"]
    else:
        assert out == case["stdout"]
    if case["name"].startswith(("cli_percentile_", "cli_debug_")):
        # what percentile reports and what it does not (the extremes are answered silently, percentile.c:432-530),
        # and the variables it leaves behind, in the reference's order
        want = case["stderr"]
        for key in case.get("files") or {}:        # (--debug=pipe echoes the arguments: the file paths differ)
            want = want.replace("/tmp/golden_%s_%s" % (case["name"], key), "@%s@" % key)
            err = err.replace(os.path.join(str(tmp_path), key + ".dat"), "@%s@" % key)
        assert err == want
    # the percentile report line goes to stderr in both programs
    for line in case["stderr"].splitlines():
        if line.startswith("percentile "):
            assert line in err.splitlines()


DIGEST_CASES = [c for c in golden().meta["cases"] if c["kind"] == "cli_digest"]


@pytest.mark.parametrize("order", list(ORDERS))
@pytest.mark.parametrize("case", DIGEST_CASES, ids=[c["name"] for c in DIGEST_CASES])
def test_cli_random_pipelines_match_the_reference_binary(case, order, tmp_path):
    """48 random command lines (global flags, one to five random operators, shuffled valued intervals over two
    chromosomes) recorded from the reference binary as a digest of its stdout: the whole driver -- ingest,
    operators, named variables, report -- must print the same bytes."""
    assert case["returncode"] == 0
    rc, out, err = run(ORDERS[order] + case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert cli_compare.assert_matches_reference(case, rc, out, err) == "digest"


def test_named_variable_feeds_threshold(tmp_path):
    """percentile -> binarize --threshold=<variable> (README.md:111-118 in the reference); here the
    signal survives percentile, so no re-input is needed."""
    chroms = "chr1 100\nchr2 50\n"
    iv = "chr1 10 20\nchr1 15 30\nchr2 0 5\nchr1 15 18\n"
    rc, out, err = run(["--novalue", "=", "percentile", "50", "--min=1/inf", "=", "binarize", "--threshold=percentile50"],
                       iv, chroms, tmp_path)
    assert rc == 0, err
    assert "using percentile50 = " in err
    # depth>1 only on chr1 15..20
    assert out == "chr1\t15\t20\t1\n"


def test_interval_file_operators(tmp_path):
    chroms = "chr1 100\nchr2 50\n"
    iv = "chr1 10 20 2.5\nchr1 15 30 1.5\nchr2 0 5 4\n"
    addf = tmp_path / "add.dat"
    addf.write_text("chr1 0 12 1\nchr2 3 8 0.5\n")
    rc, out, err = run(["--precision=2", "=", "add", str(addf)], iv, chroms, tmp_path)
    assert rc == 0, err
    assert out == ("chr1\t0\t10\t1.00\nchr1\t10\t12\t3.50\nchr1\t12\t15\t2.50\nchr1\t15\t20\t4.00\n"
                   "chr1\t20\t30\t1.50\nchr2\t0\t3\t4.00\nchr2\t3\t5\t4.50\nchr2\t5\t8\t0.50\n")
    mulf = tmp_path / "mul.dat"
    mulf.write_text("chr1 0 16 2\nchr1 18 40 3\n")
    rc, out, err = run(["--precision=2", "=", "multiply", str(mulf)], iv, chroms, tmp_path)
    assert rc == 0, err
    assert out == "chr1\t10\t15\t5.00\nchr1\t15\t16\t8.00\nchr1\t18\t20\t12.00\nchr1\t20\t30\t4.50\n"
    outf = tmp_path / "mid.dat"
    rc, out, err = run(["--precision=1", "=", "output", str(outf), "=", "addconst", "1", "=", "input", str(outf)],
                       iv, chroms, tmp_path)
    assert rc == 0, err
    assert out == outf.read_text()


def test_errors_are_loud(tmp_path):
    chroms = "chr1 100\n"
    rc, out, err = run(["=", "nosuchop"], "", chroms, tmp_path)
    assert rc != 0 and "not a known operation" in err
    rc, out, err = run(["=", "clump", "3", "--length=0"], "", chroms, tmp_path)
    assert rc != 0 and "minimum length can't be zero" in err
    rc, out, err = run(["=", "clump", "--average=nothere"], "chr1 1 2 3\n", chroms, tmp_path)
    assert rc != 0 and "no such variable" in err
    rc, out, err = run(["--novalue"], "chr1 90 120\n", chroms, tmp_path)
    assert rc != 0 and "beyond the end of the chromosome" in err
    rc, out, err = run(["--novalue", "--cliptochromosome"], "chr1 90 120\n", chroms, tmp_path)
    assert rc == 0 and out == "chr1\t90\t100\t1\n"
    rc, out, err = run(["=", "binarize", "--threshold=nothere"], "chr1 1 2 3\n", chroms, tmp_path)
    assert rc != 0 and "no such variable" in err
    rc, out, err = run(["=", "smooth", "W=60000"], "", chroms, tmp_path)
    assert rc != 0 and "exceeds" in err


def test_chromosome_specs_on_command_line_and_origin(tmp_path):
    rc, out, err = run(["chrZ:1000", "--novalue", "--origin=one"], "chrZ 1 10\nchrZ 5 12\n")
    assert rc == 0, err
    assert out == "chrZ\t1\t4\t1\nchrZ\t5\t10\t2\nchrZ\t11\t12\t1\n"
    rc, out, err = run(["chrZ:100:200", "--novalue"], "chrZ 90 110\nchrZ 150 160\nchrZ 195 300\n")
    assert rc == 0, err
    assert out == "chrZ\t100\t110\t1\nchrZ\t150\t160\t1\nchrZ\t195\t200\t1\n"


@pytest.mark.parametrize("name", ["cli_smooth_localmax", "cli_dilate_erode_binarize"])
def test_fused_chains_equal_separate_operators(name, tmp_path):
    """The driver fuses these chains by default; --nofuse runs one kernel per operator.  Both must
    print what the reference printed."""
    case = golden().cases[name]
    for extra in ([], ["--nofuse"]):
        rc, out, err = run(extra + case["args"], case["stdin"], case["chroms_text"], tmp_path)
        assert rc == 0, err
        assert out == case["stdout"], extra


def test_sharded_driver_gives_the_same_output(tmp_path, monkeypatch):
    """--gpus=N deals chromosomes to N device shards (own stream, scratch and staging each).  With
    GDSP_OVERSUBSCRIBE_GPUS the shards share the visible GPU(s), which exercises the sharded code
    path -- per-shard percentile histograms summed on the host included -- on a one-GPU box."""
    monkeypatch.setenv("GDSP_OVERSUBSCRIBE_GPUS", "1")
    chroms = "".join("chr%d %d\n" % (i, 3000 + 700 * i) for i in range(7))
    rng_lines = []
    import numpy as np
    rng = np.random.default_rng(4)
    for i in range(7):
        n = 3000 + 700 * i
        for _ in range(150):
            s = int(rng.integers(0, n - 60))
            rng_lines.append("chr%d %d %d %d" % (i, s, s + int(rng.integers(1, 60)), int(rng.integers(1, 6))))
    iv = "\n".join(rng_lines) + "\n"
    pipeline = ["=", "smooth", "W=21", "=", "percentile", "90", "--min=1/inf", "=", "clip", "--max=percentile90",
                "=", "dilate", "30", "=", "erode", "30", "=", "invert"]
    outs = []
    for gpus in (1, 3):
        rc, out, err = run(["--precision=9", "--gpus=%d" % gpus] + pipeline, iv, chroms, tmp_path)
        assert rc == 0, err
        outs.append((out, [l for l in err.splitlines() if l.startswith("percentile")]))
    assert outs[0] == outs[1]


@pytest.mark.parametrize("case", DIGEST_CASES[::2], ids=[c["name"] for c in DIGEST_CASES[::2]])
def test_base_sharding_prints_what_the_reference_printed(case, tmp_path, monkeypatch):
    """--gpus=3 --sharding=bases: the chromosomes are cut into equal shares of the genome's bases; runs of
    operators with a bounded reach (smooth, extrema, morphology, pointwise) work on the stretches, each carrying the
    halo the chain reaches into and refreshed from its neighbours between runs; everything else (running sums, clump,
    file-driven operators, ingest, report) gets whole chromosomes back.  The three shards share the one GPU of the
    test box.  Same bytes as the reference binary."""
    import hashlib
    monkeypatch.setenv("GDSP_OVERSUBSCRIBE_GPUS", "1")
    rc, out, err = run(["--gpus=3", "--sharding=bases", "--progress=operations", "--batch"] + case["args"], case["stdin"],
                       case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    assert hashlib.sha256(out.encode()).hexdigest() == case["sha256"], (case["args"], out[:300])
    for line in case["stderr_percentile"]:
        assert line in err.splitlines()


def test_base_sharding_cuts_chromosomes_and_matches_whole_chromosomes(tmp_path, monkeypatch):
    """Two chromosomes of very different length over 4 shards: the long one must be cut (the progress lines name the
    stretches), and every pipeline -- fused chains, long reaches, percentile with a window, invert, a running sum in
    the middle (whole chromosomes again) -- prints exactly what --sharding=chromosomes prints."""
    import numpy as np
    monkeypatch.setenv("GDSP_OVERSUBSCRIBE_GPUS", "1")
    chroms = "chrL 90000\nchrS 7000\n"
    rng = np.random.default_rng(21)
    lines = []
    for c, n in (("chrL", 90000), ("chrS", 7000)):
        for _ in range(n // 20):
            a = int(rng.integers(0, n - 300))
            lines.append("%s %d %d %.2f" % (c, a, a + int(rng.integers(1, 300)), rng.random() * 6 - 1))
    iv = "\n".join(lines) + "\n"
    pipelines = [["=", "smooth", "W=101", "=", "localmax", "N=11"],
                 ["--nofuse", "=", "smooth", "W=101", "=", "localmax", "N=11"],
                 ["=", "dilate", "1001", "=", "erode", "1001", "=", "binarize"],
                 ["=", "close", "300", "=", "open", "40", "=", "bestmax", "W=500", "=", "localmin", "N=201", "--infinity=9"],
                 ["=", "percentile", "90", "--window=7", "--min=0.5", "=", "clip", "--max=percentile90", "=", "invert", "=", "abs"],
                 ["=", "smooth", "W=21", "=", "cumulativesum", "=", "smooth", "W=51", "=", "slidingsum", "W=30", "=", "bestmin", "W=9"],
                 ["=", "clip", "--min=0", "=", "addconst", "0.25", "=", "smooth", "W=5001"]]
    for pl in pipelines:
        got = {}
        for how in ("chromosomes", "bases"):
            rc, out, err = run(["--precision=10", "--gpus=4", "--sharding=" + how, "--progress=operations", "--batch"] + pl, iv, chroms, tmp_path)
            assert rc == 0, err
            got[how] = out
            if how == "bases" and pl[1] != "percentile":
                first = pl[2] if pl[0] == "--nofuse" else pl[1]
                if "cumulativesum" in pl:        # a running sum anywhere in a run of operators: the run gets whole chromosomes
                    assert ("%s(chrL)" % first) in err and "chrL:0-" not in err
                else:
                    assert ("%s(chrL:0-" % first) in err and ("%s(chrS:0-7000)" % first) in err, err[-1500:]
        assert got["bases"] == got["chromosomes"], pl
        assert len(got["bases"].splitlines()) >= 2


def test_progress_lines_come_in_the_reference_order(tmp_path):
    """--progress=operations: the reference applies a run of operators chromosome by chromosome, longest first
    (genodsp.c:909-921), and prints `operator(chromosome)` as it goes; the driver keeps that order when the lines are
    asked for (one launch per operator and device otherwise, or with --batch), and prints the same signal either way."""
    chroms = "chrA 5000\nchrB 9000\n"
    iv = "chrA 10 400 2\nchrB 100 900 3\nchrB 500 2500 1\n"
    pl = ["=", "addconst", "1", "=", "clip", "--max=3", "=", "abs"]
    rc, out, err = run(["--progress=operations"] + pl, iv, chroms, tmp_path)
    assert rc == 0, err
    ops = [l for l in err.splitlines() if l.split("(")[0] in ("addconst", "clip", "abs")]
    assert ops == ["addconst(chrB)", "clip(chrB)", "abs(chrB)", "addconst(chrA)", "clip(chrA)", "abs(chrA)"], err
    rc, out2, err2 = run(["--progress=operations", "--batch"] + pl, iv, chroms, tmp_path)
    assert rc == 0, err2
    ops2 = [l for l in err2.splitlines() if l.split("(")[0] in ("addconst", "clip", "abs")]
    assert ops2 == ["addconst(chrB)", "addconst(chrA)", "clip(chrB)", "clip(chrA)", "abs(chrB)", "abs(chrA)"], err2
    assert out == out2


def test_rccl_communicator_reduces_percentile_and_invert(tmp_path, monkeypatch):
    """--reduce=rccl: percentile's histograms / counters and invert's extremes go through ncclCommInitAll +
    ncclAllReduce (gdsp_comm.hip) even with one device, so communicator creation and the u64 sum / min / max and
    f64 min / max all-reduces run on the one-GPU test box; values and output equal the host-sum path's.  With a
    communicator the bracket route stays resident (gdsp_percentile.hip: pc_resident, every digit pass cut at its
    reduction -- count, ncclAllReduce queued on the stream, pick): one read-back, as on one device without one."""
    monkeypatch.setenv("GDSP_PERCENTILE_REPORT", "1")
    chroms = "".join("chr%d %d\n" % (i, 30000 + 7000 * i) for i in range(5))
    import numpy as np
    rng = np.random.default_rng(14)
    lines = []
    for i in range(5):
        n = 30000 + 7000 * i
        for _ in range(1500):
            a = int(rng.integers(0, n - 90))
            lines.append("chr%d %d %d %.3f" % (i, a, a + int(rng.integers(1, 90)), rng.random() * 7 - 2))
    iv = "\n".join(lines) + "\n"
    got = {}
    for how in ("host", "rccl"):
        for route in ("radix", "bracket"):
            rc, out, err = run(["--precision=12", "--reduce=" + how, "--percentile=" + route, "--progress=operations", "--batch",
                                "=", "percentile", "5..95by15", "--min=-1", "=", "clip", "--max=percentile80", "=", "invert",
                                "=", "percentile", "0,100", "=", "variables"], iv, chroms, tmp_path)
            assert rc == 0, err
            assert ("reduce(rccl" in err) == (how == "rccl"), err
            routes = [l for l in err.splitlines() if l.startswith("[percentile] route=")]
            assert len(routes) == 2 and "fallbacks=0" in routes[0], err
            if route == "bracket":
                assert "route=bracket resident=1 readbacks=1" in routes[0], (how, routes)
            else:
                assert "route=radix resident=0" in routes[0], (how, routes)
            got[how, route] = (out, [l for l in err.splitlines() if "percentile" in l and "(" not in l and not l.startswith("[percentile] route=")])
    assert len(got["host", "radix"][1]) >= 9
    assert got["host", "radix"] == got["rccl", "radix"] == got["host", "bracket"] == got["rccl", "bracket"]
    rc, out, err = run(["--gpus=2", "--reduce=rccl"], "", "chr1 100\n", tmp_path)
    assert rc != 0                                   # one GPU here: two shards need two GPUs (or GDSP_OVERSUBSCRIBE_GPUS + host sums)


def test_smooth_arithmetic_modes_on_the_command_line(tmp_path):
    """--smooth=exact (default) prints the reference's digits; fma and hann are within one rounding per
    operation, far below what --precision=9 shows on this signal."""
    import numpy as np
    rng = np.random.default_rng(8)
    n = 9000
    lines = []
    for _ in range(900):
        s = int(rng.integers(0, n - 200))
        lines.append("chrQ %d %d %d" % (s, s + int(rng.integers(20, 200)), int(rng.integers(1, 9))))
    iv = "\n".join(lines) + "\n"
    outs = {}
    for mode in ("exact", "fma", "hann"):
        rc, out, err = run(["--precision=9", "--smooth=" + mode, "=", "smooth", "W=101"], iv, "chrQ %d\n" % n, tmp_path)
        assert rc == 0, err
        v = np.zeros(n)
        for l in out.splitlines():                   # runs of equal values are collapsed: expand them
            f = l.split()
            v[int(f[1]):int(f[2])] = float(f[3])
        outs[mode] = v
        assert len(out.splitlines()) > 1000
    for mode in ("fma", "hann"):
        assert np.abs(outs[mode] - outs["exact"]).max() <= 2e-9
    rc, out, err = run(["--smooth=nosuch"], "", "chrQ 10\n", tmp_path)
    assert rc != 0


def test_hann_in_front_of_strict_comparisons_prints_the_fma_bytes(tmp_path):
    """--smooth=hann is not shift invariant (equal windows can differ in their last bits), so a smooth that feeds
    localmax / localmin is evaluated tap by tap with fused multiply-adds instead (ops_sum.c,
    ops_fused.c): the peaks printed under --smooth=hann are byte for byte those of --smooth=fma, fused or with --nofuse,
    in either launch order -- and the smoothed track itself (no comparison behind it) does differ between the two."""
    import numpy as np
    rng = np.random.default_rng(18)
    chroms = "chrP 60000\nchrQ 9100\n"
    lines = []
    for c, n in (("chrP", 60000), ("chrQ", 9100)):
        for _ in range(n // 12):
            a = int(rng.integers(0, n - 250))
            lines.append("%s %d %d %.3f" % (c, a, a + int(rng.integers(10, 250)), rng.random() * 5))
    iv = "\n".join(lines) + "\n"
    for tail, orders in ((["=", "localmax", "N=11"], ([], ["--nofuse"], ["--nobatch"], ["--nofuse", "--nobatch"])),
                         (["=", "localmin", "N=7", "--infinity=99"], ([], ["--nofuse"]))):
        got = {}
        for mode in ("fma", "hann"):
            for extra in orders:
                rc, out, err = run(["--precision=15", "--smooth=" + mode] + extra + ["=", "smooth", "W=101"] + tail, iv, chroms, tmp_path)
                assert rc == 0, err
                got[mode, tuple(extra)] = out
        assert len(set(got.values())) == 1, tail
        assert len(got["fma", ()].splitlines()) > 50
    plain = {}
    for mode in ("fma", "hann"):
        rc, plain[mode], err = run(["--precision=17", "--smooth=" + mode, "=", "smooth", "W=101"], iv, chroms, tmp_path)
        assert rc == 0, err
    assert plain["fma"] != plain["hann"]             # the substitution is the comparison's, not the smooth's


def _reads(rng, n_lines, chrom_len, with_values):
    lines = []
    for i in range(n_lines):
        c = "chr%d" % (1 + int(rng.integers(0, 3)))
        a = int(rng.integers(0, chrom_len - 300))
        b = a + int(rng.integers(1, 300))
        if with_values:
            lines.append("%s\t%d\t%d\t%s" % (c, a, b, ("%d" % rng.integers(1, 9)) if i % 3 else ("%.3f" % (rng.random() * 4))))
        else:
            lines.append("%s %d %d" % (c, a, b))
        if i % 5000 == 17:
            lines.append("# a comment")
        if i % 7000 == 23:
            lines.append("track name=x")
        if i % 9000 == 31:
            lines.append("")
    return lines


@pytest.mark.parametrize("with_values", [False, True])
def test_threaded_ingest_reads_what_the_line_reader_reads(with_values, tmp_path, monkeypatch):
    """The block reader (ingest.c: 16 MiB blocks cut into one stretch per thread) against the line-at-a-time
    reader it replaces (GDSP_INGEST_THREADS=1): same output, and for broken input the same first complaint."""
    import numpy as np
    rng = np.random.default_rng(5)
    chroms = "chr1 400000\nchr2 300000\nchr3 200000\n"
    good = _reads(rng, 120000, 200000, with_values)                 # ~2.5 MB: several stretches
    args = ["--precision=3", "=", "smooth", "W=11"] if with_values else ["--novalue"]

    def both(lines, final_newline=True):
        text = "\n".join(lines) + ("\n" if final_newline else "")
        got = []
        for threads in ("1", "7"):
            monkeypatch.setenv("GDSP_INGEST_THREADS", threads)
            got.append(run(args, text, chroms, tmp_path))
        assert got[0] == got[1], (got[0][0], got[1][0], got[0][2][-300:], got[1][2][-300:])
        return got[0]

    rc, out, err = both(good)
    assert rc == 0 and len(out.splitlines()) > 1000
    rc2, out2, _ = both(good, final_newline=False)
    assert (rc2, out2) == (rc, out)
    for where in (3, 60000, len(good) - 2):                          # first, middle and last stretch
        for bad in ("chr1 12", "chr1", "chr1 x12 40", "chr1 12 -4", " chr1 5 9", "chr2 7 9 " + ("zz" if with_values else "1"),
                    "chr1 5 " + "9" * 1100, "chr1\t10\t20\t1\t" + "x" * 1200):
            lines = list(good)
            lines[where] = bad
            rc, out, err = both(lines)
            if bad.endswith("zz") or "x12" in bad or "-4" in bad or len(bad) > 1000 or bad in ("chr1 12", "chr1", " chr1 5 9"):
                assert rc != 0 and err != "", bad
    # two broken lines: the earlier one is the one reported
    lines = list(good)
    lines[90000] = "chr1 1"
    lines[20000] = "chr1 zz 4"
    rc, out, err = both(lines)
    assert rc != 0 and "zz" in err


@pytest.mark.parametrize("case", DIGEST_CASES[::6], ids=[c["name"] for c in DIGEST_CASES[::6]])
def test_many_small_interval_batches_give_the_same_output(case, tmp_path, monkeypatch):
    """Ingest and the interval-file operators buffer 8 M intervals before they go to the device; GDSP_BATCH_INTERVALS
    forces a batch every 7 intervals, so that the batch seams (first-touch rule of `input`, file order of
    overlapping `add`) sit all over these inputs.  The reference binary's digest must still be met."""
    import hashlib
    monkeypatch.setenv("GDSP_BATCH_INTERVALS", "7")
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    assert hashlib.sha256(out.encode()).hexdigest() == case["sha256"], case["args"]


def test_percentile_feeding_binarize_runs_in_one_read_of_the_signal(tmp_path, monkeypatch):
    """`= percentile P = binarize --threshold=percentileP` without --preserve: the driver hands both operators to
    gdsp_percentiles_binarize (the counting pass writes one / zero wherever its bracket decides).  Same stdout, same
    stderr (the percentile line, then binarize's note about its threshold) as the two operators one after the other
    (--nofuse), with read depth and with real values, several percentiles, ties above, under --gpus=2 shards too."""
    import numpy as np
    monkeypatch.setenv("GDSP_OVERSUBSCRIBE_GPUS", "1")
    rng = np.random.default_rng(31)
    n = 2_600_000                                        # above the 2^20 values where the bracketing route starts
    for valued in (False, True):
        lines = []
        for _ in range(60000):
            a = int(rng.integers(0, n - 400))
            z = a + int(rng.integers(30, 400))
            lines.append("chrP\t%d\t%d\t%.3f" % (a, z, rng.random() * 3 + 0.1) if valued else "chrP\t%d\t%d" % (a, z))
        for i in range(300):
            lines.append("chrQ\t%d\t%d%s" % (40 * i, 40 * i + 55, "\t2.5" if valued else ""))
        iv = "\n".join(lines) + "\n"
        base = ([] if valued else ["--novalue"]) + ["--precision=3"]
        for ops in (["=", "percentile", "90", "--min=1/inf", "=", "binarize", "--threshold=percentile90"],
                    ["=", "percentile", "50..99by7", "=", "binarize", "--threshold=percentile92", "--ties:above", "--one=5", "--zero=-1", "=", "addconst", "1"],
                    ["=", "smooth", "W=11", "=", "percentile", "99.5", "--min=1/inf", "--quiet", "=", "binarize", "T=percentile99.5", "=", "dilate", "30"]):
            got = {}
            for extra in ([], ["--nofuse"], ["--gpus=2"]):
                rc, out, err = run(base + extra + ops, iv, "chrP %d\nchrQ 13000\n" % n, tmp_path)
                assert rc == 0, err
                got[tuple(extra)] = (out, err)
            assert got[()] == got[("--nofuse",)] == got[("--gpus=2",)], ops
            assert len(got[()][0].splitlines()) > 100 and "[binarize] using percentile" in got[()][1]


POISONS = ["1e300", "-1e300", "nan"]


# every case under 1e300, every second one under the other two
POISONED = [(c, p) for i, c in enumerate(DIGEST_CASES) for p in POISONS if p == "1e300" or (i % 2 == POISONS.index(p) - 1)]


@pytest.mark.parametrize("case,poison", POISONED, ids=["%s-%s" % (c["name"], p) for c, p in POISONED])
def test_no_operator_reads_memory_nobody_wrote(case, poison, tmp_path, monkeypatch):
    """GDSP_POISON: every device allocation is filled with the value before it is handed out and a vector's partner is
    refilled after every flip (gdsp_runtime.hip, genodsp_hip.c: flip_spec).  A fresh box hands out zeros, which hides a
    kernel that reads a base nobody wrote (the tail of an odd-length vector, a partner, a workspace); with a value that
    wins every maximum (1e300), every minimum (-1e300) or spoils every sum (nan) such a read changes the output.  The
    reference binary's digest must still be met."""
    monkeypatch.setenv("GDSP_POISON", poison)
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert cli_compare.assert_matches_reference(case, rc, out, err) == "digest"
