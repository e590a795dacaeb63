"""CPU: the driver's host code -- everything that parses untrusted text (host/ingest.c's thread team and
read_interval, the chromosome file and option parsers of genodsp_hip.c, utilities.c's number parsers, put_fixed's
hand-made %.*f) -- under AddressSanitizer + UndefinedBehaviorSanitizer, fed malformed, huge and truncated input, and
what it accepts, rejects, prints and complains about compared with the reference binary (oracle/_ref/genodsp, where it
has been built).  tests/host_asan builds the host sources against a stub of the GPU library (malloc for HBM, the
oracle for ingest / report): test infrastructure, never the product.
Reference: genodsp.c:728-814 (chromosome file), :1187-1350, :1384-1534 (intervals), utilities.c:236-355 (numbers).
"""
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

DIR = os.path.join(ROOT, "tests", "host_asan")
BIN = os.path.join(DIR, "genodsp_host_asan")
REF = os.path.join(ROOT, "oracle", "_ref", "genodsp")
ENV = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=99", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1:exitcode=98")


@pytest.fixture(scope="module", autouse=True)
def built():
    subprocess.check_call(["make", "-s", "-C", DIR])


def run(exe, args, stdin, tmp_path, chroms="chr1 1000\nchr2 500\n", files=None):
    path = os.path.join(str(tmp_path), "g.chroms")
    if chroms is not None:
        with open(path, "wb") as f:
            f.write(chroms if isinstance(chroms, bytes) else chroms.encode())
    real = []
    for a in args:
        for key, text in (files or {}).items():
            fp = os.path.join(str(tmp_path), key + ".dat")
            with open(fp, "wb") as f:
                f.write(text if isinstance(text, bytes) else text.encode())
            a = a.replace("@%s@" % key, fp)
        real.append(a)
    if chroms is not None:
        real = ["--chromosomes=" + path] + real
    p = subprocess.run([exe] + real, input=stdin if isinstance(stdin, bytes) else stdin.encode(), capture_output=True,
                       timeout=120, env=ENV)
    return p.returncode, p.stdout, p.stderr.decode(errors="replace")


def complaint(err):
    """what the program said went wrong: stderr up to the usage text, program names aside; the echo of the offending
    argument at the end of an operator's complaint is left out (the reference's parsers cut their arguments up in
    place, e.g. percentile.c:262-300, and echo what is left of them)"""
    import re
    keep = []
    for line in err.splitlines():
        if line.startswith(("usage:", "  ")) or line.strip() == "":
            break
        keep.append(re.sub(r' \("[^"]*"\)$', "", line.replace("genodsp_hip", "genodsp")))
    return keep


def same_as_reference(args, stdin, tmp_path, **kw):
    rc, out, err = run(BIN, args, stdin, tmp_path, **kw)
    assert rc not in (97, 98, 99) and "Sanitizer" not in err and "runtime error" not in err, err[-3000:]
    if os.path.exists(REF):
        rrc, rout, rerr = run(REF, args, stdin, tmp_path, **kw)
        assert (rc == 0) == (rrc == 0), (args, stdin[:200], err[-500:], rerr[-500:])
        assert out == rout, (args, stdin[:200])
        assert complaint(err) == complaint(rerr), (args, stdin[:200])
    return rc, out, err


LONG = "chr1 10 20 " + "9" * 1200
LINES = {
    "valid": "chr1 10 20 3\nchr2 0 500 1.5\nchr1 15 30 2\n",
    "no_end": "chr1 10\n",
    "only_chrom": "chr1\n",
    "text_start": "chr1 x 20 1\n",
    "text_end": "chr1 10 2y0 1\n",
    "negative_start": "chr1 -5 20 1\n",
    "overflowing_end": "chr1 5 99999999999 1\n",
    "u32_max_end": "chr1 5 4294967295 1\n",
    "end_before_start": "chr1 50 20 1\n",
    "empty_interval": "chr1 50 50 1\nchr1 60 70 2\n",
    "beyond_chromosome": "chr1 990 1010 1\n",
    "missing_value": "chr1 10 20\n",
    "text_value": "chr1 10 20 abc\n",
    "odd_values": "chr1 10 20 1e3\nchr1 20 30 +5\nchr1 30 40 .5\nchr1 40 50 5.\nchr1 50 60 -0\nchr1 60 70 1e-320\n",
    "hex_value": "chr1 10 20 0x10\n",
    "nan_value": "chr1 10 20 nan\n",
    "inf_value": "chr1 10 20 inf\nchr1 30 40 -inf\n",
    "huge_value": "chr1 10 20 1e999\n",
    "line_too_long": LONG + "\n",
    "line_of_999": ("chr1 10 20 1 " + "#" * 1000)[:999] + "\n",
    "line_of_1000": ("chr1 10 20 1 " + "#" * 1000)[:1000] + "\nchr1 1 2 3\n",
    "no_final_newline": "chr1 10 20 3\nchr1 15 30 2",
    "crlf": "chr1 10 20 3\r\nchr1 15 30 2\r\n",
    "tabs_and_blanks": "chr1\t10\t20\t3\nchr1   15    30  2\n  chr1 40 50 1\n\t\n",
    "comments_tracks_blanks": "# hello\n\ntrack name=x\nchr1 10 20 3\n   # indented\nbrowser position\n",
    "unknown_chromosome": "chrZ 10 20 3\nchr1 10 20 3\n",
    "nul_byte": b"chr1 10 20 3\nchr1 1\x005 30 2\n",
    "high_bytes": b"chr1 10 20 3\n\xff\xfe\x80 10 20 1\nchr\xc3\xa9 1 2 3\n",
    "extra_columns": "chr1 10 20 3 foo bar 7\n",
    "leading_zeros_plus": "chr1 +010 0020 3\n",
    "float_coordinates": "chr1 10.5 20 3\n",
    "only_whitespace": "   \n\t\t\n",
    "empty": "",
}


@pytest.mark.parametrize("name", list(LINES))
@pytest.mark.parametrize("flags", [[], ["--novalue"], ["--cliptochromosome", "--origin=one"], ["--value=5", "--precision=3", "--uncovered:NA"]])
def test_interval_lines(name, flags, tmp_path):
    same_as_reference(flags, LINES[name], tmp_path)


CHROMS = {
    "valid": "chr1 1000\nchr2 500\n",
    "comment_blank": "# genome\n\nchr1 1000\n",
    "no_length": "chr1\n",
    "text_length": "chr1 abc\n",
    "negative_length": "chr1 -5\n",
    "zero_length": "chr1 0\nchr2 500\n",
    "duplicate": "chr1 100\nchr1 200\n",
    "long_line": "chr1 100 " + "x" * 1100 + "\n",
    "no_final_newline": "chr1 1000",
    "crlf": "chr1 1000\r\nchr2 500\r\n",
    "empty": "",
    "extra_columns": "chr1 1000 whatever\n",
}


@pytest.mark.parametrize("name", list(CHROMS))
def test_chromosome_files(name, tmp_path):
    same_as_reference(["--novalue"], "chr1 10 20\nchr2 5 9\n", tmp_path, chroms=CHROMS[name])


OPTIONS = [["--precision=-1"], ["--precision=x"], ["--precision=400"], ["--value=0"], ["--value=2"], ["--value=-3"], ["--value=x"],
           ["--window=0"], ["W=-4"], ["W=10k"], ["W=1.5M"], ["--nosuch"], ["chrQ:100"], ["chrQ:10:100"], ["chrQ"], ["chrQ:"], ["chrQ:x"],
           ["chrQ:100", "chrQ:200"], ["="], ["=", "nosuch"], ["=nosuch"], ["=", "binarize", "--nosuch"], ["=", "binarize", "x"],
           ["=", "addconst"], ["=", "addconst", "1", "2"], ["=", "smooth", "W=0"], ["=", "smooth", "W=-1"], ["=", "smooth", "W=99999999999"],
           ["=", "dilate"], ["=", "dilate", "-5"], ["=", "erode", "--left=x"], ["=", "percentile"], ["--nooutput", "=", "percentile", "101"],
           ["--nooutput", "=", "percentile", "50..40"], ["=", "percentile", "10..90by0"], ["=", "clip"], ["=", "erase", "--keep:nosuch"],
           ["=", "localmax", "N=0"], ["=", "bestmax", "W=1"], ["=", "sum", "W=2"], ["=", "slidingsum", "--denom=x"],
           ["=", "clump"], ["=", "clump", "1", "--length=0"], ["=", "map"], ["=", "add"], ["=", "input"], ["=", "output"],
           ["--uncovered:nosuch"], ["--origin=2"], ["--progress=input:x"], ["=", "abs", "=", "addconst", "2", "=", "binarize", "1"]]


@pytest.mark.parametrize("opts", OPTIONS, ids=[" ".join(o) for o in OPTIONS])
def test_command_lines(opts, tmp_path):
    chroms = None if any(o.startswith("chrQ") for o in opts) else "chr1 1000\n"
    same_as_reference(["--novalue"] + opts, "chr1 10 20\nchrQ 5 9\n", tmp_path, chroms=chroms)


def test_files_given_to_operators(tmp_path):
    for text in (LINES["valid"], LINES["text_start"], LINES["beyond_chromosome"], LINES["line_too_long"], LINES["no_final_newline"],
                 LINES["crlf"], LINES["nul_byte"], ""):
        for op in (["=", "input", "@f@"], ["=", "add", "@f@"], ["=", "input", "@f@", "--missing=2", "--value=4"],
                   ["=", "add", "@f@", "--novalue", "--origin=one"], ["=", "subtract", "@nosuch@"]):
            same_as_reference(["--precision=2"] + op, "chr1 100 200 1\n", tmp_path, files={"f": text})


def test_mutated_lines_fuzz(tmp_path):
    """seeded byte-level mutations of valid input (deletions, insertions, swaps, digit runs, separators): 400 files"""
    rng = np.random.default_rng(20240611)
    alphabet = b"0123456789 \t\n.-+eEx#chr12\r\x00\xff"
    base = b"".join(b"chr%d\t%d\t%d\t%d\n" % (1 + i % 2, 10 * i, 10 * i + 25, i % 7) for i in range(40))
    for k in range(400):
        b = bytearray(base)
        for _ in range(int(rng.integers(1, 6))):
            pos = int(rng.integers(0, len(b)))
            how = int(rng.integers(0, 5))
            if how == 0:
                del b[pos:pos + int(rng.integers(1, 9))]
            elif how == 1:
                b[pos:pos] = bytes(rng.choice(list(alphabet), int(rng.integers(1, 12))).astype(np.uint8))
            elif how == 2:
                b[pos] = int(rng.choice(list(alphabet)))
            elif how == 3:
                b[pos:pos] = b"9" * int(rng.integers(5, 30))
            else:
                b[pos:pos] = b"x" * int(rng.choice([990, 1000, 1010, 3000]))
        flags = [[], ["--novalue"], ["--cliptochromosome"], ["--value=5"]][k % 4]
        same_as_reference(flags, bytes(b), tmp_path)


def test_output_formatting_extremes(tmp_path):
    """put_fixed (the hand-made %.*f of the report) against the reference's printf on awkward values and precisions"""
    vals = ["0.5", "1.5", "2.5", "0.125", "0.0625", "1e-7", "123456789.987654321", "1e15", "1e17", "1e22", "4.35", "0.285", "1e-300",
            "9.999999", "99999.99995", "0.045", "-0.5", "-2.5", "-1e-9", "1.7976931348623157e308", "5e-324", "0.1", "0.7", "2.675"]
    text = "".join("chr1 %d %d %s\n" % (3 * i, 3 * i + 2, v) for i, v in enumerate(vals))
    for prec in (0, 1, 2, 3, 6, 9, 10, 15, 17, 25):
        same_as_reference(["--precision=%d" % prec], text, tmp_path)


def test_committed_command_lines_under_the_sanitizers(tmp_path):
    """every CLI fixture recorded from the reference (tests/golden/golden.json: 40 with their full stdout, 88 random
    command lines as digests) through the sanitized host code: option and operator parsing, ingest, the apply shims,
    scratch and named-variable bookkeeping, the report's formatting -- same bytes out as the reference's"""
    import hashlib
    from conftest import golden
    ran = 0
    for case in golden().meta["cases"]:
        if case["kind"] not in ("cli", "cli_digest") or case["returncode"] != 0:
            continue
        if case["name"] in ("cli_percentile99", "cli_percentile_extremes_0_map"):
            continue                                         # (the reference prints its sort-scrambled signal there: tests/test_cli_hip.py)
        rc, out, err = run(BIN, case["args"], case["stdin"], tmp_path, chroms=case["chroms_text"], files=case.get("files"))
        assert rc == 0 and "Sanitizer" not in err and "runtime error" not in err, (case["name"], err[-2000:])
        if case["kind"] == "cli":
            assert out.decode() == case["stdout"], case["name"]
        else:
            assert hashlib.sha256(out).hexdigest() == case["sha256"], (case["name"], case["args"])
        ran += 1
    assert ran >= 120


def test_seam_crossing_command_lines_under_the_sanitizers(tmp_path):
    """golden_seams.json.gz (chromosomes of 120-270 kbp, thousands of intervals; and the running-sum pipelines) through the
    sanitized host code: ingest batches and report chunks at size.  The stub's operators are the oracle's sequential
    loops, so here even the running sums print the reference's bytes."""
    import gzip
    import hashlib
    import json
    from conftest import GOLDEN_DIR
    with gzip.open(os.path.join(GOLDEN_DIR, "golden_seams.json.gz")) as f:
        cases = json.load(f)["cases"]
    for case in cases:
        rc, out, err = run(BIN, case["args"], case["stdin"], tmp_path, chroms=case["chroms_text"], files=case.get("files"))
        assert rc == 0 and "Sanitizer" not in err and "runtime error" not in err, (case["name"], err[-2000:])
        assert hashlib.sha256(out).hexdigest() == case["sha256"], (case["name"], case["args"])


def test_threaded_output_formatting_under_the_sanitizers(tmp_path):
    """more than 65 536 output lines: the report is formatted by a team of threads (format_runs, genodsp_hip.c), every
    stretch into its own buffer -- the same bytes as the reference prints, NA lines and a value that needs printf's
    slow path included, whatever the team's size"""
    chroms = "chrT 150000\nchrU 300\n"
    text = "chrT 100 70000 3\nchrT 70000 70001 1e300\nchrT 70500 149000 2.5\nchrU 10 20 1\n"
    for flags in (["--nocollapse", "--precision=2"], ["--nocollapse", "--uncovered:NA", "--precision=0"],
                  ["--nocollapse", "--precision=12", "--origin=one"]):
        outs = []
        for threads in ("1", "4", "7"):
            ENV["GDSP_OUTPUT_THREADS"] = threads
            try:
                rc, out, err = same_as_reference(flags, text, tmp_path, chroms=chroms)
            finally:
                del ENV["GDSP_OUTPUT_THREADS"]
            assert rc == 0 and out.count(b"\n") > 140000
            outs.append(out)
        assert outs[0] == outs[1] == outs[2]
