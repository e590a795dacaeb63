"""GPU: `genodsp_hip` on command lines big enough to cross every seam -- kernel tiles (2304 .. 16384 bases), ingest
batches, the report's chunks -- against the reference binary's recorded digests, and the running-sum pipelines held to
a stated bound.  tests/golden/golden_seams.json.gz (written by tests/golden/make_golden.py from the unmodified
reference): 40 random pipelines over chromosomes of 120-270 kbp with 1800-5400 intervals each (digests), and six
pipelines in which `slidingsum` / `cumulativesum` follow `smooth` on real values, with the reference's whole output.

Why the six are not digests: the reference adds along the whole chromosome with ONE accumulator (sum.c:438-455,
:785-790), rounding at every base; a parallel scan associates differently, so the last bits -- and with them, now and
then, a printed digit or the base at which a sum returns to exactly zero -- can differ.  The bound: every base within
one unit of the last printed digit plus 8 n eps max|v| of what the reference printed (n = chromosome length; both
programs round the same exact sum, n roundings each, printed to `--precision` digits)."""
import gzip
import hashlib
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from test_cli_hip import run

pytestmark = pytest.mark.gpu

with gzip.open(os.path.join(GOLDEN_DIR, "golden_seams.json.gz")) as f:
    CASES = json.load(f)["cases"]
DIGESTS = [c for c in CASES if c["kind"] == "cli_digest"]
RUNNING = [c for c in CASES if c["kind"] == "cli_running_sum"]


@pytest.mark.parametrize("case", DIGESTS, ids=[c["name"] for c in DIGESTS])
def test_seam_crossing_pipelines_match_the_reference_binary(case, tmp_path):
    assert case["returncode"] == 0
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    body = out.splitlines()
    assert (len(body), body[:5], body[-3:]) == (case["lines"], case["head"], case["tail"]), case["args"]
    assert hashlib.sha256(out.encode()).hexdigest() == case["sha256"], case["args"]
    for line in case["stderr_percentile"]:
        assert line in err.splitlines()


def per_base(text, chroms_text, args):
    """the printed signal, base by base (what is not printed is zero)"""
    origin = 1 if "--origin=one" in args else 0
    out = {}
    for line in chroms_text.splitlines():
        name, n = line.split()
        out[name] = np.zeros(int(n))
    for line in text.splitlines():
        f = line.split("\t")
        v = 1.0 if len(f) < 4 else (0.0 if f[3] == "NA" else float(f[3]))
        out[f[0]][int(f[1]) - origin:int(f[2])] = v
    return out


@pytest.mark.parametrize("case", RUNNING, ids=[c["name"] for c in RUNNING])
def test_running_sums_behind_smooth_stay_within_the_stated_bound(case, tmp_path):
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    precision = int([a for a in case["args"] if a.startswith("--precision=")][0].split("=")[1])
    got, want = per_base(out, case["chroms_text"], case["args"]), per_base(case["stdout"], case["chroms_text"], case["args"])
    for chrom in want:
        n = want[chrom].size
        bound = 10.0 ** -precision + 8 * n * 2.0 ** -52 * max(1.0, float(np.abs(want[chrom]).max()))
        worst = float(np.abs(got[chrom] - want[chrom]).max())
        assert worst <= bound, (case["args"], chrom, worst, bound)
