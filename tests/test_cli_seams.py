"""GPU: `genodsp_hip` on command lines big enough to cross every seam -- kernel tiles (2304 .. 16384 bases), ingest
batches, the report's chunks -- against the reference binary's recorded digests, and the running-sum pipelines held to
a stated bound.  tests/golden/golden_seams.json.gz (written by tests/golden/make_golden.py from the unmodified
reference): 40 random pipelines over chromosomes of 120-270 kbp with 1800-5400 intervals each (digests), 16 pipelines
with windows beyond what one LDS tile holds (extrema of 8193-30 000 bases, smooth of 4801 / 10 001 taps, morphology
reaching 8500-20 001, sums over 9000 / 20 001: the whole-vector routes) and six pipelines in which `slidingsum` /
`cumulativesum` follow `smooth` on real values, with the reference's whole output.  Every digest case runs in the
driver's default operator-major order and with --nobatch (the reference's order).

Why the six are not digests: the reference adds along the whole chromosome with ONE accumulator (sum.c:438-455,
:785-790), rounding at every base; a parallel scan associates differently, so the last bits -- and with them, now and
then, a printed digit or the base at which a sum returns to exactly zero -- can differ.  The bound: every base within
one unit of the last printed digit plus 8 n eps max|v| of what the reference printed (n = chromosome length; both
programs round the same exact sum, n roundings each, printed to `--precision` digits)."""
import gzip
import json
import os

import pytest

import cli_compare
from conftest import GOLDEN_DIR
from test_cli_hip import ORDERS, run

pytestmark = pytest.mark.gpu

with gzip.open(os.path.join(GOLDEN_DIR, "golden_seams.json.gz")) as f:
    CASES = json.load(f)["cases"]
DIGESTS = [c for c in CASES if c["kind"] == "cli_digest"]
RUNNING = [c for c in CASES if c["kind"] == "cli_running_sum"]


@pytest.mark.parametrize("order", list(ORDERS))
@pytest.mark.parametrize("case", DIGESTS, ids=[c["name"] for c in DIGESTS])
def test_seam_crossing_pipelines_match_the_reference_binary(case, order, tmp_path):
    assert case["returncode"] == 0
    rc, out, err = run(ORDERS[order] + case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    cli_compare.assert_digest(case, out, err)


@pytest.mark.parametrize("case", RUNNING, ids=[c["name"] for c in RUNNING])
def test_running_sums_behind_smooth_stay_within_the_stated_bound(case, tmp_path):
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert cli_compare.comparison(case["args"]) == "bound"
    assert cli_compare.assert_matches_reference(case, rc, out, err) == "bound"


@pytest.mark.parametrize("poison", ["1e300", "-1e300"])
@pytest.mark.parametrize("case", DIGESTS[::3], ids=[c["name"] for c in DIGESTS[::3]])
def test_seam_crossing_pipelines_do_not_read_memory_nobody_wrote(case, poison, tmp_path, monkeypatch):
    """as tests/test_cli_hip.py::test_no_operator_reads_memory_nobody_wrote, at sizes where tiles meet their seams and
    on the whole-vector routes of the long windows"""
    monkeypatch.setenv("GDSP_POISON", poison)
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    cli_compare.assert_digest(case, out, err)


@pytest.mark.parametrize("sharding", ["chromosomes", "bases"])
@pytest.mark.parametrize("case", DIGESTS[1::6], ids=[c["name"] for c in DIGESTS[1::6]])
def test_seam_crossing_pipelines_over_eight_shards(case, sharding, tmp_path, monkeypatch):
    """the contract's device count (BASELINE: 8 x MI355X) rehearsed on the one GPU: two or three chromosomes dealt over
    eight devices leave most of them empty (--sharding=chromosomes), eight stretches of bases cut every chromosome
    (--sharding=bases); either way the reference binary's bytes"""
    monkeypatch.setenv("GDSP_OVERSUBSCRIBE_GPUS", "1")
    rc, out, err = run(["--gpus=8", "--sharding=" + sharding] + case["args"], case["stdin"], case["chroms_text"], tmp_path,
                       case.get("files"))
    assert rc == 0, err
    cli_compare.assert_digest(case, out, err)
