"""GPU, opt-in: the HIP driver against the reference binary on command lines drawn by tools/cli_campaign.py.

The draws are one-offs, not fixtures: nothing runs unless GDSP_CAMPAIGN names the files (a glob, e.g.
GDSP_CAMPAIGN='build/campaign/*.json').  Each case is compared the way its pipeline calls for
(tests/cli_compare.py): a digest of the reference's stdout, or -- `slidingsum` / `cumulativesum` behind `smooth`,
where the reference's single running accumulator (sum.c:438-455, :785-790) fixes the last bits -- base by base within
the stated bound against the reference's whole stdout, which the tool keeps for exactly those cases."""
import glob
import json
import os

import pytest

import cli_compare
from test_cli_hip import run

pytestmark = pytest.mark.gpu

CASES = []
for PATH in sorted(glob.glob(os.environ.get("GDSP_CAMPAIGN", ""))) if os.environ.get("GDSP_CAMPAIGN") else []:
    CASES += json.load(open(PATH))["cases"]


@pytest.mark.skipif(not CASES, reason="opt-in: GDSP_CAMPAIGN=<glob of files drawn by tools/cli_campaign.py>")
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_campaign_case(case, tmp_path):
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    how = cli_compare.assert_matches_reference(case, rc, out, err)
    if how == "skip":
        pytest.skip("a running sum behind smooth feeds a discontinuous operator (or the draw kept no stdout): texts not comparable")
