"""GPU: the HIP driver against digests of the reference binary for command lines drawn by tools/cli_campaign.py.
The file tests/golden/campaign.json is a one-off (not committed): without it there is nothing to run."""
import hashlib
import json
import os

import pytest

from test_cli_hip import run

pytestmark = pytest.mark.gpu
import glob
CASES = []
for PATH in sorted(glob.glob(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "campaign*.json"))):
    CASES += json.load(open(PATH))["cases"]


@pytest.mark.skipif(not CASES, reason="no campaign drawn (tools/cli_campaign.py)")
@pytest.mark.parametrize("case", CASES, ids=[c["name"] for c in CASES])
def test_campaign_case(case, tmp_path):
    rc, out, err = run(case["args"], case["stdin"], case["chroms_text"], tmp_path, case.get("files"))
    assert rc == 0, err
    body = out.splitlines()
    assert (len(body), body[:5], body[-3:]) == (case["lines"], case["head"], case["tail"]), case["args"]
    assert hashlib.sha256(out.encode()).hexdigest() == case["sha256"], case["args"]
    for line in case["stderr_percentile"]:
        assert line in err.splitlines()
