#!/usr/bin/env python3
"""A bigger draw from the random command-line generators of tests/golden/make_golden.py, as a one-off hunt: what the
reference binary prints for seeds the committed fixtures do not hold.  Not fixtures and not shipped: the files go to
gpurun_out/campaign/ (scratch); for a run on the GPU box copy them somewhere that travels and name them,
  mkdir -p build/campaign && cp gpurun_out/campaign/*.json build/campaign/
  gpurun -- 'GDSP_CAMPAIGN="build/campaign/*.json" python -m pytest tests/test_cli_campaign.py -q -m gpu'
tests/test_cli_campaign.py compares each case the way its pipeline calls for (tests/cli_compare.py); the reference's
whole stdout is kept only for the cases held to a bound (a running sum behind smooth), digests for the rest.
usage (in the build container, where /root/reference exists): python3 tools/cli_campaign.py [first] [count] [scale] [file]"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cli_compare  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # chromosomes and interval counts of the per-base cases, times this
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)              # (regenerates the committed fixtures on the way: same bytes)
before = len(mod.cases)
for k in range(first, first + count):
    mod.random_cli_case(k, scale, keep_stdout=True)
    if cli_compare.comparison(mod.cases[-1]["args"]) != "bound":
        del mod.cases[-1]["stdout"]
    if scale == 1:
        mod.random_file_case(k)
extra = [c for c in mod.cases[before:] if c["returncode"] == 0]
out_name = sys.argv[4] if len(sys.argv) > 4 else "campaign.json"
os.makedirs(os.path.join(ROOT, "gpurun_out", "campaign"), exist_ok=True)
with open(os.path.join(ROOT, "gpurun_out", "campaign", out_name), "w") as f:
    json.dump({"first": first, "count": count, "scale": scale, "cases": extra}, f)
print("%d cases" % len(extra))
