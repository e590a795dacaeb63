#!/usr/bin/env python3
"""A bigger draw from the random command-line generators of tests/golden/make_golden.py, as a one-off hunt: the
reference binary's digests for seeds the committed fixtures do not hold go to tests/golden/campaign.json (not
committed; .gitignore), and tests/test_cli_campaign.py compares the HIP driver with them when the file is there.
usage (in the build container, where /root/reference exists): python3 tools/cli_campaign.py [first] [count] [scale] [file]
(file: name under tests/golden, default campaign.json; tests/test_cli_campaign.py reads every campaign*.json there)"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
first = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
count = int(sys.argv[2]) if len(sys.argv) > 2 else 300
scale = int(sys.argv[3]) if len(sys.argv) > 3 else 1      # chromosomes and interval counts of the per-base cases, times this
spec = importlib.util.spec_from_file_location("make_golden", os.path.join(ROOT, "tests", "golden", "make_golden.py"))
mod = importlib.util.module_from_spec(spec)
spec.loader.exec_module(mod)              # (regenerates the committed fixtures on the way: same bytes)
before = len(mod.cases)
for k in range(first, first + count):
    mod.random_cli_case(k, scale)
    if scale == 1:
        mod.random_file_case(k)
extra = [c for c in mod.cases[before:] if c["returncode"] == 0]
out_name = sys.argv[4] if len(sys.argv) > 4 else "campaign.json"
with open(os.path.join(ROOT, "tests", "golden", out_name), "w") as f:
    json.dump({"first": first, "count": count, "scale": scale, "cases": extra}, f)
print("%d cases" % len(extra))
