"""Shared by the CLI tests (-m gpu): how a command line's stdout is compared with what the reference binary printed,
and what is kept when it differs.

Three classes of pipeline (`comparison(args)`):
  "digest"  every byte must be the reference's (sha256, line count, first and last lines);
  "bound"   `slidingsum` / `cumulativesum` behind `smooth`: the reference adds along the whole chromosome with ONE
            accumulator (sum.c:438-455, :785-790), rounding at every base; a parallel scan associates differently, so
            the last bits -- and now and then a printed digit, or the base at which a sum returns to exactly zero --
            differ.  Held base by base to one unit of the last printed digit plus 8 n eps max|v| (n = chromosome
            length), as long as every later operator is continuous (it cannot turn a last-bit difference into another
            answer).  Needs the reference's whole stdout in the fixture;
  "skip"    such a running sum followed by a discontinuous operator (a comparison against a threshold or a
            neighbour): the two texts are not comparable line by line, nothing is asserted.

Artifacts: every `run()` of a test is remembered (tests/test_cli_hip.py); when the test fails, tests/conftest.py
writes argv, GDSP_* environment, the library's id, stdin, stdout and stderr of each run to
gpurun_out/artifacts/<test id>/ -- the directory gpurun brings back -- so that a digest that differs once leaves
something to reason from."""
import hashlib
import json
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ARTIFACTS = os.path.join(ROOT, "gpurun_out", "artifacts")
RUNNING = {"slidingsum", "cumulativesum"}
CONTINUOUS = {"smooth", "slidingsum", "cumulativesum", "sum", "bestmin", "bestmax", "addconst", "abs", "invert", "clip"}
KEEP_BYTES = 32 << 20

RUNS = []          # what run() did during the current test: dicts of argv / env / stdin / rc / stdout / stderr


def operators(args):
    """operator names of a command line, in pipeline order (`= name args…`, the `=` glued or apart)"""
    ops, take = [], False
    for a in args:
        if a == "=":
            take = True
        elif a.startswith("="):
            ops.append(a[1:])
            take = False
        elif take:
            ops.append(a)
            take = False
    return ops


def comparison(args):
    ops = operators(args)
    if "smooth" not in ops:
        return "digest"
    after = ops[ops.index("smooth") + 1:]
    first = next((i for i, o in enumerate(after) if o in RUNNING), None)
    if first is None:
        return "digest"
    return "bound" if all(o in CONTINUOUS for o in after[first + 1:]) else "skip"


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


def assert_within_running_sum_bound(case, out):
    precision = int([a for a in case["args"] if a.startswith("--precision=")][0].split("=")[1])
    got, want = per_base(out, case["chroms_text"], case["args"]), per_base(case["stdout"], case["chroms_text"], case["args"])
    for chrom in want:
        n = want[chrom].size
        bound = 10.0 ** -precision + 8 * n * 2.0 ** -52 * max(1.0, float(np.abs(want[chrom]).max()))
        worst = float(np.abs(got[chrom] - want[chrom]).max())
        assert worst <= bound, (case["args"], chrom, worst, bound)


def assert_digest(case, out, err=""):
    body = out.splitlines()
    assert (len(body), body[:5], body[-3:]) == (case["lines"], case["head"], case["tail"]), case["args"]
    assert hashlib.sha256(out.encode()).hexdigest() == case["sha256"], case["args"]
    for line in case.get("stderr_percentile", []):
        assert line in err.splitlines()


def assert_matches_reference(case, rc, out, err):
    """the comparison the case's pipeline calls for; -> the class used"""
    assert rc == 0, err
    how = comparison(case["args"])
    if how == "digest":
        assert_digest(case, out, err)
    elif how == "bound" and "stdout" in case:
        assert_within_running_sum_bound(case, out)
    else:
        how = "skip"
    return how


def library_id():
    try:
        import subprocess
        p = subprocess.run([os.path.join(ROOT, "genodsp_amd", "genodsp_hip"), "--version"], capture_output=True, text=True, timeout=60)
        return (p.stderr + p.stdout).strip()
    except Exception as e:                                          # noqa: BLE001 (an artifact writer must not raise)
        return "unknown (%s)" % e


def remember(argv, env, stdin_text, rc, out, err, files=None):
    RUNS.append({"argv": list(argv), "files": dict(files or {}), "env": {k: v for k, v in sorted((env or os.environ).items()) if k.startswith(("GDSP_", "HIP_", "HSA_", "AMD_", "ROCR_"))},
                 "stdin": stdin_text, "rc": rc, "stdout": out, "stderr": err})


def write_artifacts(test_id):
    """called by conftest when a test fails: everything run() saw during it"""
    if not RUNS:
        return None
    d = os.path.join(ARTIFACTS, re.sub(r"[^A-Za-z0-9_.\-]+", "_", test_id)[-150:])
    os.makedirs(d, exist_ok=True)
    meta = {"test": test_id, "library": library_id(), "runs": []}
    for k, r in enumerate(RUNS):
        for part in ("stdin", "stdout", "stderr"):
            text = r[part] if isinstance(r[part], str) else ""
            with open(os.path.join(d, "run%d.%s" % (k, part)), "w") as f:
                f.write(text[:KEEP_BYTES])
        meta["runs"].append({"argv": r["argv"], "env": r["env"], "rc": r["rc"], "files": r["files"],
                             "stdout_sha256": hashlib.sha256((r["stdout"] or "").encode()).hexdigest(),
                             "stdout_lines": len((r["stdout"] or "").splitlines())})
    with open(os.path.join(d, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)
    return d
