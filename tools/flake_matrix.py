#!/usr/bin/env python3
"""The random command lines of tests/golden/golden.json, over and over, in a matrix of ways to run the driver, with the
GPU kept busy by other processes -- the hunt for the one wrong digest of round 3 (`= bestmax W=4`).

  python3 tools/flake_matrix.py [passes per cell, default 20] [workers, default 4] [cells, comma separated, default all]

Cells: default | nobatch (--nobatch: the reference's order) | ingest1 (GDSP_INGEST_THREADS=1) | output1
(GDSP_OUTPUT_THREADS=1) | serialize (AMD_SERIALIZE_KERNEL=3) | poison (GDSP_POISON=nan: every allocation and every
flipped partner refilled with NaN) | nostream (a library built with -DGDSP_STREAMING=0: plain loads and stores; needs
build/variant_nostream/, see tools/flake_matrix.sh).
Load: `workers` command lines run at once, and one more process streams `smooth` launches through the same GPU for
the whole run (the driver's boxes are shared the same way a test suite shares them with itself).
Every run whose digest differs (or whose exit status is not 0) leaves argv, environment, stdin, stdout and stderr under
gpurun_out/artifacts/flake/<cell>/<case>_<k>/; the table goes to stdout."""
import concurrent.futures as cf
import hashlib
import json
import os
import pathlib
import subprocess
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import cli_compare  # noqa: E402

BIN = os.path.join(ROOT, "genodsp_amd", "genodsp_hip")
VARIANT = os.path.join(ROOT, "build", "variant_nostream", "genodsp_hip")
CELLS = {
    "default": ([], {}, BIN),
    "nobatch": (["--nobatch"], {}, BIN),
    "ingest1": ([], {"GDSP_INGEST_THREADS": "1"}, BIN),
    "output1": ([], {"GDSP_OUTPUT_THREADS": "1"}, BIN),
    "serialize": ([], {"AMD_SERIALIZE_KERNEL": "3"}, BIN),
    "poison": ([], {"GDSP_POISON": "nan"}, BIN),
    "nostream": ([], {}, VARIANT),
}
ART = os.path.join(ROOT, "gpurun_out", "artifacts", "flake")


def one(cell, case, k):
    extra, env_add, binary = CELLS[cell]
    env = dict(os.environ)
    env.update(env_add)
    with tempfile.TemporaryDirectory() as t:
        args = list(case["args"])
        for key, text in (case.get("files") or {}).items():
            path = os.path.join(t, key + ".dat")
            pathlib.Path(path).write_text(text)
            args = [a.replace("@%s@" % key, path) for a in args]
        chroms = os.path.join(t, "genome.chroms")
        pathlib.Path(chroms).write_text(case["chroms_text"])
        argv = [binary, "--chromosomes=" + chroms] + extra + args
        p = subprocess.run(argv, input=case["stdin"], capture_output=True, text=True, timeout=300, env=env)
    ok = (p.returncode == 0)
    if ok:
        how = cli_compare.comparison(case["args"])
        if how == "digest":
            ok = hashlib.sha256(p.stdout.encode()).hexdigest() == case["sha256"]
        elif how == "bound" and "stdout" in case:
            try:
                cli_compare.assert_within_running_sum_bound(case, p.stdout)
            except AssertionError:
                ok = False
    if not ok:
        d = os.path.join(ART, cell, "%s_%d" % (case["name"], k))
        os.makedirs(d, exist_ok=True)
        for name, text in (("stdin", case["stdin"]), ("stdout", p.stdout), ("stderr", p.stderr)):
            pathlib.Path(os.path.join(d, name)).write_text(text or "")
        json.dump({"argv": argv, "env": env_add, "rc": p.returncode, "want_sha256": case.get("sha256"),
                   "got_sha256": hashlib.sha256(p.stdout.encode()).hexdigest(), "files": case.get("files") or {}},
                  open(os.path.join(d, "meta.json"), "w"), indent=1)
    return ok


def main():
    passes = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    workers = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    cells = sys.argv[3].split(",") if len(sys.argv) > 3 else list(CELLS)
    meta = json.load(open(os.path.join(ROOT, "tests", "golden", "golden.json")))
    cases = [c for c in meta["cases"] if c["kind"] == "cli_digest" and c["returncode"] == 0]
    print("library: %s" % cli_compare.library_id())
    print("%d random command lines x %d passes per cell, %d at once, one more process streaming smooth launches" % (len(cases), passes, workers))
    load = subprocess.Popen([sys.executable, os.path.join(ROOT, "tools", "gpu_load.py")], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        print("%-10s %8s %8s %8s   %s" % ("cell", "runs", "bad", "seconds", "cases that differed"))
        for cell in cells:
            if not os.path.exists(CELLS[cell][2]):
                print("%-10s not built (%s)" % (cell, CELLS[cell][2]))
                continue
            t0 = time.time()
            bad = {}
            with cf.ThreadPoolExecutor(workers) as pool:
                jobs = {pool.submit(one, cell, c, k): (c["name"], k) for k in range(passes) for c in cases}
                for j in cf.as_completed(jobs):
                    if not j.result():
                        bad[jobs[j][0]] = bad.get(jobs[j][0], 0) + 1
            print("%-10s %8d %8d %8.0f   %s" % (cell, len(jobs), sum(bad.values()), time.time() - t0, json.dumps(bad) if bad else "-"), flush=True)
        print("load process alive at the end: %s" % (load.poll() is None))
    finally:
        load.terminate()
        try:
            load.wait(30)
        except subprocess.TimeoutExpired:
            load.kill()


if __name__ == "__main__":
    main()
