"""The dspop plugin surface: include/genodsp_interface.h + include/utilities.h hold every name an
operator group of the reference uses -- the five-function groups, dspinforecord, the sixteen host services of
genodsp_interface.h:167-190 -- so such a group compiles against include/ unchanged and is linked into the
driver's table the way the reference links its own (a link-time function group).

CPU: (1) where /root/reference exists, every operator file of the reference passes
`gcc -fsyntax-only -Wall -Wextra -Werror` against include/ (its own sources are read where they lie, nothing is
copied; the files dereference v on the host, so they are compiled, never linked or run); (2) a small group written
in the reference's style (tests/plugin/demo_ops.c: scratch vector, scratch ints, named globals, read/write_all,
valtype_ascending, chastise, tracking_report) compiles with -Werror and links into a driver binary.
GPU: that driver runs the group in a pipeline and prints what the same pipeline must print.
"""
import os
import shutil
import subprocess

import numpy as np
import pytest

from conftest import ROOT

INCLUDE = os.path.join(ROOT, "include")
PLUGIN = os.path.join(ROOT, "tests", "plugin")
HOST = os.path.join(ROOT, "genodsp_amd", "host")
REFERENCE = "/root/reference"
REF_OPERATOR_FILES = ["sum", "clump", "percentile", "add", "multiply", "mask", "logical", "minmax", "morphology",
                      "map", "opio", "variables"]
SERVICES = ["chastise", "find_chromosome_spec", "read_intervals", "read_interval", "report_intervals",
            "read_all_chromosomes", "write_all_chromosomes", "get_scratch_vector", "get_scratch_ints",
            "release_scratch_vector", "release_scratch_ints", "set_named_global", "get_named_global",
            "named_global_exists", "report_named_globals", "tracking_report", "valtype_ascending"]


@pytest.mark.skipif(not os.path.isdir(REFERENCE), reason="the reference's sources exist only in the build container")
@pytest.mark.parametrize("name", REF_OPERATOR_FILES)
def test_reference_operator_file_compiles_against_our_headers(name, tmp_path):
    # fed on stdin from an empty directory, so that "utilities.h" / "genodsp_interface.h" resolve to include/
    # (first -I) and only the operator's own header (sum.h ...) comes from the reference
    with open(os.path.join(REFERENCE, name + ".c"), "rb") as src:
        p = subprocess.run(["gcc", "-x", "c", "-std=gnu99", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-H",
                            "-I", INCLUDE, "-I", REFERENCE, "-"], stdin=src, capture_output=True, cwd=str(tmp_path))
    err = p.stderr.decode()
    assert p.returncode == 0, err
    used = [l.split()[-1] for l in err.splitlines() if l.startswith(".") and l.split()[-1].endswith(".h")]
    assert os.path.join(INCLUDE, "genodsp_interface.h") in used and os.path.join(INCLUDE, "utilities.h") in used
    assert os.path.join(REFERENCE, "genodsp_interface.h") not in used and os.path.join(REFERENCE, "utilities.h") not in used


def test_every_host_service_is_declared_and_defined():
    header = open(os.path.join(INCLUDE, "genodsp_interface.h")).read()
    driver = os.path.join(ROOT, "genodsp_amd", "genodsp_hip")
    if not os.path.exists(driver):
        subprocess.check_call(["make", "-s", "-C", HOST])
    syms = subprocess.run(["nm", "--defined-only", driver], capture_output=True, text=True, check=True).stdout
    defined = {l.split()[-1] for l in syms.splitlines() if l.split()[1:2] == ["T"]}
    for s in SERVICES:
        assert (s + " ") in header or (s + "(") in header, s + " is not declared in include/genodsp_interface.h"
        assert s in defined, s + " is not defined by the driver"


def build_demo_driver(outdir):
    out = os.path.join(str(outdir), "genodsp_hip_demo")
    subprocess.check_call(["make", "-s", "-C", HOST, "OUT=" + out,
                           "EXTRA_OPS_HEADER=" + os.path.join(PLUGIN, "demo_ops.h"),
                           "EXTRA_OPS_SRCS=" + os.path.join(PLUGIN, "demo_ops.c"),
                           "CFLAGS=-O2 -std=gnu99 -Wall -Wextra -Werror -Wno-unused-parameter -I%s -I%s" % (INCLUDE, PLUGIN)])
    return out


def test_group_in_the_reference_style_compiles_and_links(tmp_path):
    subprocess.check_call(["gcc", "-std=gnu99", "-Wall", "-Wextra", "-Werror", "-I", INCLUDE, "-I", PLUGIN, "-c",
                           os.path.join(PLUGIN, "demo_ops.c"), "-o", os.path.join(str(tmp_path), "demo_ops.o")])
    out = build_demo_driver(tmp_path)
    syms = subprocess.run(["nm", "--defined-only", out], capture_output=True, text=True, check=True).stdout
    for s in ("op_demo_lift_apply", "op_demo_snapshot_parse"):
        assert s in syms


@pytest.mark.gpu
def test_group_runs_in_the_pipeline_on_the_gpu(tmp_path):
    binary = build_demo_driver(tmp_path)
    chroms = [("chrA", 5000), ("chrB", 1200)]
    rng = np.random.default_rng(5)
    want = {c: np.zeros(n) for c, n in chroms}
    lines = []
    for c, n in chroms:
        for _ in range(60):
            a = int(rng.integers(0, n - 200))
            b = a + int(rng.integers(1, 200))
            val = int(rng.integers(-4, 5))
            lines.append("%s\t%d\t%d\t%d" % (c, a, b, val))
            want[c][a:b] += val
    snap = os.path.join(str(tmp_path), "snapshot.dat")
    args = [binary] + ["%s:%d" % cn for cn in chroms] + ["--progress=operations", "=", "percentile", "100", "--quiet",
            "=", "demo_lift", "percentile100", "=", "demosnapshot", snap, "=", "variables"]
    p = subprocess.run(args, input="\n".join(lines) + "\n", capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr
    top = max(float(v.max()) for v in want.values())
    expect = []
    for c, n in chroms:
        v = np.abs(want[c]) + top
        edges = np.flatnonzero(np.diff(v)) + 1
        starts = np.concatenate([[0], edges])
        ends = np.concatenate([edges, [n]])
        expect += ["%s\t%d\t%d\t%d" % (c, s, e, v[s]) for s, e in zip(starts, ends) if v[s] != 0]
    assert p.stdout.splitlines() == expect
    err = p.stderr
    assert "demosnapshot(%s)" % snap in err                                 # tracking_report
    for text in ("demoLength_chrA = 5000", "demoLength_chrB = 1200", "demoShortest = 1200", "demoLongest = 5000"):
        assert text in " ".join(err.split()), err                              # set_named_global -> `variables`
    assert len(open(snap).read().splitlines()) == len(expect)                  # write_all_chromosomes: the same runs, 10 decimals
    q = subprocess.run([binary, "chrA:100", "=", "demolift", "--bogus"], input="", capture_output=True, text=True)
    assert q.returncode != 0 and "Can't understand" in q.stderr and "usage: demolift" in q.stderr   # chastise + usage


def test_base_sharding_plan_is_even_at_eight_devices():
    """--sharding=bases --shards=show (no GPU needed): the hg38-like genome of BASELINE.json over 8 devices.  Whole
    chromosomes dealt longest-first reach 0.965 of a perfect split; equal shares of the bases,
    chromosomes cut where needed and halos counted as work, must reach 0.995 or better, every base owned once."""
    import re
    genome = [("chr1", 248956422), ("chr2", 242193529), ("chr3", 198295559), ("chr4", 190214555), ("chr5", 181538259),
              ("chr6", 170805979), ("chr7", 159345973), ("chr8", 145138636), ("chr9", 138394717), ("chr10", 133797422),
              ("chr11", 135086622), ("chr12", 133275309), ("chr13", 114364328), ("chr14", 107043718), ("chr15", 101991189),
              ("chr16", 90338345), ("chr17", 83257441), ("chr18", 80373285), ("chr19", 58617616), ("chr20", 64444167),
              ("chr21", 46709983), ("chr22", 50818468), ("chrX", 156040895), ("chrY", 57227415)]
    driver = os.path.join(ROOT, "genodsp_amd", "genodsp_hip")
    eff = {}
    for how in ("chromosomes", "bases"):
        for gpus in (2, 4, 8):
            p = subprocess.run([driver] + ["%s:%d" % c for c in genome] + ["--gpus=%d" % gpus, "--sharding=" + how, "--shards=show",
                               "=", "dilate", "1001", "=", "erode", "1001", "=", "binarize"], capture_output=True, text=True)
            assert p.returncode == 0, p.stderr
            eff[how, gpus] = float(re.search(r"makespan efficiency ([0-9.]+)", p.stderr).group(1))
            if how == "bases":
                owned = {}
                for c, a, b in re.findall(r" (chr\w+):(\d+)-(\d+)", p.stderr):
                    owned.setdefault(c, []).append((int(a), int(b)))
                for c, n in genome:                              # the stretches of a chromosome tile it exactly
                    at = 0
                    for a, b in sorted(owned[c]):
                        assert a == at and b > a
                        at = b
                    assert at == n
    assert abs(eff["chromosomes", 8] - 0.965) < 0.001 and abs(eff["chromosomes", 4] - 0.995) < 0.001
    assert eff["bases", 8] >= 0.995 and eff["bases", 4] >= 0.995 and eff["bases", 2] >= 0.995
