"""GPU: the drop-in itself at the size the contract names -- `genodsp_hip` (the C driver, not the Python binding) on
BASELINE configs[1..4] over the 24-chromosome 3 088 269 832-base genome, against what the unmodified reference printed
for the same input (tests/golden/genome_cli.json, recorded by tools/make_genome_golden.py in the build container:
sha256, line and byte counts, first and last lines of stdout, the percentile line of stderr).

The 12 M-read input is not committed: tools/genome_reads.c regenerates it here (seeded) and its digest is checked
first.  Each pipeline runs with whole chromosomes on one device and again as three stretches-of-bases shards sharing
the GPU (`--gpus=3 --sharding=bases`, GDSP_OVERSUBSCRIBE_GPUS=1).  Reference: genodsp.c:822-990 end to end
(read_intervals :1187-1350, the batching loop :900-936, report_intervals :1561-1691).

GDSP_GENOME_REPORT=<file>: append each run's `--report=gpu` table and wall time there (profiles/r03_cli_genome.txt).
"""
import json
import os
import subprocess

import pytest

import genome_cli as gc
from conftest import ROOT

pytestmark = pytest.mark.gpu

BIN = os.path.join(ROOT, "genodsp_amd", "genodsp_hip")


@pytest.fixture(scope="module")
def genome():
    if not os.path.exists(gc.GOLDEN):
        pytest.skip("tests/golden/genome_cli.json has not been recorded")
    gold = json.load(open(gc.GOLDEN))
    if not os.path.exists(BIN):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "genodsp_amd", "host")])
    chroms, reads, sha, lines = gc.make_input(gold["seed"])
    assert (sha, lines) == (gold["input_sha256"], gold["input_lines"]), "the regenerated read file differs from the recorded one"
    return gold, chroms, reads


# the 8-shard rows are the contract's device count (BASELINE configs[3], [4]: "across 8xMI355X") rehearsed on the one GPU:
# the LPT deal of 24 chromosomes over 8 devices, the genome cut into 8 stretches of bases, percentile's reductions over 8
SHARDINGS = {"chromosomes": [], "bases_3_shards": ["--gpus=3", "--sharding=bases"],
             "chromosomes_8_shards": ["--gpus=8", "--sharding=chromosomes"], "bases_8_shards": ["--gpus=8", "--sharding=bases"]}


EIGHT = {("config2_peaks", "bases_8_shards"), ("config3_morphology", "chromosomes_8_shards"),
         ("config4_percentile_without_preserve", "bases_8_shards")}

# configs[4] once more without `--preserve`: the reference needs it (its percentile sorts the signal in place,
# percentile.c:34-36, and the command line restores it from the text it wrote); here the signal is never touched, the same
# bytes come out, and `percentile = binarize` runs fused in one read of the signal instead of around 1.6 s of text
EXTRA = {"config4_percentile_without_preserve": "config4_percentile"}


@pytest.mark.parametrize("sharding", list(SHARDINGS))
@pytest.mark.parametrize("name", list(gc.PIPELINES) + list(EXTRA))
def test_genome_scale_pipeline_prints_what_the_reference_prints(name, sharding, genome):
    gold, chroms, reads = genome
    recorded = EXTRA.get(name, name)
    if sharding.endswith("_8_shards") and (name, sharding) not in EIGHT:
        pytest.skip("the 8-shard rehearsal: configs[2] and [4] over eight stretches of bases, configs[3] over eight LPT shards")
    if recorded not in gold["runs"]:
        pytest.skip("no recorded reference run for " + recorded)
    want = gold["runs"][recorded]
    assert want["returncode"] == 0
    preserve = os.path.join(gc.workdir(), "preserve.hip.dat")
    env = dict(os.environ, GDSP_OVERSUBSCRIBE_GPUS="1")
    args = gc.args_for(recorded, chroms, preserve)
    if name in EXTRA:
        args = [a for a in args if not a.startswith("--preserve=")]
    cmd = [BIN, args[0], "--report=gpu"] + SHARDINGS[sharding] + args[1:]
    got = gc.digest_run(cmd, reads, env=env)
    if os.path.exists(preserve):
        os.remove(preserve)
    report = os.environ.get("GDSP_GENOME_REPORT")
    if report:
        with open(report, "a") as f:
            f.write("== %s, sharding %s: genodsp_hip %.2f s wall (ingest of %d lines, operators, %d output lines), reference %.2f s "
                    "wall in the build container; stdout %s\n   %s\n%s\n" %
                    (name, sharding, got["wall_s"], gold["input_lines"], got["lines"], want["wall_s"],
                     "identical (sha256 %s)" % got["sha256"][:16] if got["sha256"] == want["sha256"] else "DIFFERS",
                     " ".join(cmd[2:]), got["stderr"]))
    assert got["returncode"] == 0, got["stderr"][-2000:]
    assert got["head"] == want["head"]
    assert got["tail"] == want["tail"]
    assert (got["lines"], got["bytes"]) == (want["lines"], want["bytes"])
    assert got["sha256"] == want["sha256"]
    for line in want["stderr"].splitlines():                      # "percentile 99.000 is ..."
        assert line in got["stderr"].splitlines()
    assert "--report=gpu" in got["stderr"]
