#!/usr/bin/env python3
"""Record what the UNMODIFIED reference (oracle/_ref/genodsp, built by `make -C oracle ref`) prints for BASELINE
configs[1..4] on the seeded 24-chromosome 3.1 Gbp read file of tools/genome_reads.c -> tests/golden/genome_cli.json
(sha256, line and byte counts, first and last lines, the percentile line of stderr, wall time).  The input is not
committed: tests/test_cli_genome.py regenerates it on the GPU box and checks its digest first.

Build container only (needs /root/reference for the reference build; ~25 GB of RAM and ~3 min per pipeline).
usage: tools/make_genome_golden.py [config ...]      (default: all four; existing entries are kept)"""
import json
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import genome_cli as gc

REF = os.path.join(gc.ROOT, "oracle", "_ref", "genodsp")


def main():
    names = sys.argv[1:] or list(gc.PIPELINES)
    if not os.path.exists(REF):
        sys.exit("build the reference first: make -C oracle ref")
    chroms, reads, sha, lines = gc.make_input()
    out = {"seed": gc.SEED, "input_sha256": sha, "input_lines": lines, "input_bytes": os.path.getsize(reads), "runs": {}}
    if os.path.exists(gc.GOLDEN):
        old = json.load(open(gc.GOLDEN))
        if old.get("input_sha256") == sha:
            out["runs"] = old["runs"]
    print("input: %d lines, sha256 %s" % (lines, sha), flush=True)
    for name in names:
        preserve = os.path.join(gc.workdir(), "preserve.ref.dat")
        r = gc.digest_run([REF] + gc.args_for(name, chroms, preserve), reads)
        if os.path.exists(preserve):
            os.remove(preserve)
        r["args"] = gc.PIPELINES[name]
        r["stderr"] = "\n".join(x for x in r["stderr"].splitlines() if x.startswith("percentile "))
        out["runs"][name] = r
        print(name, r["returncode"], r["lines"], "lines", r["bytes"], "bytes", r["wall_s"], "s", r["sha256"], flush=True)
        with open(gc.GOLDEN, "w") as f:
            json.dump(out, f, indent=1)
            f.write("\n")


if __name__ == "__main__":
    main()
