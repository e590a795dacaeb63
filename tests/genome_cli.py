"""Shared by tools/make_genome_golden.py (build container: records what the reference prints) and
tests/test_cli_genome.py (GPU box: genodsp_hip must print the same): the seeded 24-chromosome, 3.1 Gbp read file of
BASELINE configs[1..4] (tools/genome_reads.c), the four pipelines, and a streaming digest of a program's stdout."""
import hashlib
import os
import subprocess
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SEED = 20240611
GOLDEN = os.path.join(ROOT, "tests", "golden", "genome_cli.json")

# BASELINE.json configs[1..4] as command lines.  @preserve@ is a scratch file.
PIPELINES = {
    "config1_smooth":     ["--novalue", "--precision=3", "=", "smooth", "W=101"],
    "config2_peaks":      ["--novalue", "--precision=3", "=", "smooth", "W=101", "=", "localmax", "N=11"],
    "config3_morphology": ["--novalue", "=", "dilate", "1001", "=", "erode", "1001", "=", "binarize"],
    "config4_percentile": ["--novalue", "=", "percentile", "99", "--min=1/inf", "--preserve=@preserve@",
                           "=", "binarize", "--threshold=percentile99"],
}


def workdir():
    d = os.path.join(os.environ.get("TMPDIR", "/tmp"), "gdsp_genome_cli")
    os.makedirs(d, exist_ok=True)
    return d


def make_input(seed=SEED):
    """(chromosomes file, intervals file, sha256 of the intervals, lines): generated once per work directory."""
    d = workdir()
    exe = os.path.join(d, "genome_reads")
    subprocess.check_call(["gcc", "-O2", "-o", exe, os.path.join(ROOT, "tools", "genome_reads.c")])
    chroms, reads = os.path.join(d, "genome.chroms"), os.path.join(d, "reads.%d.dat" % seed)
    if not os.path.exists(reads + ".ok"):
        with open(reads, "wb") as f:
            subprocess.check_call([exe, chroms, str(seed)], stdout=f)
        open(reads + ".ok", "w").close()
    else:
        subprocess.check_call([exe, chroms, str(seed)], stdout=subprocess.DEVNULL)      # (re)writes the chromosomes file
    h, lines = hashlib.sha256(), 0
    with open(reads, "rb") as f:
        while True:
            b = f.read(1 << 24)
            if not b:
                break
            h.update(b)
            lines += b.count(b"\n")
    return chroms, reads, h.hexdigest(), lines


def digest_run(cmd, stdin_path, env=None, keep=5):
    """Run cmd with stdin from a file; digest its stdout as it comes (it may be several GB).
    -> dict(returncode, sha256, lines, bytes, head, tail, stderr, wall_s)"""
    t0 = time.time()
    with open(stdin_path, "rb") as fin:
        p = subprocess.Popen(cmd, stdin=fin, stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        import threading
        err = []
        t = threading.Thread(target=lambda: err.append(p.stderr.read()))
        t.start()
        h, lines, size = hashlib.sha256(), 0, 0
        head, tail = b"", b""
        while True:
            b = p.stdout.read(1 << 24)
            if not b:
                break
            h.update(b)
            lines += b.count(b"\n")
            size += len(b)
            if len(head) < 4096:
                head += b[:4096 - len(head)]
            tail = (tail + b)[-4096:]
        rc = p.wait()
        t.join()
    head_lines = head.decode().split("\n")[:keep]
    tail_lines = tail.decode().split("\n")
    tail_lines = [x for x in tail_lines if x][-keep:]
    return {"returncode": rc, "sha256": h.hexdigest(), "lines": lines, "bytes": size, "head": head_lines,
            "tail": tail_lines, "stderr": err[0].decode(errors="replace"), "wall_s": round(time.time() - t0, 2)}


def args_for(name, chroms, preserve):
    return ["--chromosomes=" + chroms] + [a.replace("@preserve@", preserve) for a in PIPELINES[name]]
