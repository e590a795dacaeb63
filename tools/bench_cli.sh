#!/bin/bash
# End-to-end wall time of the pipeline CLI on a synthetic read-coverage file: the unmodified
# reference binary (oracle/_ref/genodsp, when present) against genodsp_hip, same input, outputs compared.
# usage: tools/bench_cli.sh [intervals (default 5000000)]
N=${1:-5000000}
T=${TMPDIR:-/tmp}/gdsp_cli_bench
mkdir -p $T
python3 - "$N" "$T" <<'PY'
import sys, numpy as np
n, out = int(sys.argv[1]), sys.argv[2]
lens = [60000000, 50000000, 40000000, 30000000, 20000000, 10000000]
with open(out + "/genome.chroms", "w") as f:
    for i, L in enumerate(lens):
        f.write("chr%d %d\n" % (i + 1, L))
rng = np.random.default_rng(7)
with open(out + "/reads.dat", "w") as f:
    for i, L in enumerate(lens):
        k = n * L // sum(lens)
        s = np.sort(rng.integers(0, L - 160, k))
        e = s + rng.integers(50, 151, k)
        f.write("".join("chr%d\t%d\t%d\n" % (i + 1, a, b) for a, b in zip(s.tolist(), e.tolist())))
PY
PIPE="--novalue --precision=3 = smooth W=101 = localmax N=11"
timeit() { local label=$1; shift; local t0=$(date +%s.%N); "$@"; local t1=$(date +%s.%N); python3 -c "print('$label: %.2f s wall' % ($t1 - $t0))"; }
run_ref()  { oracle/_ref/genodsp --chromosomes=$T/genome.chroms $PIPE < $T/reads.dat > $T/ref.out; }
run_hip()  { genodsp_amd/genodsp_hip --chromosomes=$T/genome.chroms $PIPE < $T/reads.dat > $T/hip.out; }
run_noop() { genodsp_amd/genodsp_hip --chromosomes=$T/genome.chroms --novalue --nooutput < $T/reads.dat > /dev/null; }
if [ -x oracle/_ref/genodsp ]; then timeit "reference genodsp              " run_ref; fi
timeit "genodsp_hip (exact)            " run_hip
timeit "genodsp_hip ingest only        " run_noop
if [ -f $T/ref.out ]; then cmp $T/ref.out $T/hip.out && echo "outputs identical ($(wc -l < $T/hip.out) lines)"; fi
