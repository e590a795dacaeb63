#!/bin/bash
# Extract put_u64/put_fixed from the driver and hold them to snprintf (prints bad=0 when every string matches).
set -e
T=${TMPDIR:-/tmp}/gdsp_fmt_check
mkdir -p $T
python3 - "$T" <<'PY'
import sys
src = open("genodsp_amd/host/genodsp_hip.c").read()
a = src.index("static char* put_u64 (char* p, unsigned long long u, int minDigits)")
b = src.index("/* Millions of runs (a smoothed genome")
open(sys.argv[1] + "/put_fixed.inc", "w").write(src[a:b])
PY
gcc -O2 -I$T -o $T/check tools/check_output_format.c -lm
$T/check "$@"
