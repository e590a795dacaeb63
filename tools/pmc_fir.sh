#!/bin/bash
# PMC passes for the FIR kernel (one counter group per rocprofv3 run, kernel-trace only otherwise).
# usage: tools/pmc_fir.sh <outdir> <variant> <fma|exact|hann>
out=$1; export GDSP_FIR_VARIANT=$2; mode=$3
export TMPDIR=/tmp
run() { rocprofv3 --kernel-trace --output-format csv -d $out/$1 --pmc $2 -- python3 tools/prof_smooth.py $mode 2 > $out/$1.log 2>&1 || tail -3 $out/$1.log; }
mkdir -p $out
run sq1 "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS"
run sq2 "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SMEM SQ_INSTS_VALU SQ_INSTS_SMEM GRBM_GUI_ACTIVE"
run fetch "FETCH_SIZE"
run write "WRITE_SIZE"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
for f in sorted(glob.glob(out + "/*/*/*counter_collection.csv")):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if "fir_" in r["Kernel_Name"] or "hann_" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("%-24s mean %.6g over %d dispatches" % (k, sum(v) / len(v), len(v)))
PY
