#!/usr/bin/env python3
"""profiles/traffic.json from the rocprofv3 summaries tools/prof_any.sh writes (profiles/r02_prof_<op>.txt): per kernel the
HBM bytes of one launch (FETCH_SIZE, WRITE_SIZE, corrected as the summaries say) over its algorithmic bytes (B/base of
the operator x bases of the launch).  bench.py multiplies a live launch's algorithmic bytes by that ratio for
roofline.traffic and prints where the ratio came from.
usage: python3 tools/make_traffic.py <library build id> [profiles dir] [round tag, default r03]"""
import glob
import json
import os
import re
import subprocess
import sys

SKIP = ("synth_coverage_kernel", "__amd_rocclr", "fill", "copyBuffer")
# bytes the operator has to move per base: in + out, or in only for the passes that only read
ALGORITHMIC = {"pc_fixup_kernel": None, "pc_hist_chain_kernel": None, "pc_pick_kernel": None, "peaks_probe_kernel": None, "peaks_exact_kernel": None, "peaks_init_kernel": None, "fir_fixed_extrema_gated_kernel": None,
               "pc_sample_kernel": 8, "pc_partition_tab_kernel": 8, "cumsum_totals_kernel": 8, "report_count_kernel": 8,
               "report_write_kernel": 8, "clump_write_kernel": 8,        # (clump_chunk_stats_kernel reads the signal and writes R': the default 16)
               # launches over a few words per chunk, or whose traffic is not a per-base figure: bytes only
               "clump_scan_": None, "clump_bits_": None, "clump_mark_kernel": None, "pc_res_": None, "pc_sample_tab_kernel": 8, "report_scan_kernel": None, "pc_hist_keys_kernel": None,
               "cumsum_offsets_kernel": None, "hf_": None, "window_sum_rows_kernel": None}



def same_kernels(built):
    """the build a summary was measured on (`# library:` line, the hash gdsp_version() carries) has the kernels and headers of HEAD"""
    if not built or built.endswith("-dirty"):
        return False
    h = built.split()[-1]
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.call(["git", "-C", repo, "diff", "--quiet", h, "HEAD", "--", "genodsp_amd/csrc", "include"],
                           stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL) == 0


lib = sys.argv[1]
root = sys.argv[2] if len(sys.argv) > 2 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
kernels = {}
stale = []
tag = sys.argv[3] if len(sys.argv) > 3 else "r03"
for path in sorted(glob.glob(os.path.join(root, tag + "_prof_*.txt"))):
    lines = open(path).read().splitlines()
    m = re.match(r"# (\S+) on (\d+) bases", lines[0])
    if not m:
        continue
    bases = int(m.group(2))
    built = next((l.split(":", 1)[1].strip() for l in lines[:4] if l.startswith("# library:")), None)
    if not same_kernels(built):
        # a summary measured on another build than the tree's kernels: refuse (evidence hygiene)
        stale.append((os.path.basename(path), built))
        continue
    for line in lines:
        f = line[74:].split()
        if line.startswith("#") or line.startswith("kernel") or len(f) < 6:
            continue
        name = line[:74].split("(")[0].replace("void ", "").strip()
        if any(s in name for s in SKIP) or name in kernels:
            continue
        try:
            fetch, write = int(f[3]), int(f[4])
        except ValueError:
            continue
        per_base = 16
        for key, val in ALGORITHMIC.items():
            if name.startswith(key):
                per_base = val
        if name.startswith("pc_partition_tab_kernel") and name.endswith("true, true>"):
            per_base = 16                 # the fused form writes the binarized signal too
        launch_bases = bases
        if m.group(1).endswith("_batch") or m.group(1) == "percentile_binarize":
            # tools/prof_op.py runs these operators over three vectors of n/2, n/3 and n/6 bases; the `_batch` kernels
            # cover the three in one launch, the per-source kernels of percentile_binarize average a third of n per launch
            if "tab_kernel" not in name and "batch" not in name and name.startswith("pc_"):
                launch_bases = bases // 3
        entry = {"source": "profiles/%s (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; tools/prof_any.sh)" % os.path.basename(path),
                 "library": (built.split()[-1] if built else lib), "bases_per_launch": launch_bases, "fetch_bytes": fetch, "write_bytes": write}
        if per_base is not None:
            entry["algorithmic_bytes"] = per_base * launch_bases
            entry["hbm_bytes_over_algorithmic"] = round((fetch + write) / (per_base * launch_bases), 4)
        kernels[name] = entry
if stale and not os.environ.get("GDSP_TRAFFIC_ALLOW_STALE"):
    sys.exit("traffic.json NOT written: these summaries were not measured on library %s: %s" % (lib, stale))
note = ("HBM bytes per launch over algorithmic bytes, from rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (gfx950 "
        "corrections of MI355X_MICROARCH.md applied: FETCH_SIZE doubled, KiB -> bytes); `library` = git hash of the build that was "
        "profiled, so a stale ratio shows in bench.py's roofline.traffic_measured; written by tools/make_traffic.py")
with open(os.path.join(root, "traffic.json"), "w") as f:
    json.dump({"note": note, "kernels": kernels}, f, indent=1)
print("%d kernels" % len(kernels))
