#!/bin/bash
# gpurun_out/ (scratch) -> profiles/ (tracked): the round's bench lines, kernel stats and per-operator summaries
# usage: tools/collect_profiles.sh [round tag, default r03] [library build id for traffic.json]
R=${1:-r05}
# the library the summaries must have been measured on: the last commit that touched the kernels or their headers
ID=${2:-$(git log -1 --format=%h --abbrev=12 -- genodsp_amd/csrc include)}
G=gpurun_out; P=profiles
for f in ${R}_bench_smooth_hann.json ${R}_bench_smooth_hann_under_rocprof.json ${R}_bench_workloads.jsonl ${R}_ops_throughput.txt ${R}_cli_genome.txt; do
  [ -f $G/$f ] && cp $G/$f $P/$f
done
ks=$(ls $G/prof_bench/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$ks" ] && cp $ks $P/${R}_bench_smooth_hann_kernel_stats.csv
for t in $G/prof_${R}/*.txt; do
  op=$(basename $t .txt)
  cp $t $P/${R}_prof_$op.txt
  ks=$(ls $G/prof_${R}/${op}_stats/*/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$ks" ] && cp $ks $P/${R}_prof_${op}_kernel_stats.csv
done
python3 tools/make_traffic.py $ID $P $R
