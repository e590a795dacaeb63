#!/bin/bash
# every kernel the round's claims rest on, through tools/prof_any.sh; summaries land in <out>/<op>.txt
out=${1:-gpurun_out/prof_r02}
for op in smooth_hann localmax dilate erode close binarize morph_fused percentile cumsum sum1000 sum2000 clump report; do tools/prof_any.sh $out $op 3 > /dev/null; echo "== $op"; done
for op in smooth_exact smooth_fma peaks_exact peaks_fma; do SQ=1 tools/prof_any.sh $out $op 3 > /dev/null; echo "== $op"; done
