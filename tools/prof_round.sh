#!/bin/bash
# every kernel the round's claims rest on, through tools/prof_any.sh; summaries land in <out>/<op>.txt
out=${1:-gpurun_out/prof_r03}
for op in smooth_hann smooth_hann_batch localmax dilate erode close binarize binarize_batch morph_fused morph_fused_batch percentile cumsum sum1000 sum2000 clump report peaks_exact peaks_exact_batch; do tools/prof_any.sh $out $op 3 > /dev/null; echo "== $op"; done
for op in smooth_exact smooth_fma peaks_fma; do SQ=1 tools/prof_any.sh $out $op 3 > /dev/null; echo "== $op"; done
