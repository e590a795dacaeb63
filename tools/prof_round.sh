#!/bin/bash
# every kernel the round's claims rest on, through tools/prof_any.sh; summaries land in <out>/<op>.txt
# usage: tools/prof_round.sh [outdir] ; OPS="a b c" / SQOPS="x y" select a part (a whole round takes two calls of 20 minutes)
out=${1:-gpurun_out/prof_r05}
OPS=${OPS-smooth_hann smooth_hann_batch localmax dilate erode close binarize binarize_batch morph_fused morph_fused_batch percentile percentile_binarize smooth_exact_batch smooth_fma_batch cumsum sum1000 sum2000 clump report peaks_exact peaks_exact_batch}
SQOPS=${SQOPS-smooth_exact smooth_fma peaks_fma}
for op in $OPS; do tools/prof_any.sh $out $op 3 > /dev/null; echo "== $op"; done
for op in $SQOPS; do SQ=1 tools/prof_any.sh $out $op 3 > /dev/null; echo "== $op"; done
