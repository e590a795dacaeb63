#!/bin/bash
# A/B of the non-temporal 16-byte accesses (gdsp_ld2 / gdsp_st2, GDSP_STREAMING in gdsp_common.h) against plain ones:
# build both libraries into build/ab/ first (they travel to the GPU box with the snapshot):
#   make -C genodsp_amd/csrc clean && make -C genodsp_amd/csrc && cp genodsp_amd/libgenodsp_hip.so build/ab/lib_nt.so
#   make -C genodsp_amd/csrc clean && make -C genodsp_amd/csrc HIPFLAGS="... -DGDSP_STREAMING=0" && cp ... build/ab/lib_plain.so
# then on the box: bash tools/ab_streaming.sh [ops] | sort -k1,1 -s      (leaves the non-temporal library in place)
ops=${1:-smooth_hann,smooth_fma,smooth_exact,smooth_hann1001,localmax11,dilate,close,binarize,sum100,sum1000,sum2000,cumsum,clump,peaks_fma}
for v in plain nt plain nt; do cp build/ab/lib_$v.so genodsp_amd/libgenodsp_hip.so; BURST=10 TAG=$v python tools/bench_one.py $ops; done
for v in plain nt plain nt; do cp build/ab/lib_$v.so genodsp_amd/libgenodsp_hip.so; TAG=$v python tools/bench_percentile.py 248956422 8; done
