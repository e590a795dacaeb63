#!/bin/bash
# long Hann windows: the 256-, 384- and 768-thread forms of hann_blocks_rt_kernel side by side on one box
O=gpurun_out/r04_hann_forms.txt
: > $O
for t in 256 384 768; do
  GDSP_HANN_THREADS=$t BURST=10 TAG="threads=$t" python tools/bench_one.py smooth_hann801,smooth_hann1001,smooth_hann1201,smooth_hann1501,smooth_hann1701,smooth_hann2001,smooth_hann2401,smooth_hann3201,smooth_hann4001 2>&1 | grep smooth_hann >> $O
done
sort -k1,1 -s $O
