#!/bin/bash
# round 4, second measurement call: long Hann forms; where the sliding FIR's time goes (short strips: inputs per second);
# per-kernel profiles of the percentile calls and the peaks route
O=gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_hip_fir_slide.py -x -q > $O/s2_slide_tests.log 2>&1; echo "slide tests rc=$?" > $O/s2_slide.txt
for strip in 512 1024 2048 4096; do
  GDSP_FIR_SLIDE=2 GDSP_FIR_SLIDE_STRIP=$strip BURST=10 TAG="slide form 2 (LDS-DMA) strip=$strip" python tools/bench_one.py smooth_exact 2>&1 | tail -1 >> $O/s2_slide.txt
done
bash tools/r04_hann_forms.sh > $O/s2_hann_forms.log 2>&1
for strip in 64 128 256 512; do
  GDSP_FIR_SLIDE=1 GDSP_FIR_SLIDE_STRIP=$strip BURST=10 TAG="slide form 1 strip=$strip" python tools/bench_one.py smooth_exact 2>&1 | tail -1 >> $O/s2_slide.txt
done
bash tools/prof_any.sh $O/prof percentile 3 248956422 > /dev/null 2>&1
SQ=1 bash tools/prof_any.sh $O/prof peaks_exact_batch 3 > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/percentile_genome_stats -- python3 tools/prof_op.py percentile_genome 3 > $O/prof/percentile_genome.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof/percentile_binarize_genome_stats -- python3 tools/prof_op.py percentile_binarize_genome 3 > $O/prof/percentile_binarize_genome.log 2>&1
for f in $O/prof/percentile_genome_stats/*/*kernel_stats.csv $O/prof/percentile_binarize_genome_stats/*/*kernel_stats.csv; do echo "== $f"; cut -d, -f1-4,8 $f | head -24; done > $O/s2_pct_genome_stats.txt
cat $O/s2_hann_forms.log $O/s2_slide.txt $O/prof/percentile.txt $O/prof/peaks_exact_batch.txt $O/s2_pct_genome_stats.txt
