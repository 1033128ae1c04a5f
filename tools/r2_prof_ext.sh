#!/bin/bash
# rocprofv3 kernel stats of the extended mode (7 views 1080p, 4 iterations)
set -o pipefail
export TMPDIR=/tmp
OUT=gpurun_out/prof/ext
mkdir -p $OUT
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 tools/extended_eval.py --height 1080 --width 1920 --iters 4 > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
f=$(find $OUT/trace -name "*_kernel_stats.csv" | head -1)
{ echo "# r02 extended mode: rocprofv3 --kernel-trace --stats -- python3 tools/extended_eval.py --height 1080 --width 1920 --iters 4"
  echo "# (7 views 1920x1080, 7x7 window stride 2, 4 sources; warm-up run of 1 iteration + extended run of 4 + parity run of 4; amvs kernels only)"
  head -1 $f; grep "amvs::" $f | head -30; } > gpurun_out/r02_extended.txt
cat gpurun_out/r02_extended.txt | cut -c1-200
find $OUT -name "*.db" -delete; find $OUT -name "*_kernel_trace.csv" -delete
