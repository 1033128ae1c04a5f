#!/bin/bash
# kernel-trace of a short bench with an experimental libamvs variant; prints per-kernel averages
export TMPDIR=/tmp
for v in "$@"; do
  export AMVS_LIB=$PWD/build/variants/libamvs_$v.so
  rm -rf gpurun_out/sp_$v; mkdir -p gpurun_out/sp_$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/sp_$v -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --iters 2 --samples 8 > gpurun_out/sp_$v.log 2>&1
  f=$(find gpurun_out/sp_$v -name "*_kernel_stats.csv" | head -1)
  echo "== $v"; python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'amvs::pm_' in r['Name']: print(r['Name'].split('(')[0][:60], r['Calls'], round(float(r['AverageNs'])/1e6,4), 'ms')"
  rm -rf gpurun_out/sp_$v
done
