#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
ab() { v=$1; shift; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v$TAG FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
for v in base pl_nodout pl_nocost pl_noflush pl_noqueue pl_ntstore; do TAG=_th24 ab $v; done
echo cycle-done
