#!/bin/bash
# GPU cycle 7: pipelined fast step, occupancy x strip height
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
ab() { v=$1; shift; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v$TAG FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
for v in occ12 occ8 occ8s4; do for th in 24 32 48 64; do TAG=_th$th ab $v --tile-rows $th; done; done
unset AMVS_LIB
timeout -k 10 600 python -m pytest tests/test_hip_fullsize_parity.py -m gpu -q -p no:cacheprovider -k "config2" > gpurun_out/r2_tests7.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_tests7.log
echo cycle-done
