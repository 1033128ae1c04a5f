#!/bin/bash
# GPU cycle 5: timing-only ablations at 16 waves/CU (scratch builds), occupancy/strip sweep
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
ab() { v=$1; shift; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v$TAG FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
for v in ab_occ16 ab_nohsum ab_l1gather ab_l1_nohsum ab_nowin; do TAG=_th24 ab $v; done
for th in 12 20 32 48; do TAG=_th$th ab ab_occ16 --tile-rows $th; done
for th in 24 48 96; do TAG=_th$th ab ab_l1gather --tile-rows $th; done
unset AMVS_LIB
timeout -k 10 300 python -m pytest tests/test_hip_fast_parity.py -m gpu -q -p no:cacheprovider -k mode_selection > gpurun_out/r2_tests5.log 2>&1; echo "rc=$?"; tail -2 gpurun_out/r2_tests5.log
echo cycle-done
