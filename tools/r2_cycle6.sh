#!/bin/bash
# GPU cycle 6: software-pipelined fast step
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== parity"; timeout -k 10 900 python -m pytest tests/test_hip_fast_parity.py tests/test_hip_fullsize_parity.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r2_tests6.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r2_tests6.log
ab() { v=$1; shift; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v$TAG FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
TAG=_pipe ab base
for th in 12 16 20 32 48; do TAG=_pipe_th$th ab base --tile-rows $th; done
for v in s1 s4 l4; do TAG=_pipe ab $v; done
TAG=_exact ab base --mode exact
echo cycle-done
