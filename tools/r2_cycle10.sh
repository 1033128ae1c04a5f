#!/bin/bash
# GPU cycle 10: full test suite + benches of the settled kernels (16 waves/CU cap, global loads)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 1500 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > gpurun_out/r2_tests10.log 2>&1; echo "tests rc=$?"; tail -6 gpurun_out/r2_tests10.log
ab() { v=$1; shift; timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v$TAG FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
TAG=_fast ab base
TAG=_exact ab base --mode exact
for th in 16 20 28 32; do TAG=_fast_th$th ab base --tile-rows $th; TAG=_exact_th$th ab base --mode exact --tile-rows $th; done
echo "== smoke"; timeout -k 10 300 python __graft_entry__.py smoke 2>&1 | tail -3
echo cycle-done
