#!/bin/bash
# GPU cycle 1 of round 2: parity tests (both arithmetic modes, full sizes), benches, fast-kernel A/B.
set -o pipefail
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 1500 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > gpurun_out/r2_tests1.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r2_tests1.log
echo "== bench fast"; timeout -k 10 400 python bench.py --steps 5 --warmup 1 > gpurun_out/r2_bench_fast.log 2>&1; echo rc=$?; tail -1 gpurun_out/r2_bench_fast.log
echo "== bench exact"; timeout -k 10 300 python bench.py --steps 5 --warmup 1 --mode exact --no-cpu-baseline --no-planesweep > gpurun_out/r2_bench_exact.log 2>&1; echo rc=$?; tail -1 gpurun_out/r2_bench_exact.log
ab() { v=$1; shift; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
echo "== A/B"
for v in base s1 s4 l2 l4 wm1 wp1; do ab $v; done
unset AMVS_LIB
for th in 16 20 28 32 40; do TAG=_th$th ab base --tile-rows $th; done
echo cycle-done
