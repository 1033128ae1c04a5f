#!/bin/bash
# GPU cycle 4: band-major schedule
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
ab() { v=$1; shift; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v$TAG FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
TAG=_bm ab base
TAG=_vm ab base --schedule view-major
for th in 270 135 90 68 45 34 24; do TAG=_bm_th$th ab base --tile-rows $th; done
for v in occ20 occ16; do for th in 135 68; do TAG=_bm_th$th ab $v --tile-rows $th; done; done
TAG=_bm_exact ab base --mode exact
unset AMVS_LIB
pmc() { name=$1; shift; rocprofv3 --pmc $PMC --output-format csv -d gpurun_out/r2_pmc/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 gpurun_out/r2_pmc_$name.log; }; }
PMC="FETCH_SIZE" pmc fetch_bm
PMC="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" pmc l2_bm
for f in $(find gpurun_out/r2_pmc -name "*_counter_collection.csv"); do head -1 "$f" > "$f.tmp"; grep "amvs::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; done
find gpurun_out/r2_pmc -name "*.db" -delete
echo "== parity of the new schedule"; timeout -k 10 900 python -m pytest tests/test_hip_fast_parity.py tests/test_hip_fullsize_parity.py -m gpu -q -p no:cacheprovider -x > gpurun_out/r2_tests4.log 2>&1; echo "rc=$?"; tail -4 gpurun_out/r2_tests4.log
echo cycle-done
