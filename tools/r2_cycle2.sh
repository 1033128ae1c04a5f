#!/bin/bash
# GPU cycle 2: strip-height / occupancy sweep of the fast kernel + L2 traffic counters (locality model)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== failed tests again"; timeout -k 10 600 python -m pytest tests/test_hip_fast_parity.py -m gpu -q -p no:cacheprovider -k "sampling_stage or mode_selection" > gpurun_out/r2_tests2.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_tests2.log
ab() { v=$1; shift; if [ $v = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$v.so; fi
  timeout -k 10 200 python bench.py --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_ab_$v$TAG.log 2>&1 || { echo "$v$TAG FAILED"; tail -3 gpurun_out/r2_ab_$v$TAG.log; return; }
  tail -1 gpurun_out/r2_ab_$v$TAG.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$v$TAG', round(d['value']), d['roofline']['avg_launch_ms'], d['config']['tile_rows'])"; }
for th in 6 8 10 12 14; do TAG=_th$th ab base --tile-rows $th; done
for th in 12 16; do TAG=_exact_th$th ab base --mode exact --tile-rows $th; done
for v in occ20 occ16 occ12; do for th in 16 24; do TAG=_th$th ab $v --tile-rows $th; done; done
unset AMVS_LIB
pmc() { name=$1; shift; rocprofv3 --pmc $PMC --output-format csv -d gpurun_out/r2_pmc/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-planesweep "$@" > gpurun_out/r2_pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 gpurun_out/r2_pmc_$name.log; }; }
for th in 24 12; do
  PMC="FETCH_SIZE" pmc fetch_th$th --tile-rows $th
  PMC="WRITE_SIZE TCC_HIT_sum TCC_MISS_sum" pmc l2_th$th --tile-rows $th
done
for f in $(find gpurun_out/r2_pmc -name "*_counter_collection.csv"); do head -1 "$f" > "$f.tmp"; grep "amvs::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; done
find gpurun_out/r2_pmc -name "*.db" -delete
du -sh gpurun_out/r2_pmc
echo cycle-done
