#!/bin/bash
# GPU cycle 3: SQ counters of the fast sweep kernel (where do the wave-cycles go?)
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
pmc() { name=$1; shift; rocprofv3 --pmc $PMC --output-format csv -d gpurun_out/r2_pmc/$name -- python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-planesweep $ARGS > gpurun_out/r2_pmc_$name.log 2>&1 || { echo "pmc $name failed"; tail -3 gpurun_out/r2_pmc_$name.log; }; }
for cfg in base occ16; do
  if [ $cfg = base ]; then unset AMVS_LIB; else export AMVS_LIB=$PWD/build/variants/libamvs_$cfg.so; fi
  ARGS="--tile-rows 24"
  PMC="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" pmc sq1_$cfg
  PMC="SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS" pmc sq2_$cfg
  PMC="SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_WR SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH" pmc sq3_$cfg
  PMC="SQ_INSTS_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_INST_CYCLES_VMEM SQ_WAVE32_INSTS GRBM_GUI_ACTIVE SQ_CYCLES SQ_THREAD_CYCLES_VALU" pmc sq4_$cfg
done
for f in $(find gpurun_out/r2_pmc -name "*_counter_collection.csv"); do head -1 "$f" > "$f.tmp"; grep "amvs::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"; done
find gpurun_out/r2_pmc -name "*.db" -delete
echo cycle-done
