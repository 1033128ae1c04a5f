#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats of a short bench,
# then PMC passes in separate runs (never combined with tracing domains).
#   BENCH_ARGS   extra bench.py arguments (default: 2 iterations)
#   PMC_ONLY=1   skip the kernel trace
set -o pipefail
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof
mkdir -p "$OUT"
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:---iters 2 --samples 8}"
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1 || exit 1
fi
# PMC_GROUPS="pmc_fetch pmc_write" restricts the passes
pmc() { name=$1; shift; if [ -n "$PMC_GROUPS" ] && ! echo " $PMC_GROUPS " | grep -q " $name "; then return; fi; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1 || { echo "pmc $name failed"; tail -3 "$OUT/$name.log"; }; }
# (a TA_*/TCP_* group hung rocprofv3 on this pool once: not collected)
pmc pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pmc pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS
pmc pmc_sq3 SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SMEM SQ_IFETCH SQ_INSTS_BRANCH
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# keep the copy-back small: only rows of this repository's kernels survive (the traces of the
# torch kernels that render the synthetic scene are hundreds of MB)
for f in $(find "$OUT" -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv"); do
  head -1 "$f" > "$f.tmp"; grep "amvs::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
done
find "$OUT" -name "*.db" -delete
echo profile-done
