#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats of one full bench step,
# then PMC passes in separate runs (never combined with tracing domains).
#   BENCH_ARGS   bench.py arguments after the fixed "--steps 1 --warmup 1 --no-cpu-baseline"
#                (default: the full config-3 schedule, fast arithmetic, no plane-sweep sub-record)
#   PROF_TAG     sub-directory of gpurun_out/prof (default "pm")
#   PMC_ONLY=1   skip the kernel trace;  PMC_GROUPS="pmc_fetch pmc_write" restricts the passes
set -o pipefail
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof/${PROF_TAG:-pm}
mkdir -p "$OUT"
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:---no-planesweep}"
if [ -z "$PMC_ONLY" ]; then
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1 || { echo "trace failed"; tail -5 "$OUT/trace.log"; exit 1; }
fi
pmc() { name=$1; shift; if [ -n "$PMC_GROUPS" ] && ! echo " $PMC_GROUPS " | grep -q " $name "; then return; fi; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $ARGS > "$OUT/$name.log" 2>&1 || { echo "pmc $name failed"; tail -3 "$OUT/$name.log"; }; }
# (a TA_*/TCP_* group hung rocprofv3 on this pool once in round 1: not collected)
pmc pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pmc pmc_sq2 SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INSTS_SMEM SQ_INSTS_LDS
pmc pmc_fetch FETCH_SIZE
pmc pmc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
# keep the copy-back small: only rows of this repository's kernels survive (the traces of the
# torch kernels that render the synthetic scene are hundreds of MB)
for f in $(find "$OUT" -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv"); do
  head -1 "$f" > "$f.tmp"; grep "amvs::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
done
find "$OUT" -name "*.db" -delete
echo profile-done $OUT
