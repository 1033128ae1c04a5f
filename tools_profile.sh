#!/bin/bash
# Profiling recipe (run on the GPU box through gpurun): kernel trace + stats of a short bench,
# then PMC passes in separate runs (never combined with tracing domains).
set -o pipefail
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof
mkdir -p "$OUT"
ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:---iters 2 --samples 8}"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 $ARGS > "$OUT/trace.log" 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d "$OUT/pmc_sq" -- python3 $ARGS > "$OUT/pmc_sq.log" 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- python3 $ARGS > "$OUT/pmc_fetch.log" 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_write" -- python3 $ARGS > "$OUT/pmc_write.log" 2>&1 || exit 1
echo profile-done
