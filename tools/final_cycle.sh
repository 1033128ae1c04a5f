#!/bin/bash
# One GPU cycle for a finished build (run under gpurun from the repository root): GPU tests, the
# rocprofv3 profiles of both sweeps in both arithmetic modes and of the CLI-default operating point
# (summaries -> profiles/ with tools/profile_summary.py afterwards, see tools/summarise_profiles.sh), the
# default bench, the rehearsal of the N > 1 path, the randomised parity sweeps.  Every step stops the cycle on failure.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$PWD}" || exit 1
mkdir -p gpurun_out
if [ -z "$SKIP_TESTS" ]; then
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -5 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
fi
PROF_TAG=pm_fast  BENCH_ARGS="--no-planesweep"                     tools/profile.sh > gpurun_out/profile_pm_fast.log 2>&1 || exit 1
PROF_TAG=pm_exact BENCH_ARGS="--no-planesweep --mode exact"        tools/profile.sh > gpurun_out/profile_pm_exact.log 2>&1 || exit 1
PROF_TAG=pm_cli   BENCH_ARGS="--no-planesweep --mode exact --patch 11 --iters 3 --height 756 --width 1008" tools/profile.sh > gpurun_out/profile_pm_cli.log 2>&1 || exit 1
PROF_TAG=ps_fast  BENCH_ARGS="--workload planesweep --steps 10 --warmup 8"               tools/profile.sh > gpurun_out/profile_ps_fast.log 2>&1 || exit 1
PROF_TAG=ps_exact BENCH_ARGS="--workload planesweep --mode exact --steps 10 --warmup 8"  tools/profile.sh > gpurun_out/profile_ps_exact.log 2>&1 || exit 1
timeout -k 10 400 python bench.py > gpurun_out/bench_full.json 2> gpurun_out/bench_full.err || exit 1
cut -c1-200 gpurun_out/bench_full.json
if [ -z "$SKIP_FUZZ" ]; then
AMVS_BENCH_BACKEND=gloo AMVS_BENCH_ONE_DEVICE=1 timeout -k 10 600 python bench.py --gpus 2 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/rehearse.log 2>&1 || { tail -5 gpurun_out/rehearse.log; exit 1; }
timeout -k 10 600 python tools/fuzz_parity.py --cases 80 > gpurun_out/fuzz_pm.log 2>&1 || { tail -3 gpurun_out/fuzz_pm.log; exit 1; }
timeout -k 10 600 python tools/fuzz_parity.py --sweep --cases 80 > gpurun_out/fuzz_ps.log 2>&1 || { tail -3 gpurun_out/fuzz_ps.log; exit 1; }
timeout -k 10 600 python tools/fuzz_knn.py --cases 40 > gpurun_out/fuzz_knn.log 2>&1 || { tail -3 gpurun_out/fuzz_knn.log; exit 1; }
tail -1 gpurun_out/fuzz_pm.log; tail -1 gpurun_out/fuzz_ps.log; tail -1 gpurun_out/fuzz_knn.log
fi
echo cycle-ok
