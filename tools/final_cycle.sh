#!/bin/bash
# One GPU cycle for a finished build: GPU tests, rocprofv3 profiles of both sweeps, both benches.
# Run under gpurun from the repository root.
export TMPDIR=/tmp
cd "${GRAFT_REPO_ROOT:-$PWD}"
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests.log 2>&1 || { tail -5 gpurun_out/gpu_tests.log; exit 1; }
tail -1 gpurun_out/gpu_tests.log
rm -rf gpurun_out/prof gpurun_out/prof_pm gpurun_out/prof_ps
tools/profile.sh > gpurun_out/profile.log 2>&1 || exit 1
mv gpurun_out/prof gpurun_out/prof_pm
BENCH_ARGS="--workload planesweep" tools/profile.sh > gpurun_out/profile_ps.log 2>&1 || exit 1
mv gpurun_out/prof gpurun_out/prof_ps
timeout -k 10 300 python bench.py > gpurun_out/bench_full.log 2>&1 || exit 1
tail -1 gpurun_out/bench_full.log | cut -c1-160
timeout -k 10 300 python bench.py --workload planesweep > gpurun_out/bench_ps.log 2>&1 || exit 1
tail -1 gpurun_out/bench_ps.log | cut -c1-160
