#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
echo "== tests"; timeout -k 10 1500 python -m pytest tests -m gpu -q --maxfail=8 -p no:cacheprovider > gpurun_out/r2_tests12.log 2>&1; echo "tests rc=$?"; tail -12 gpurun_out/r2_tests12.log
timeout -k 10 200 python tools/e2e_stereo.py > gpurun_out/r2_e2e_stereo.log 2>&1; tail -4 gpurun_out/r2_e2e_stereo.log
timeout -k 10 200 python tools/e2e_time.py > gpurun_out/r2_e2e_pm.log 2>&1; tail -4 gpurun_out/r2_e2e_pm.log
echo cycle-done
