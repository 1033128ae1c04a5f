#!/bin/bash
set -o pipefail
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests/test_hip_fast_parity.py tests/test_hip_fullsize_parity.py -m gpu -q -p no:cacheprovider -k "plane_sweep or config2" > gpurun_out/r2_tests13.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/r2_tests13.log
for m in fast exact; do timeout -k 10 200 python bench.py --workload planesweep --steps 5 --warmup 1 --no-cpu-baseline --mode $m > gpurun_out/r2_ps_$m.log 2>&1; tail -1 gpurun_out/r2_ps_$m.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$m', round(d['value']), d['roofline']['avg_launch_ms'])"; done
echo cycle-done
