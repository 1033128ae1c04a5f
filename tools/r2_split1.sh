#!/bin/bash
# split-schedule bring-up: parity tests, then bench A/B (fused vs split, group / row variants)
set -o pipefail
O=gpurun_out/r2_split1
mkdir -p $O
timeout -k 10 500 python -m pytest tests/test_hip_split_schedule.py -m gpu -q -x -p no:cacheprovider > $O/tests.log 2>&1
echo "tests rc=$?" | tee -a $O/tests.log
tail -5 $O/tests.log
for v in "view-major 0 0" "split 2 16" "split 2 24" "split 4 16" "split 2 8" "split 1 16"; do
  set -- $v
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --schedule $1 --split-groups $2 --split-rows $3 --no-planesweep --no-cpu-baseline > $O/bench_$1_$2_$3.json 2> $O/bench_$1_$2_$3.err || { echo "bench $v failed"; tail -5 $O/bench_$1_$2_$3.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("$O/bench_$1_$2_$3.json").read().strip().splitlines()[-1])
print("$v", r["value"], r["ms_per_step"], r["roofline"]["frac"])
PY
done
