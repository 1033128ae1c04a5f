#!/bin/bash
# split schedule: window-kernel strip height x sampling-strip height x groups
set -o pipefail
O=gpurun_out/r2_split3
mkdir -p $O
for v in "2 8 32" "2 8 48" "2 8 64" "2 8 96" "2 4 64" "2 12 64" "3 8 64" "4 8 64" "1 8 64"; do
  set -- $v
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --schedule split --split-groups $1 --split-rows $2 --tile-rows $3 --no-planesweep --no-cpu-baseline > $O/b_$1_$2_$3.json 2> $O/b_$1_$2_$3.err || { echo "bench $v failed"; tail -5 $O/b_$1_$2_$3.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("$O/b_$1_$2_$3.json").read().strip().splitlines()[-1])
print("groups/rows/tile_rows $v", r["value"], r["ms_per_step"], r["roofline"]["frac"])
PY
done
