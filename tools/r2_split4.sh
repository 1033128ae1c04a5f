#!/bin/bash
set -o pipefail
O=gpurun_out/r2_split4
mkdir -p $O
for v in "2 4 64 27648" "2 2 64 27648" "2 1 64 27648" "2 3 64 27648" "2 4 64 16384" "2 4 64 20480" "2 4 64 32768" "2 4 64 40960" "2 2 64 20480" "2 2 64 32768"; do
  set -- $v
  timeout -k 10 200 python bench.py --steps 6 --warmup 2 --schedule split --split-groups $1 --split-rows $2 --tile-rows $3 --split-lds $4 --no-planesweep --no-cpu-baseline > $O/b_$1_$2_$3_$4.json 2> $O/b_$1_$2_$3_$4.err || { echo "bench $v failed"; tail -5 $O/b_$1_$2_$3_$4.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("$O/b_$1_$2_$3_$4.json").read().strip().splitlines()[-1])
print("groups/rows/tile_rows/lds $v", r["value"], r["ms_per_step"], r["roofline"]["frac"])
PY
done
