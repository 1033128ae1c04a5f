#!/bin/bash
set -o pipefail
O=gpurun_out/r2_vpl2
mkdir -p $O
IFS=";" read -ra VARS <<< "${VPL_VARIANTS}"; unset IFS
for v in "${VARS[@]}"; do
  set -- $v
  timeout -k 10 200 python bench.py --steps 3 --warmup 1 --views-per-launch $1 --tile-rows $2 --no-planesweep --no-cpu-baseline > $O/b_$1_$2.json 2> $O/b_$1_$2.err || { echo "bench $v failed"; tail -5 $O/b_$1_$2.err; exit 1; }
  python - <<PY
import json
r=json.loads(open("$O/b_$1_$2.json").read().strip().splitlines()[-1])
print("vpl/tile_rows $v", r["value"], r["ms_per_step"], r["config"]["tile_rows"])
PY
done
