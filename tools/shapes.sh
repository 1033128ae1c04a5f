#!/bin/bash
# other shapes in the fast arithmetic ("Other shapes")
set -o pipefail
O=gpurun_out/${SHAPES_TAG:-r3_shapes}
mkdir -p $O
run() { tag=$1; shift
  timeout -k 10 500 python bench.py --no-planesweep --no-cpu-baseline "$@" > $O/$tag.json 2> $O/$tag.err || { echo "$tag failed"; tail -3 $O/$tag.err; return; }
  python - <<PY
import json
r=json.loads(open("$O/$tag.json").read().strip().splitlines()[-1])
print("$tag", r["value"], r["ms_per_step"], r["roofline"]["frac"], r["config"]["tile_rows"], r.get("dense_points", {}).get("final"))
PY
}
run k3 --patch 3 --steps 3
run k5 --patch 5 --steps 3
run k9 --patch 9 --steps 3
run k11 --patch 11 --steps 3
run v8_4k --views-per-gpu 8 --height 2160 --width 3840 --steps 2
run v16_1440 --views-per-gpu 16 --height 1440 --width 2560 --steps 2
run v32_1080 --views-per-gpu 32 --steps 2
run v4_1080 --views-per-gpu 4 --steps 3
run exact_k7 --mode exact --steps 3
run config5_1gpu --config5 --steps 1 --warmup 1
