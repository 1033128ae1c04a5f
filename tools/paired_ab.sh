#!/bin/bash
# paired-band schedule against the classic strips across shapes (fast arithmetic), one run per pair
run() { tag=$1; shift
  python bench.py --no-planesweep --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tag', r['value'], r['ms_per_step'], 'rows', r['config']['tile_rows'], 'vpl', r['config']['views_per_launch'])"
}
for sch in view-major paired; do
run "k7 $sch" --steps 4 --schedule $sch
run "k5 $sch" --patch 5 --steps 3 --schedule $sch
run "k3 $sch" --patch 3 --steps 3 --schedule $sch
run "1440p $sch" --height 1440 --width 2560 --steps 2 --schedule $sch
run "4k8 $sch" --views-per-gpu 8 --height 2160 --width 3840 --steps 2 --schedule $sch
run "v32 $sch" --views-per-gpu 32 --steps 2 --schedule $sch
run "k7 vpl16 $sch" --steps 3 --views-per-launch 16 --schedule $sch
done
