#!/bin/bash
# paired-band schedule: automatic strip height / views per launch against explicit ones across shapes
run() { tag=$1; shift
  python bench.py --no-planesweep --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; r=json.loads(sys.stdin.read()); print('$tag', r['value'], r['ms_per_step'], 'rows', r['config']['tile_rows'], 'vpl', r['config']['views_per_launch'])"
}
run "k7 16v auto" --steps 3
run "k7 16v vpl4 rows18" --steps 3 --views-per-launch 4 --tile-rows 18
run "k7 8v auto" --steps 4 --views-per-gpu 8
run "k7 8v vpl4" --steps 4 --views-per-gpu 8 --views-per-launch 4
run "k7 4v auto" --steps 6 --views-per-gpu 4
run "k7 4v rows12" --steps 6 --views-per-gpu 4 --tile-rows 12
run "k7 32v auto" --steps 2 --views-per-gpu 32
run "k7 32v vpl16" --steps 2 --views-per-gpu 32 --views-per-launch 16
run "k5 auto" --patch 5 --steps 3
run "k5 rows16" --patch 5 --steps 3 --tile-rows 16
run "k5 rows10" --patch 5 --steps 3 --tile-rows 10
run "k5 vpl4 rows12" --patch 5 --steps 3 --views-per-launch 4 --tile-rows 12
run "k5 vpl4 rows16" --patch 5 --steps 3 --views-per-launch 4 --tile-rows 16
run "1440p auto" --height 1440 --width 2560 --steps 2
run "1440p rows16" --height 1440 --width 2560 --steps 2 --tile-rows 16
run "1440p rows18" --height 1440 --width 2560 --steps 2 --tile-rows 18
run "4k8 auto" --views-per-gpu 8 --height 2160 --width 3840 --steps 2
run "4k8 rows16" --views-per-gpu 8 --height 2160 --width 3840 --steps 2 --tile-rows 16
run "4k8 rows18" --views-per-gpu 8 --height 2160 --width 3840 --steps 2 --tile-rows 18
