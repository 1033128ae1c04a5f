#!/bin/bash
# paired bands against classic strips for the large patches (round 4): config-3 shape, one run per cell
for mode in fast exact; do for k in 9 11; do for sch in view-major paired view-major paired; do
  python bench.py --patch $k --mode $mode --schedule $sch --steps 3 --warmup 1 --no-cpu-baseline --no-planesweep 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode k=$k $sch', d['value'], d['roofline']['avg_launch_ms'], 'rows', d['config']['tile_rows'], 'vpl', d['config']['views_per_launch'])"
done; done; done
