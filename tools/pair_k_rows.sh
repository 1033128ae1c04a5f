#!/bin/bash
# paired bands for the large patches: strip height scan (config-3 shape)
mode=${MODE:-fast}
for k in 11 9; do for rows in 0 24 30 34 40 46 54; do
  python bench.py --patch $k --mode $mode --schedule paired --tile-rows $rows --steps 3 --warmup 1 --no-cpu-baseline --no-planesweep 2>/dev/null | \
    python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode k=$k paired rows=$rows', d['value'], d['roofline']['avg_launch_ms'], 'rows', d['config']['tile_rows'])"
done; done
