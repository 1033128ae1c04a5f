#!/bin/bash
# Why are tall strips slower?  L2 hit rate and wave-time counters of pm_step for several strip heights
# (16 views per launch, full schedule), one PMC pass each.
set -o pipefail
export TMPDIR=/tmp
for th in ${HEIGHTS:-24 72 155}; do
  OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof/tall_$th
  mkdir -p $OUT
  ARGS="bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-planesweep --views-per-launch 16 --tile-rows $th"
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT -- python3 $ARGS > $OUT.log 2>&1 || { echo "pmc failed for $th"; tail -3 $OUT.log; exit 1; }
  python3 - "$OUT" $th <<'PY'
import csv, glob, sys, collections
f = glob.glob(sys.argv[1] + "/*/*_counter_collection.csv")[0]
acc = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    if "pm_step_fast_kernel<7, 4, 2" in r["Kernel_Name"]:
        acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
print("rows", sys.argv[2], {k: f"{v:.4g}" for k, v in m.items()}, "L2 hit", round(m["TCC_HIT_sum"] / (m["TCC_HIT_sum"] + m["TCC_MISS_sum"]), 3))
PY
  find $OUT -name "*.db" -delete; find $OUT -name "*.csv" -delete
done
