#!/bin/bash
# Extended mode under rocprofv3 (7 views 1920x1080, 4 iterations): kernel trace + separate PMC passes;
# prints per-kernel durations and the counters of xpm_sweep_kernel (run under gpurun).
set -o pipefail
export TMPDIR=/tmp
OUT=${GRAFT_REPO_ROOT:-$PWD}/gpurun_out/prof/${PROF_TAG:-ext3}
mkdir -p $OUT
CMD="tools/extended_eval.py --height 1080 --width 1920 --iters 4"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 $CMD > $OUT/trace.log 2>&1 || { echo trace failed; tail -5 $OUT/trace.log; exit 1; }
pmc() { name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $CMD > "$OUT/$name.log" 2>&1 || { echo "pmc $name failed"; tail -3 "$OUT/$name.log"; }; }
pmc pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU
pmc pmc_write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
pmc pmc_fetch FETCH_SIZE
for f in $(find "$OUT" -name "*_kernel_trace.csv" -o -name "*_counter_collection.csv"); do
  head -1 "$f" > "$f.tmp"; grep "amvs::" "$f" >> "$f.tmp"; mv "$f.tmp" "$f"
done
find $OUT -name "*.db" -delete
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
st = glob.glob(out + "/trace/*/*_kernel_stats.csv")[0]
print("Name,Calls,TotalDurationNs,AverageNs,Percentage")
for r in csv.DictReader(open(st)):
    if "amvs::" in r["Name"] and float(r["Percentage"]) > 0.05:
        print(",".join([r["Name"].split("(")[0][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"]]))
for name in ("pmc_sq", "pmc_write", "pmc_fetch"):
    fs = glob.glob(f"{out}/{name}/*/*_counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if "xpm_sweep_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print(f"pmc[{name}] xpm_sweep_kernel {k}: mean_per_launch={sum(v)/len(v):.6g} n={len(v)}")
PY
