#!/bin/bash
# kernel durations of the split schedule: groups = 1 (no overlap: clean A / B times) and groups = 2
set -o pipefail
export TMPDIR=/tmp
for g in 1 2; do
  PMC_GROUPS=none PROF_TAG=split_g$g BENCH_ARGS="--no-planesweep --schedule split --split-groups $g --split-rows 16" ./tools/profile.sh || exit 1
  python3 - <<PY
import csv, glob, collections
f = glob.glob("gpurun_out/prof/split_g$g/trace/**/*_kernel_trace.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:90]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(d.items(), key=lambda kv: -sum(kv[1])):
    print("g=$g %-92s n=%4d mean %8.1f us total %9.1f us" % (k, len(v), sum(v) / len(v), sum(v)))
PY
done
