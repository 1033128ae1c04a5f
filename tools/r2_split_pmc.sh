#!/bin/bash
PMC_ONLY=1 PMC_GROUPS="pmc_fetch pmc_write" PROF_TAG=split_pmc BENCH_ARGS="--no-planesweep --schedule split --split-groups 1 --split-rows 8 --iters 2" ./tools/profile.sh > gpurun_out/prof_a.log 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for grp in ("pmc_fetch", "pmc_write"):
    for f in glob.glob("gpurun_out/prof/split_pmc/%s/**/*_counter_collection.csv" % grp, recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0]
            if "pm_s" in k:
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    d = {c: sum(v) / len(v) for c, v in cs.items()}
    hit = d.get("TCC_HIT_sum", 0) / max(d.get("TCC_HIT_sum", 0) + d.get("TCC_MISS_sum", 0), 1)
    print(k[:60], "FETCH_KiB %.3g WRITE_KiB %.3g L2 hit %.3f misses %.3g -> traffic 2*FETCH+WRITE = %.2f GB" % (
        d.get("FETCH_SIZE", 0), d.get("WRITE_SIZE", 0), hit, d.get("TCC_MISS_sum", 0),
        (2 * d.get("FETCH_SIZE", 0) + d.get("WRITE_SIZE", 0)) * 1024 / 1e9))
PY
