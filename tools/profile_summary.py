#!/usr/bin/env python3
"""Condense gpurun_out/prof (tools/profile.sh output) into a small text summary for profiles/."""
import collections
import csv
import glob
import os
import sys

PROF = os.environ.get("PROF_DIR", "gpurun_out/prof")
tag = sys.argv[1] if len(sys.argv) > 1 else "run"
note = sys.argv[2] if len(sys.argv) > 2 else ""
KF = (sys.argv[3].split("<")[0] + "<") if len(sys.argv) > 3 else "pm_step_kernel<"
out = [f"# {tag}: {note}",
       "# recipe: tools/profile.sh (rocprofv3 --kernel-trace --stats; separate --pmc passes)",
       "Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
st = sorted(glob.glob(PROF + "/trace/runc/*_kernel_stats.csv"), key=os.path.getmtime)[-1]
for r in csv.DictReader(open(st)):
    if "amvs" in r["Name"] or float(r["Percentage"]) > 1.0:
        out.append(",".join([r["Name"].split("(")[0][:70], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                             r["Percentage"], r["MinNs"], r["MaxNs"]]))
tr = sorted(glob.glob(PROF + "/trace/runc/*_kernel_trace.csv"), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(tr)) if KF in r["Kernel_Name"]]
d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
out.append(f"{KF} per-dispatch ms: " + " ".join(f"{x:.2f}" for x in d))
out.append(f"{KF} registers: arch_vgpr={rows[0]['VGPR_Count']} accum_vgpr={rows[0]['Accum_VGPR_Count']} "
           f"sgpr={rows[0]['SGPR_Count']} lds={rows[0]['LDS_Block_Size']} grid={rows[0]['Grid_Size_X']} wg={rows[0]['Workgroup_Size_X']}")
for name in ("pmc_sq", "pmc_sq2", "pmc_sq3", "pmc_ta", "pmc_fetch", "pmc_write"):
    fs = glob.glob(f"{PROF}/{name}/runc/*_counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(sorted(fs, key=os.path.getmtime)[-1])):
        if KF in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.append(f"pmc[{name}] {KF} {k}: mean_per_launch={sum(v)/len(v):.6g} n={len(v)}")
print("\n".join(out))

# machine-readable traffic entry for bench.py (profiles/traffic.json): pass "kernel_key workload_key"
if len(sys.argv) > 4:
    import json
    kernel_key, workload_key = sys.argv[3], sys.argv[4]
    vals = {}
    for name in ("pmc_fetch", "pmc_write"):
        fs = glob.glob(f"{PROF}/{name}/runc/*_counter_collection.csv")
        if fs:
            acc = collections.defaultdict(list)
            for r in csv.DictReader(open(sorted(fs, key=os.path.getmtime)[-1])):
                if (kernel_key.split("<")[0] + "<") in r["Kernel_Name"]:
                    acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
            for k, v in acc.items():
                vals[k] = sum(v) / len(v)
    if "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
        path = "profiles/traffic.json"
        t = json.load(open(path)) if os.path.exists(path) else {}
        t[kernel_key] = {"workload": workload_key, "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"],
                         "source": tag}
        json.dump(t, open(path, "w"), indent=1)
