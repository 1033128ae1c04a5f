#!/usr/bin/env python3
"""Condense gpurun_out/prof/<tag> (tools/profile.sh output) into a small text summary for profiles/.

    tools/profile_summary.py PROF_SUBDIR TITLE NOTE [KERNEL_PREFIX WORKLOAD_KEY]

KERNEL_PREFIX (e.g. "pm_step_fast_kernel<7,4>") selects the kernel whose per-dispatch times, registers
and counters are listed; with WORKLOAD_KEY the FETCH_SIZE / WRITE_SIZE means are also written to
profiles/traffic.json, which bench.py reports as `roofline.traffic`.
"""
import collections
import csv
import glob
import json
import os
import sys

sub, tag = sys.argv[1], sys.argv[2]
note = sys.argv[3] if len(sys.argv) > 3 else ""
kernel_key = sys.argv[4] if len(sys.argv) > 4 else "pm_step_fast_kernel<7,4>"
workload_key = sys.argv[5] if len(sys.argv) > 5 else None
PROF = os.path.join(os.environ.get("PROF_DIR", "gpurun_out/prof"), sub)
KF = kernel_key.split("<")[0] + "<"


def latest(pattern):
    fs = sorted(glob.glob(pattern), key=os.path.getmtime)
    return fs[-1] if fs else None


out = [f"# {tag}: {note}",
       "# recipe: tools/profile.sh (rocprofv3 --kernel-trace --stats; separate --pmc passes), summarised by tools/profile_summary.py",
       "Name,Calls,TotalDurationNs,AverageNs,Percentage,MinNs,MaxNs"]
st = latest(PROF + "/trace/*/*_kernel_stats.csv")
if st:
    for r in csv.DictReader(open(st)):
        if "amvs::" in r["Name"] and float(r["Percentage"]) > 0.05:
            out.append(",".join([r["Name"].split("(")[0][:80], r["Calls"], r["TotalDurationNs"], r["AverageNs"],
                                 r["Percentage"], r["MinNs"], r["MaxNs"]]))
tr = latest(PROF + "/trace/*/*_kernel_trace.csv")
if tr:
    rows = [r for r in csv.DictReader(open(tr)) if KF in r["Kernel_Name"]]
    by = collections.OrderedDict()
    for r in rows:
        by.setdefault(r["Kernel_Name"].split("(")[0], []).append(r)
    tot_n = tot_t = 0
    for name, rs in by.items():
        d = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rs]
        tot_n += len(d); tot_t += sum(d)
        out.append(f"{name}: {len(d)} dispatches, mean {sum(d)/len(d):.4f} ms; per dispatch: " + " ".join(f"{x:.2f}" for x in d))
        # rocprofv3's VGPR_Count is the allocation in its own granule units, not the count hipcc
        # reports (-Rpass-analysis=kernel-resource-usage); both are listed in DESIGN.md section 5
        out.append(f"{name}: rocprofv3 fields arch_vgpr={rs[0]['VGPR_Count']} accum_vgpr={rs[0]['Accum_VGPR_Count']} "
                   f"sgpr={rs[0]['SGPR_Count']} lds={rs[0]['LDS_Block_Size']} grid={rs[0]['Grid_Size_X']} wg={rs[0]['Workgroup_Size_X']}")
    if tot_n:
        out.append(f"{KF}*: {tot_n} dispatches in the traced step(s), launch-weighted mean {tot_t/tot_n:.4f} ms, sum {tot_t:.2f} ms")
vals = {}
for name in ("pmc_sq", "pmc_sq2", "pmc_fetch", "pmc_write"):
    f = latest(f"{PROF}/{name}/*/*_counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if KF in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        vals[k] = sum(v) / len(v)
        out.append(f"pmc[{name}] {KF}* {k}: mean_per_launch={sum(v)/len(v):.6g} n={len(v)}")
if "TCC_HIT_sum" in vals and "TCC_MISS_sum" in vals:
    out.append(f"L2 hit rate {vals['TCC_HIT_sum']/(vals['TCC_HIT_sum']+vals['TCC_MISS_sum']):.3f}; "
               f"misses are 128-byte lines: {vals['TCC_MISS_sum']*128/1e9:.2f} GB per launch from the fabric "
               "(FETCH_SIZE tallies them at 64 B, MI355X_MICROARCH.md)")
print("\n".join(out))
if workload_key and "FETCH_SIZE" in vals and "WRITE_SIZE" in vals:
    path = "profiles/traffic.json"
    t = json.load(open(path)) if os.path.exists(path) else {}
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from bench import kernel_source_hash            # the counters belong to THESE kernel sources
    t[kernel_key] = {"workload": workload_key, "fetch_kib": vals["FETCH_SIZE"], "write_kib": vals["WRITE_SIZE"],
                     "fetch_correction": 2.0, "source": tag, "source_hash": kernel_source_hash(kernel_key)}
    for extra in ("TCC_HIT_sum", "TCC_MISS_sum"):
        if extra in vals:
            t[kernel_key][extra.lower()] = vals[extra]
    json.dump(t, open(path, "w"), indent=1)
