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
# split by launch kind and iteration: the sweep's dispatches come in schedule order -- per step and
# view group, per iteration 2 propagation launches (template argument 1) and `samples` refinement
# launches (argument 2) -- so the n-th propagation dispatch belongs to iteration (n // 2) % iters, the
# n-th refinement dispatch to iteration (n // samples) % iters
ITERS, SAMPLES = int(os.environ.get("PM_ITERS", "8")), int(os.environ.get("PM_SAMPLES", "8"))


def kind_of(name):
    m = name.split(KF)[1].split(">")[0].split(",") if KF in name else []
    return {"1": "PROP", "2": "REFINE"}.get(m[2].strip(), None) if len(m) > 2 else None


def by_iteration(rows, value):
    """rows in dispatch order -> {kind: [mean per iteration]}"""
    seq = collections.defaultdict(list)
    for r in rows:
        k = kind_of(r["Kernel_Name"])
        if k:
            seq[k].append(value(r))
    out_ = {}
    for k, v in seq.items():
        per = 2 if k == "PROP" else SAMPLES
        acc = [[] for _ in range(ITERS)]
        for n, x in enumerate(v):
            acc[(n // per) % ITERS].append(x)
        out_[k] = [sum(a) / len(a) if a else float("nan") for a in acc]
    return out_


if tr:
    rows = sorted((r for r in csv.DictReader(open(tr)) if KF in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
    for k, v in by_iteration(rows, lambda r: (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6).items():
        out.append(f"{KF}* {k} mean ms by iteration: " + " ".join(f"{x:.3f}" for x in v))
for name in ("pmc_sq", "pmc_write", "pmc_fetch"):
    f = latest(f"{PROF}/{name}/*/*_counter_collection.csv")
    if not f:
        continue
    rows = [r for r in csv.DictReader(open(f)) if KF in r["Kernel_Name"]]
    for cname in sorted({r["Counter_Name"] for r in rows}):
        if cname not in ("SQ_INSTS_VALU", "SQ_INSTS_VMEM_RD", "SQ_WAIT_INST_ANY", "SQ_BUSY_CYCLES", "TCC_HIT_sum", "TCC_MISS_sum",
                         "FETCH_SIZE", "WRITE_SIZE"):
            continue
        sel = sorted((r for r in rows if r["Counter_Name"] == cname), key=lambda r: int(r["Dispatch_Id"]))
        for k, v in by_iteration(sel, lambda r: float(r["Counter_Value"])).items():
            out.append(f"pmc[{name}] {cname} {k} by iteration: " + " ".join(f"{x:.4g}" for x in v))
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
    for extra in ("TCC_HIT_sum", "TCC_MISS_sum", "SQ_INSTS_VALU"):
        if extra in vals:
            t[kernel_key][extra.lower()] = vals[extra]
    json.dump(t, open(path, "w"), indent=1)
