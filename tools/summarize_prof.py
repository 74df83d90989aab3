#!/usr/bin/env python3
"""Condense rocprofv3 CSV output (gpurun_out/prof_<tag>/) into a small text summary for profiles/."""
import csv, sys, os, collections
d = sys.argv[1]
kern = sys.argv[2] if len(sys.argv) > 2 else "k_is_valid"
out = []
ks = os.path.join(d, "trace_kernel_stats.csv")
if os.path.exists(ks):
    out.append("== rocprofv3 --kernel-trace --stats (kernel_stats.csv), kernels matching our library ==")
    for r in csv.DictReader(open(ks)):
        if r["Name"].startswith("void k_") or "k_" in r["Name"][:12]:
            out.append(f'{r["Name"][:60]:60s} calls={r["Calls"]} avg_ns={float(r["AverageNs"]):.0f} min_ns={r["MinNs"]} max_ns={r["MaxNs"]} pct={r["Percentage"]}')
for f in sorted(os.listdir(d)):
    if not f.endswith("counter_collection.csv"): continue
    agg = collections.defaultdict(list); meta = {}
    for r in csv.DictReader(open(os.path.join(d, f))):
        if kern not in r["Kernel_Name"]: continue
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
        meta = {k: r[k] for k in ("Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size", "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count") if k in r}
    if not agg: continue
    out.append(f"== {f} ({kern}; per-dispatch mean over {len(next(iter(agg.values())))} dispatches) {meta}")
    for k, v in agg.items():
        out.append(f"  {k:28s} {sum(v)/len(v):.6g}")
print("\n".join(out))
