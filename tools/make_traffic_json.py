#!/usr/bin/env python3
"""Condense the PMC passes of tools/profile.sh (gpurun_out/prof_<tag>/) into profiles/<round>/k_is_valid_traffic.json, the file
bench.py reads `roofline.traffic` / `roofline.valu` from.  The file is stamped with the sha256 of the libmopa_hip.so it was
measured on (the library travels to the GPU box unchanged); bench.py ignores it for any other build.

    python tools/make_traffic_json.py gpurun_out/prof_r02 profiles/r02 [kernel-name-prefix [output-file-name]]
"""
import csv
import glob
import hashlib
import json
import os
import sys

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
src, dst = sys.argv[1], sys.argv[2]
prefix = sys.argv[3] if len(sys.argv) > 3 else "k_is_valid_v5"
N = 1 << 20


def mean_counter(pattern, counter):
    vals, name = [], None
    for f in sorted(glob.glob(os.path.join(src, pattern))):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void ", "")
            if k.startswith(prefix) and r["Counter_Name"] == counter and int(r["Grid_Size"]) >= 256 * 256:
                vals.append(float(r["Counter_Value"]))
                name = k
    return (sum(vals) / len(vals), len(vals), name) if vals else (None, 0, None)


fetch, nf, kname = mean_counter("pmc_fetch*counter_collection*.csv", "FETCH_SIZE")
write, nw, _ = mean_counter("pmc_write*counter_collection*.csv", "WRITE_SIZE")
valu, nv, _ = mean_counter("pmc_sq_*counter_collection*.csv", "SQ_INSTS_VALU")
avg_ns = None
ks = os.path.join(src, "trace_kernel_stats.csv")
if os.path.exists(ks):
    for r in csv.DictReader(open(ks)):
        if r["Name"].replace("void ", "").startswith(prefix):
            avg_ns = float(r["AverageNs"])
            break
if fetch is None or write is None:
    sys.exit(f"no FETCH_SIZE / WRITE_SIZE rows for {prefix} under {src}")
sha = hashlib.sha256(open(os.path.join(ROOT, "mopa_rl_amd", "csrc", "libmopa_hip.so"), "rb").read()).hexdigest()
sys.path.insert(0, ROOT)
import bench as _bench
from mopa_rl_amd.scene import planner_inputs as _pi
NQ = int(_pi(_bench.ENV).model.nq)          # (MOPA_BENCH_ENV selects the scene, as for the profiled bench run)
out = {
    "kernel": kname, "states_per_launch": N, "lib_sha256": sha, "k1_sources_sha256": _bench.k1_sources_sha256(),
    "FETCH_SIZE_KB": fetch, "WRITE_SIZE_KB": write, "dispatches_averaged": [nf, nw, nv],
    "traffic_bytes_per_launch": int((fetch + write) * 1024),
    "algorithmic_bytes_per_launch": int(N * (7 * 8 + 1 + NQ * 8 / 256)),
    "scene": _bench.ENV,
    "kernel_avg_ns_rocprof": avg_ns,
    "note": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes (tools/profile.sh), raw KB x 1024, mean per dispatch of the "
            "1 048 576-state launches; the guide's gfx950 half-count correction is calibrated for 16 B/lane streaming reads and is NOT "
            "applied (this kernel's accesses are 8 B/lane: uncalibrated; reads may be up to 2x more). Infinity-Cache hits are counted.",
}
if valu is not None:
    out["valu_insts_per_launch"] = valu
    out["valu_note"] = "SQ_INSTS_VALU per dispatch (pmc_sq pass)"
# both stamps (library, K1 sources) are written HERE, in the run that read the counters, and sealed: bench.py recomputes `record_sha256` over
# every other field and ignores a record that does not carry its own seal (a stamp edited afterwards breaks it)
out["record_sha256"] = _bench.traffic_record_seal(out)
os.makedirs(dst, exist_ok=True)
name = "k_is_valid_traffic.json" if len(sys.argv) <= 4 else sys.argv[4]
json.dump(out, open(os.path.join(dst, name), "w"), indent=1)
print(json.dumps(out, indent=1))
