#!/usr/bin/env python3
"""Summarise a gpurun_out/prof_<tag>/ directory (written by profiles/run_profile.sh) into
profiles/<tag>_*.  Keeps: rocprofv3 kernel stats (verbatim), per-kernel PMC averages with the
gfx950 FETCH_SIZE correction of MI355X_MICROARCH.md §HBM (x2 for wide coalesced reads; units are
KiB), and the raw PMC rows of the tvz kernels."""
import collections
import csv
import json
import os
import shutil
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r1"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"prof_{tag}")
dst = os.path.join(root, "profiles")
shutil.copy(os.path.join(src, "trace", "trace_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
for f in ("bench_under_trace.json",):
    if os.path.exists(os.path.join(src, f)):
        shutil.copy(os.path.join(src, f), os.path.join(dst, f"{tag}_{f}"))
OURS = ("luma_sad", "scene_finalize", "scene_tail", "scene_select", "ts_match_tile", "ts_match_join", "ts_join_build",
        "ts_match_q1", "ts_topk_select", "ts_topk", "ts_prep")
summary = {}
for counter, sub, fn in (("FETCH_SIZE", "pmc_fetch", "fetch"), ("WRITE_SIZE", "pmc_write", "write")):
    path = os.path.join(src, sub, f"{fn}_counter_collection.csv")
    if not os.path.exists(path):
        continue
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == counter
            and any(o in r["Kernel_Name"] for o in OURS)]
    with open(os.path.join(dst, f"{tag}_pmc_{fn}_tvz_kernels.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel", "Grid_Size", "Workgroup_Size", "VGPR_Count", "SGPR_Count", "LDS_Block_Size",
                    "Counter_Name", "Counter_Value_KiB", "Duration_ns"])
        for r in rows:
            name = next(o for o in OURS if o in r["Kernel_Name"])
            if "luma_sad_flat_kernel<" in r["Kernel_Name"]:
                name = r["Kernel_Name"].split("::")[-1].split("(")[0]
            w.writerow([name, r["Grid_Size"], r["Workgroup_Size"], r["VGPR_Count"], r["SGPR_Count"],
                        r["LDS_Block_Size"], counter, r["Counter_Value"],
                        int(r["End_Timestamp"]) - int(r["Start_Timestamp"])])
    # bench.py also launches the scene kernels on small inputs (config0's 480p clips, the e2e leg's
    # micro-batches): per kernel only the launches of the LARGEST grid - the headline workload - count
    big = collections.defaultdict(int)
    for r in rows:
        k = next(o for o in OURS if o in r["Kernel_Name"])
        big[k] = max(big[k], int(r["Grid_Size"]))
    agg = collections.defaultdict(list)
    for r in rows:
        k = next(o for o in OURS if o in r["Kernel_Name"])
        if int(r["Grid_Size"]) == big[k]:
            agg[k].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        summary.setdefault(k, {})[counter + "_KiB_avg"] = sum(v) / len(v)
        summary[k][counter + "_launches"] = len(v)
for k, d in summary.items():
    if "FETCH_SIZE_KiB_avg" in d:
        d["hbm_read_bytes_corrected"] = d["FETCH_SIZE_KiB_avg"] * 1024 * 2   # gfx950: FETCH_SIZE = 1/2 bytes
    if "WRITE_SIZE_KiB_avg" in d:
        d["hbm_write_bytes"] = d["WRITE_SIZE_KiB_avg"] * 1024
json.dump(summary, open(os.path.join(dst, f"{tag}_pmc_summary.json"), "w"), indent=1)
print(json.dumps(summary, indent=1))
