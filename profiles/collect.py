#!/usr/bin/env python3
"""Copy what a round's documents cite from gpurun_out/ (scratch, untracked) into profiles/ (tracked):
   python profiles/collect.py <tag>
after `bash profiles/run_all.sh <tag>` ran on the GPU box."""
import os
import shutil
import subprocess
import sys

tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = os.path.join(root, "gpurun_out", f"final_{tag}")
dst = os.path.join(root, "profiles")
subprocess.check_call([sys.executable, os.path.join(dst, "summarize.py"), tag], stdout=subprocess.DEVNULL)
subprocess.check_call([sys.executable, os.path.join(dst, "summarize_match.py"), tag], stdout=subprocess.DEVNULL)
for name, out in (("bench.json", f"{tag}_bench.json"), ("predicted_scaling.json", f"{tag}_predicted_scaling.json"),
                  ("find_dup_latency.txt", f"{tag}_find_dup_latency.txt"), ("e2e_service.txt", f"{tag}_e2e_service.txt"),
                  ("match_ab.txt", f"{tag}_match_ab_raw.txt"), ("rebuild_latency.txt", f"{tag}_rebuild_latency.txt"),
                  ("ix_stamps.txt", f"{tag}_ix_stamps.txt"), ("scale_probe.txt", f"{tag}_scale_probe.txt"),
                  ("fuzz_parity.txt", f"{tag}_fuzz_parity.txt"), ("rebuild_trace.txt", f"{tag}_rebuild_trace.txt"),
                  ("topk_probe.txt", f"{tag}_topk_probe.txt"), ("ab_wave.txt", f"{tag}_shard_shapes.txt")):
    p = os.path.join(src, name)
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, out))
# the kernel-trace stats of each matcher workload (rocprofv3 --kernel-trace --stats)
import glob
for w in ("topk", "index", "index1", "join", "q1_100k", "q1_5k", "tile", "shard8", "shard8_wave", "shard8_block"):
    f = glob.glob(os.path.join(root, "gpurun_out", f"pmc_match_{tag}", w, "trace", "**", "*kernel_stats.csv"), recursive=True)
    if f:
        shutil.copy(f[0], os.path.join(dst, f"{tag}_match_{w}_kernel_stats.csv"))
print(sorted(x for x in os.listdir(dst) if x.startswith(tag + "_")))
