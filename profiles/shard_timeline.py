#!/usr/bin/env python3
"""Read a rocprofv3 kernel trace of profiles/shard_trace.py: per kernel name the average duration, and
the steady-state period of a batch.   python profiles/shard_timeline.py <t_kernel_trace.csv> [last_n]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
last = int(sys.argv[2]) if len(sys.argv) > 2 else 200
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-last:]
t0 = int(rows[0]["Start_Timestamp"])
per = defaultdict(list)
for r in rows:
    per[r["Kernel_Name"][:60]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    print(f"{k:60s} n={len(v):4d} avg_us={sum(v) / len(v):8.2f} total_us={sum(v):9.1f}")
span = (int(rows[-1]["End_Timestamp"]) - t0) / 1e3
busy = sum(sum(v) for v in per.values())
print(f"span_us={span:.1f} sum_of_kernel_us={busy:.1f}")
print("timeline of the last 24 kernels (start_us, dur_us, queue, name):")
for r in rows[-24:]:
    print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:10.1f} {(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:8.2f} q{r.get('Queue_Id', '?'):>3} {r['Kernel_Name'][:70]}")
