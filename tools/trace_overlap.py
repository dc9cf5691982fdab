#!/usr/bin/env python3
"""Reads a rocprofv3 kernel-trace CSV and prints the last launches (all kernels, or those matching a substring) with
start/end relative times per queue.  usage: trace_overlap.py <kernel_trace.csv> [kernel substring|-] [count]"""
import csv, sys
rows = []
sub = sys.argv[2] if len(sys.argv) > 2 and sys.argv[2] != "-" else ""
with open(sys.argv[1]) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "?"), r["Kernel_Name"][:40]))
rows.sort()
cnt = int(sys.argv[3]) if len(sys.argv) > 3 else 24
rows = rows[-cnt:]
t0 = rows[0][0]
last_end = {}
for s, e, qid, name in rows:
    gap = (s - last_end[qid]) / 1e3 if qid in last_end else 0.0
    print(f"queue {qid:>3}  start {(s - t0) / 1e3:9.2f}  end {(e - t0) / 1e3:9.2f}  dur {(e - s) / 1e3:7.2f}  gap on its queue {gap:7.2f} us  {name}")
    last_end[qid] = e
print(f"span per launch: {(rows[-1][1] - rows[0][0]) / 1e3 / len(rows):.2f} us")
