#!/usr/bin/env python3
"""Launch-to-launch time of the headline kernel from a rocprofv3 kernel trace of `bench.py`.
With the pipelined block calls two launches of the kernel are in flight at a time (one per lane of the stream
object), so the trace's per-launch DURATION (kernel_stats.csv AverageNs) is ~1.7x the launch-to-launch time that
`roofline.kernel_ms` reports.  This reads the trace itself: the launches on the two lane queues (the two queues that
carry equally many launches of the kernel), the last `steps * blocks` of them = the timed region, span / launches.
usage: bench_span.py <kernel_trace.csv> [steps=20] [blocks=16] [kernel substring]"""
import collections
import csv
import sys

path = sys.argv[1]
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 16
sub = sys.argv[4] if len(sys.argv) > 4 else "firfft_crcf_4096_freq_kernel"
rows = []
with open(path) as f:
    for r in csv.DictReader(f):
        if sub in r["Kernel_Name"]:
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Queue_Id"]))
cnt = collections.Counter(q for _, _, q in rows)
# the two lane queues carry the same number of launches (the blocks alternate between them); the caller's own queue
# carries the one-stream legs and their warm-up passes, which may be more
pairs = [(cnt[a] + cnt[b], a, b) for a in cnt for b in cnt if a < b and abs(cnt[a] - cnt[b]) <= 2]
lanes = list(max(pairs)[1:]) if pairs else [q for q, _ in cnt.most_common(2)]
lane_rows = sorted(r for r in rows if r[2] in lanes)
n = steps * blocks
timed = lane_rows[-n:]
span = (max(e for _, e, _ in timed) - timed[0][0]) / 1e3
dur = [(e - s) / 1e3 for s, e, _ in timed]
print(f"kernel {sub}: {len(rows)} launches in the trace, {len(lane_rows)} on the two lane queues {lanes} "
      f"({', '.join(f'{q}: {c}' for q, c in cnt.most_common())})")
print(f"timed region = the last {n} pipelined launches: span {span:.1f} us = {span / n:.2f} us launch to launch "
      f"({16 * 16777216 / (span / n) / 1e6:.3f} TB/s algorithmic at 2^24 samples per launch = "
      f"{16 * 16777216 / (span / n) / 1e6 / 8:.4f} of 8 TB/s); mean launch duration inside it {sum(dur) / len(dur):.2f} us "
      f"(min {min(dur):.2f}, max {max(dur):.2f}): {sum(dur) / span:.2f} launches in flight on average")
others = [r for r in rows if r[2] not in lanes]
if others:
    d2 = [(e - s) / 1e3 for s, e, _ in others]
    print(f"launches on other queues (the one-stream legs: l3_resident, plain_block_calls): {len(others)}, mean duration "
          f"{sum(d2) / len(d2):.2f} us")
