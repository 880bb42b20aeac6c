#!/usr/bin/env python3
"""Concurrency of the kernels in a rocprofv3 --kernel-trace CSV: union busy time, time by number of kernels in
flight, and per-kernel totals inside the window of the batched launches.  timeline.py <kernel_trace.csv>"""
import collections
import csv
import re
import sys

rows = []
for r in csv.DictReader(open(sys.argv[1])):
    if "JobArray" in r["Kernel_Name"]:
        m = re.search(r"(destuff_kernel|huff_sync_intra|huff_sync_tail|huff_seq_tails|huff_write|idct_kernel|front_\w+)", r["Kernel_Name"])
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(1) if m else "other"))
rows.sort()
# take the densest second half (the timed region) of the launches
rows = rows[len(rows) // 3:]
t0, t1 = rows[0][0], max(r[1] for r in rows)
ev = []
for a, b, n in rows:
    ev.append((a, 1))
    ev.append((b, -1))
ev.sort()
depth, last, by_depth = 0, t0, collections.Counter()
for t, d in ev:
    by_depth[depth] += t - last
    last = t
    depth += d
wall = t1 - t0
print("window %.2f ms, %d launches" % (wall / 1e6, len(rows)))
for k in sorted(by_depth):
    print("  %d kernels in flight: %5.1f %%" % (k, 100.0 * by_depth[k] / wall))
tot = collections.Counter()
for a, b, n in rows:
    tot[n] += b - a
for n, v in tot.most_common():
    print("  %-16s sum of durations %.2f ms (%.0f %% of window)" % (n, v / 1e6, 100.0 * v / wall))
