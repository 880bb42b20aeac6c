#!/usr/bin/env python3
"""Per-kernel averages of a rocprofv3 --pmc pass (bench_counter_collection.csv under <dir>), batched
(JobArray) launches only, with the kernel's average duration from the trace of the same pass."""
import collections
import csv
import glob
import os
import re
import sys

KERNELS = r"(destuff_kernel|huff_sync_intra|huff_sync_tail|huff_seq_tails|huff_tail_write|huff_write|idct_kernel)"


def main():
    d = sys.argv[1]
    cc = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(cc[0])):
        m = re.search(KERNELS, r["Kernel_Name"])
        if m and ("JobArray" in r["Kernel_Name"] or m.group(1) == "huff_tail_write"):
            acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    dur = collections.defaultdict(list)
    if kt:
        for r in csv.DictReader(open(kt[0])):
            m = re.search(KERNELS, r["Kernel_Name"])
            if m and ("JobArray" in r["Kernel_Name"] or m.group(1) == "huff_tail_write"):
                dur[m.group(1)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k in sorted(acc):
        parts = ["%s=%.4g" % (c, sum(v) / len(v)) for c, v in sorted(acc[k].items())]
        us = sum(dur[k]) / len(dur[k]) if dur[k] else float("nan")
        print("%-16s n=%d avg_us=%.1f  %s" % (k, len(next(iter(acc[k].values()))), us, "  ".join(parts)))


if __name__ == "__main__":
    main()
