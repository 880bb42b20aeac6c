#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of one round (gpurun_out/) into the summaries committed under profiles/.

  python tools/summarize_profiles.py <kernel_stats_dir> <pmc_FETCH_dir> <pmc_WRITE_dir> <bench.json> <round>

HBM traffic per the guide's recipe (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and
WRITE_SIZE from separate --pmc passes, in KiB; on gfx950 FETCH_SIZE counts half of a wide coalesced
read, so bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 per launch.
"""
import collections
import csv
import json
import os
import re
import shutil
import sys

KERNELS = r"(destuff_kernel|huff_sync_intra|huff_sync_tail|huff_seq_tails|huff_write|idct_kernel)"


def main():
    stats_dir, fetch_dir, write_dir, bench_json, rnd = sys.argv[1:6]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    prof = os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    shutil.copy(os.path.join(stats_dir, "bench_kernel_stats.csv"), os.path.join(prof, "%s_bench_kernel_stats.csv" % rnd))
    shutil.copy(bench_json, os.path.join(prof, "%s_bench_default.json" % rnd))
    out = {}
    for counter, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(os.path.join(d, "bench_counter_collection.csv"))):
            m = re.search(KERNELS, r["Kernel_Name"])
            if r["Counter_Name"] == counter and m and "JobArray" in r["Kernel_Name"]:
                acc[m.group(1)].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {})[counter] = sum(v) / len(v)
    traffic, total = {}, 0.0
    for name, v in sorted(out.items()):
        per = (2 * v.get("FETCH_SIZE", 0) + v.get("WRITE_SIZE", 0)) * 1024 / 32
        total += per
        traffic[name] = {"per_image_bytes": per, "fetch_kib_per_launch_raw": v.get("FETCH_SIZE"),
                         "write_kib_per_launch": v.get("WRITE_SIZE"), "images_per_launch": 32,
                         "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} -- python3 bench.py "
                                    "--steps 3 --warmup 1 --batch 32 --streams 1"}
        print("%-18s fetch %10.0f KiB  write %10.0f KiB  -> %7.2f MB/image" % (
            name, v.get("FETCH_SIZE", 0), v.get("WRITE_SIZE", 0), per / 1e6))
    print("total %.1f MB/image" % (total / 1e6))
    json.dump(traffic, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(prof, "%s_pmc_traffic_batch32.json" % rnd), "w"), indent=1)


if __name__ == "__main__":
    main()
