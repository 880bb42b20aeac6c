#!/usr/bin/env python3
"""Turns the outputs of tools/collect_profiles.sh (gpurun_out/<round>_*) into the summaries committed under profiles/.

  python tools/summarize_profiles.py <round>

HBM traffic per the guide's recipe (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE from
separate --pmc passes, in KiB. On gfx950 FETCH_SIZE counts half of a WIDE COALESCED streaming read (16 B per lane):
that holds for destuff_kernel, which is the only kernel here that reads that way; the Huffman kernels fetch the
bitstream in 4-byte refills and the IDCT gathers 4-byte entries, so their raw figure is taken as it is (it matches
their algorithmic bytes). Both are recorded: per_image_bytes uses the per-kernel rule, per_image_bytes_doubled the
blanket 2 x FETCH of round 1.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

KERNELS = r"(destuff_kernel|huff_sync_intra|huff_sync_tail|huff_seq_tails|huff_write|idct_kernel)"
WIDE_READERS = {"destuff_kernel"}
IMAGES_PER_LAUNCH = 64


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(KERNELS, r["Kernel_Name"])
            if m and "JobArray" in r["Kernel_Name"]:
                acc[m.group(1)][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in acc.items()}


def main():
    rnd = sys.argv[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    go, prof = os.path.join(root, "gpurun_out"), os.path.join(root, "profiles")
    os.makedirs(prof, exist_ok=True)
    shutil.copy(os.path.join(go, "%s_bench_default.json" % rnd), os.path.join(prof, "%s_bench_default.json" % rnd))
    for kind in ("default", "serialized"):
        src = glob.glob(os.path.join(go, "%s_stats_%s" % (rnd, kind), "**", "*kernel_stats.csv"), recursive=True)
        if src:
            shutil.copy(src[0], os.path.join(prof, "%s_kernel_stats_%s.csv" % (rnd, kind)))
        js = os.path.join(go, "%s_stats_%s.json" % (rnd, kind))
        if os.path.exists(js):
            shutil.copy(js, os.path.join(prof, "%s_bench_under_rocprof_%s.json" % (rnd, kind)))
    merged = collections.defaultdict(dict)
    for i in range(8):
        d = os.path.join(go, "%s_pmc_%d" % (rnd, i))
        if os.path.isdir(d):
            for k, cs in counters(d).items():
                merged[k].update(cs)
        t = d + ".txt"
        if os.path.exists(t):
            shutil.copy(t, os.path.join(prof, "%s_pmc_pass_%d.txt" % (rnd, i)))
    traffic, total = {}, 0.0
    for name, v in sorted(merged.items()):
        fetch, write = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        mult = 2 if name in WIDE_READERS else 1
        per = (mult * fetch + write) * 1024 / IMAGES_PER_LAUNCH
        total += per
        traffic[name] = {"per_image_bytes": per, "per_image_bytes_doubled": (2 * fetch + write) * 1024 / IMAGES_PER_LAUNCH,
                         "fetch_kib_per_launch_raw": fetch, "write_kib_per_launch": write, "fetch_multiplier": mult,
                         "images_per_launch": IMAGES_PER_LAUNCH,
                         "valu_insts_per_image": v.get("SQ_INSTS_VALU", 0.0) / IMAGES_PER_LAUNCH,
                         "command": "tools/collect_profiles.sh %s (serialized run: one stream, 64 images per launch)" % rnd}
        print("%-18s fetch %10.0f KiB  write %10.0f KiB  -> %7.2f MB/image   VALU %.2f M/image" % (
            name, fetch, write, per / 1e6, v.get("SQ_INSTS_VALU", 0.0) / IMAGES_PER_LAUNCH / 1e6))
    print("total %.1f MB/image" % (total / 1e6))
    json.dump(traffic, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(prof, "%s_pmc_traffic_batch64.json" % rnd), "w"), indent=1)
    json.dump({k: v for k, v in sorted(merged.items())}, open(os.path.join(prof, "%s_pmc_counters_batch64.json" % rnd), "w"), indent=1)


if __name__ == "__main__":
    main()
