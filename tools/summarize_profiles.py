#!/usr/bin/env python3
"""Turns the outputs of tools/collect_profiles.sh (gpurun_out/<round>_*) into the summaries committed under profiles/.

  python tools/summarize_profiles.py <round>

HBM traffic per the guide's recipe (/opt/skills/guides/MI355X_MICROARCH.md, "HBM"): FETCH_SIZE and WRITE_SIZE from
separate --pmc passes, in KiB. The guide says FETCH_SIZE counts half of a wide coalesced streaming read on gfx950 and
that every other shape has to be calibrated: tools/probe/fetch_probe.hip does that for the shapes these kernels read
and write with (16 B per lane; 4 B per lane along the rows of a tile; 2-byte gathers of half sectors; 8-byte records;
32-byte sector stores; 8-byte pixel rows) and finds FETCH_SIZE = 0.500 x the bytes moved for EVERY read shape and
WRITE_SIZE = 1.000 x for both store shapes (profiles/<round>_fetch_calibration.json). So every kernel's fetch is
divided by its shape's factor -- round 2 doubled destuff_kernel's only and took the others raw.
"""
import collections
import csv
import glob
import json
import os
import re
import shutil
import sys

KERNELS = r"(destuff_kernel|huff_sync_intra|huff_sync_tail|huff_seq_tails|huff_tail_write|huff_write|idct_kernel)"
# the read / write shape of each kernel in tools/probe/fetch_probe.hip's terms
READ_SHAPE = {"destuff_kernel": "rd_wide16", "huff_sync_intra": "rd_dword_rows", "huff_sync_tail": "rd_dword_rows",
              "huff_seq_tails": "rd_dword_rows", "huff_write": "rd_dword_rows", "huff_tail_write": "rd_dword_rows", "idct_kernel": "rd_u16_units"}
WRITE_SHAPE = {"huff_write": "wr_sector32", "huff_tail_write": "wr_sector32", "idct_kernel": "wr_row8"}
IMAGES_PER_LAUNCH = 64


def counters(d):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            m = re.search(KERNELS, r["Kernel_Name"])
            if m and ("JobArray" in r["Kernel_Name"] or m.group(1) == "huff_tail_write"):
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
    calib = {}
    cal_src = os.path.join(go, "fetch_calibration.json")
    if os.path.exists(cal_src):
        shutil.copy(cal_src, os.path.join(prof, "%s_fetch_calibration.json" % rnd))
    cal_path = os.path.join(prof, "%s_fetch_calibration.json" % rnd)
    if os.path.exists(cal_path):
        calib = json.load(open(cal_path))
    traffic, total = {}, 0.0
    for name, v in sorted(merged.items()):
        fetch, write = v.get("FETCH_SIZE", 0.0), v.get("WRITE_SIZE", 0.0)
        rf = calib.get(READ_SHAPE.get(name, "rd_dword_rows"), {}).get("counter_over_known", 0.5)
        wf = calib.get(WRITE_SHAPE.get(name, "wr_sector32"), {}).get("counter_over_known", 1.0)
        mult = 1.0 / rf
        per = (fetch / rf + write / wf) * 1024 / IMAGES_PER_LAUNCH
        total += per
        traffic[name] = {"per_image_bytes": per, "per_image_bytes_raw": (fetch + write) * 1024 / IMAGES_PER_LAUNCH,
                         "fetch_kib_per_launch_raw": fetch, "write_kib_per_launch": write, "fetch_multiplier": mult,
                         "read_shape": READ_SHAPE.get(name), "write_shape": WRITE_SHAPE.get(name),
                         "images_per_launch": IMAGES_PER_LAUNCH,
                         "valu_insts_per_image": v.get("SQ_INSTS_VALU", 0.0) / IMAGES_PER_LAUNCH,
                         "command": "tools/collect_profiles.sh %s (serialized run: one stream, 64 images per launch)" % rnd}
        print("%-18s fetch %10.0f KiB  write %10.0f KiB  -> %7.2f MB/image   VALU %.2f M/image" % (
            name, fetch, write, per / 1e6, v.get("SQ_INSTS_VALU", 0.0) / IMAGES_PER_LAUNCH / 1e6))
    print("total %.1f MB/image" % (total / 1e6))
    # the serialized run's own kernel durations (rocprofv3 --kernel-trace --stats of the same command): bench.py compares its
    # in-run durations with these and says so when they differ (the counter-derived figures are THIS profile's, not the run's)
    stats = os.path.join(prof, "%s_kernel_stats_serialized.csv" % rnd)
    if os.path.exists(stats):
        for r in csv.DictReader(open(stats)):
            m = re.search(KERNELS, r["Name"])
            if m and ("JobArray" in r["Name"] or m.group(1) == "huff_tail_write") and m.group(1) in traffic:
                traffic[m.group(1)]["profile_avg_us"] = float(r["AverageNs"]) / 1e3
    json.dump(traffic, open(os.path.join(prof, "pmc_traffic.json"), "w"), indent=1)
    json.dump(traffic, open(os.path.join(prof, "%s_pmc_traffic_batch64.json" % rnd), "w"), indent=1)
    json.dump({k: v for k, v in sorted(merged.items())}, open(os.path.join(prof, "%s_pmc_counters_batch64.json" % rnd), "w"), indent=1)
    json.dump({k: v for k, v in sorted(merged.items())}, open(os.path.join(prof, "pmc_counters.json"), "w"), indent=1)  # bench.py reads this one


if __name__ == "__main__":
    main()
