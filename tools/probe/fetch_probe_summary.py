#!/usr/bin/env python3
"""FETCH_SIZE / WRITE_SIZE (KiB) of tools/probe/fetch_probe.hip's kernels against the bytes they are known to move.

  python tools/probe/fetch_probe_summary.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [out.json]

`counter_over_known` is what the counter reports per byte really moved; tools/summarize_profiles.py divides a decode
kernel's raw FETCH_SIZE by the factor of the shape it reads with (1 / 0.5 = the guide's "double it" for wide reads)."""
import collections
import csv
import glob
import json
import os
import sys

BUF = 2 << 30
KNOWN = {"rd_wide16": BUF, "rd_dword_rows": BUF, "rd_u16_units": BUF, "rd_rec8": BUF, "wr_sector32": BUF, "wr_row8": BUF}
NOTE = {"rd_u16_units": "half of every 32-byte sector is asked for; known = whole sectors (what memory must deliver)"}


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def main():
    rd, wr = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
    out = {}
    for name, known in KNOWN.items():
        src = rd if name.startswith("rd_") else wr
        hit = [v for k, v in src.items() if name in k]
        if not hit:
            continue
        got = hit[0] * 1024
        out[name] = {"counter": "FETCH_SIZE" if name.startswith("rd_") else "WRITE_SIZE", "counter_bytes": got,
                     "known_bytes": known, "counter_over_known": got / known}
        if name in NOTE:
            out[name]["note"] = NOTE[name]
        print("%-14s %s = %8.1f MiB for %8.1f MiB moved: x%.3f" % (name, out[name]["counter"], got / 2**20, known / 2**20, got / known))
    if len(sys.argv) > 3:
        json.dump(out, open(sys.argv[3], "w"), indent=1)


if __name__ == "__main__":
    main()
