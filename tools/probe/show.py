#!/usr/bin/env python3
"""One-line summary of a bench.py output line: show.py <label> <file with the JSON line last>"""
import json
import sys

label, path = sys.argv[1], sys.argv[2]
d = json.loads(open(path).read().strip().splitlines()[-1])
k = {n.replace("huff_", "").replace("_kernel", ""): round(v["avg_launch_us"]) for n, v in (d.get("kernels") or {}).items()}
lat = d.get("latency_ms") or {}
lat2 = d.get("latency_ms_device_scan") or {}
print(label, round(d["value"]), "img/s | serialized us/launch", k, "| p50", round(lat.get("p50", 0), 3), "dev-scan", round(lat2.get("p50", 0), 3),
      "| full path", round(d.get("value_full_path") or 0), "| verified", d.get("verified"))
