#!/usr/bin/env python3
"""One-line summary of a bench.py output line: show.py <label> <file with the JSON line last>"""
import json
import sys

label, path = sys.argv[1], sys.argv[2]
d = json.loads(open(path).read().strip().splitlines()[-1])
k = {n.replace("huff_", "").replace("_kernel", ""): round(v["avg_launch_us"]) for n, v in (d.get("kernels") or {}).items()}
lat = d.get("latency_ms") or {}
lat2 = d.get("latency_ms_device_scan") or {}
r = d.get("roofline") or {}
oc = (d.get("other_configs") or {}).get("config1_photo_12MP_420_dri252") or {}
cpu = d.get("cpu_baseline") or {}
ps = {n: round(v) for n, v in (oc.get("batch_stage_us_serialized") or {}).items() if n != "front"}
if ps:
    print("   photo serialized us/launch", ps, "photo frac", round(oc.get("batch_roofline_frac") or 0, 4))
print("   pass frac", round(r.get("frac") or 0, 4), "pass us", round(r.get("avg_launch_us") or 0), "| photo p50", round(oc.get("p50_ms") or 0, 3),
      "photo batch", round(oc.get("batch_images_per_s") or 0), "| cpu", round(cpu.get("value") or 0, 1), cpu.get("kind"), "all cores", round(cpu.get("all_cores_value") or 0))
print(label, round(d["value"]), "img/s | serialized us/launch", k, "| p50", round(lat.get("p50", 0), 3), "dev-scan", round(lat2.get("p50", 0), 3),
      "| full path", round(d.get("value_full_path") or 0), "| verified", d.get("verified"))
