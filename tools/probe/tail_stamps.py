#!/usr/bin/env python3
"""Where huff_sync_tail spends its time in a 64-image batch launch, from the stamps of a probe build:
  python jpeggpu_amd/build.py jpeggpu_amd/lib/exp_probe.so -DJG_PROBE
  JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/exp_probe.so python tools/probe/tail_stamps.py [photo]"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import jpeggpu_amd as jp  # noqa: E402
from jpeggpu_amd import api  # noqa: E402
from tools import jpegsynth  # noqa: E402

dev = torch.device("cuda", 0)
lib = api.lib()
lib.jpeggpu_probe_read_tail.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
if len(sys.argv) > 1 and sys.argv[1] == "photo":
    datas = [open(os.path.join(root, "tests", "golden", "IMG_6510.JPG"), "rb").read()]
else:
    datas = [jpegsynth.config(2, seed=s) for s in range(16)]
keep, items = [], []
for i in range(64):
    dec = jp.Decoder()
    dec.set_batched(True)
    info = dec.parse_header(datas[i % len(datas)])
    n = dec.get_buffer_size()
    tmp = torch.empty(n + 256, dtype=torch.uint8, device=dev)
    base = (tmp.data_ptr() + 255) // 256 * 256
    planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device=dev) for c in range(info.num_components)]
    dec.transfer(base, n, 0)
    keep.append((dec, tmp, planes))
    items.append((dec, [p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n))
bt = jp.Batch(64)
scratch = torch.empty(bt.scratch_size, dtype=torch.uint8, device=dev)
bt.set_items(items)
bt.set_profiling(True)
for rep in range(3):
    bt.decode(scratch.data_ptr(), 0)
    torch.cuda.synchronize()
us = {k: v * 1e3 for k, v in bt.stage_ms().items()}
buf = np.zeros((128, 64), np.uint32)
assert lib.jpeggpu_probe_read_tail(buf.ctypes.data, buf.size) == 0
b = buf.astype(np.int64)
used = b[:, 63] != 0
flows, trips = b[used, 63] & 0xFFFF, b[used, 63] >> 16
b = b[used]
t0 = b[:, 0].min()
end = np.array([r[2 + min(int(t), 30) - 1] for r, t in zip(b, trips)])
print("tail stage %.0f us (sync_intra %.0f, write %.0f) | %d parts sampled" % (us["sync_inter"], us["sync_intra"], us["write"], used.sum()))
print("flows per part: median %d, max %d | loop trips per part: median %d, p90 %d, max %d" % (np.median(flows), flows.max(), np.median(trips), np.percentile(trips, 90), trips.max()))
print("start spread %.1f us | list building: median %.1f us | part lifetime: median %.1f, p90 %.1f, max %.1f us | last end - first start %.1f us" % (
    (b[:, 0].max() - t0) / 100.0, np.median(b[:, 1] - b[:, 0]) / 100.0, np.median(end - b[:, 0]) / 100.0, np.percentile(end - b[:, 0], 90) / 100.0,
    (end - b[:, 0]).max() / 100.0, (end.max() - t0) / 100.0))
per = []
for r, t in zip(b, trips):
    k = min(int(t), 30)
    per.extend(((r[2:2 + k] - r[1:1 + k]) / 100.0).tolist())
per = np.array(per)
print("one trip of the flow loop: median %.1f us, p90 %.1f us, max %.1f us (%d trips)" % (np.median(per), np.percentile(per, 90), per.max(), per.size))
worst = int(np.argmax(end - b[:, 0]))
k = min(int(trips[worst]), 30)
print("slowest part: %d flows, %d trips, trips (us):" % (flows[worst], trips[worst]), np.round((b[worst, 2:2 + k] - b[worst, 1:1 + k]) / 100.0, 1).tolist(),
      "flows per trip:", b[worst, 32:32 + k].tolist())
# time per trip by the number of flows in it (a trip of at most one flow per wave is decoded by the wave: wave_decode_subsequence)
by = {}
for r, t in zip(b, trips):
    k = min(int(t), 30)
    for d, f in zip(((r[2:2 + k] - r[1:1 + k]) / 100.0).tolist(), r[32:32 + k].tolist()):
        by.setdefault("1-4" if f <= 4 else "5-16" if f <= 16 else "17-64" if f <= 64 else ">64", []).append(d)
print("trip time by flows in the trip:", {k2: (len(v), round(float(np.median(v)), 1)) for k2, v in sorted(by.items())})
