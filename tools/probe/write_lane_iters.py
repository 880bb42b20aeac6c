#!/usr/bin/env python3
"""Loop iterations of every lane of huff_write for ONE 12 MP image (batched: 256-byte subsequences), from a probe build:
how much of a wave's time its lanes idle because the wave runs until its slowest lane is done.
  python jpeggpu_amd/build.py jpeggpu_amd/lib/exp_probe.so -DJG_PROBE
  JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/exp_probe.so python tools/probe/write_lane_iters.py [photo]"""
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
lib.jpeggpu_probe_read_lane_iters.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
if len(sys.argv) > 1 and sys.argv[1] == "photo":
    data = open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "tests", "golden", "IMG_6510.JPG"), "rb").read()
else:
    data = jpegsynth.config(2, seed=0)
dec = jp.Decoder()
dec.set_batched(True)
info = dec.parse_header(data)
lay = dec.layout()
S = lay.scans[0].num_subsequences
n = dec.get_buffer_size()
tmp = torch.empty(n + 256, dtype=torch.uint8, device=dev)
base = (tmp.data_ptr() + 255) // 256 * 256
planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device=dev) for c in range(info.num_components)]
dec.transfer(base, n, 0)
bt = jp.Batch(1)
scratch = torch.empty(bt.scratch_size, dtype=torch.uint8, device=dev)
bt.set_items([(dec, [p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n)])
bt.decode(scratch.data_ptr(), 0)
torch.cuda.synchronize()
it = np.zeros(1 << 17, np.uint16)
assert lib.jpeggpu_probe_read_lane_iters(it.ctypes.data, it.size) == 0
wave_it = it[:S].astype(np.int64)
it = it[1 << 16:][:S].astype(np.int64)
print("subsequences", S, "iterations of a lane's wave: mean %.1f | symbols per lane: mean %.1f, median %d, p5 %d, p95 %d, max %d" % (
    wave_it.mean(), it.mean(), np.median(it), np.percentile(it, 5), np.percentile(it, 95), it.max()))
print("  symbols / wave iterations = %.3f" % (it.sum() / wave_it.sum()))
rare = np.zeros(1 << 15, np.uint32)
lib.jpeggpu_probe_read_lane_rare.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.jpeggpu_probe_read_lane_rare(rare.ctypes.data, rare.size) == 0
rare = rare[:min(S, 1 << 15)]
print("  rare block: taken in %.1f %% of a wave's iterations; a lane asks for it %.2f times" % (
    100.0 * (rare & 0xFFFF).sum() / wave_it[:rare.size].sum(), (rare >> 16).mean()))
pad = (-S) % 256
w = np.concatenate([it, np.zeros(pad, np.int64)])
for lanes, name in ((64, "wave"), (256, "workgroup")):
    g = w.reshape(-1, lanes)
    print("  per %s: sum of maxima x lanes / sum of iterations = %.3f (lane utilisation %.1f %%)" % (name, g.max(1).sum() * lanes / it.sum(), 100 * it.sum() / (g.max(1).sum() * lanes)))
# what sorting the lanes of a workgroup by length would give (waves of similar lanes)
g = np.sort(w.reshape(-1, 256), axis=1).reshape(-1, 64)
print("  waves of a workgroup's lanes sorted by length: utilisation %.1f %%" % (100 * it.sum() / (g.max(1).sum() * 64)))
g = np.sort(w).reshape(-1, 64)
print("  waves of all lanes sorted by length: utilisation %.1f %%" % (100 * it.sum() / (g.max(1).sum() * 64)))
np.save(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "write_lane_iters.npy"), it.astype(np.uint16))
