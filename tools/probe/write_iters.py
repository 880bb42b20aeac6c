#!/usr/bin/env python3
"""Time per loop iteration of huff_write in a 64-image launch, from a probe build:
  python jpeggpu_amd/build.py jpeggpu_amd/lib/exp_probe.so -DJG_PROBE [-DJG_EXP_FAKE_REFILL]
  JPEGGPU_LIB=$PWD/jpeggpu_amd/lib/exp_probe.so python tools/probe/write_iters.py"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import jpeggpu_amd as jp  # noqa: E402
from jpeggpu_amd import api  # noqa: E402
from tools import jpegsynth  # noqa: E402

dev = torch.device("cuda", 0)
lib = api.lib()
lib.jpeggpu_probe_read_write.argtypes = [ctypes.c_void_p, ctypes.c_int]
datas = [jpegsynth.config(2, seed=s) for s in range(4)]
keep, items = [], []
for i in range(64):
    dec = jp.Decoder()
    dec.set_batched(True)
    info = dec.parse_header(datas[i % 4])
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
out = (ctypes.c_ulonglong * 4)()
for rep in range(4):
    lib.jpeggpu_probe_read_write(out, 1)
    bt.decode(scratch.data_ptr(), 0)
    torch.cuda.synchronize()
    us = {k: v * 1e3 for k, v in bt.stage_ms().items()}
    lib.jpeggpu_probe_read_write(out, 1)
    s, mx, lanes = out[0], out[2], out[3]
    print("write %.0f us | lanes %d, iterations: sum %d, mean %.1f, max %d | %.3f ns per lane-iteration | sync_intra %.0f tail %.0f idct %.0f" % (
        us["write"], lanes, s, s / max(lanes, 1), mx, us["write"] * 1e3 / max(s, 1) * 1, us["sync_intra"], us["sync_inter"], us["idct"]))
