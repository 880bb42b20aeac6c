#!/usr/bin/env python3
"""Per-pass time of huff_sync_intra for ONE image, from the stamps a probe build leaves
(python jpeggpu_amd/build.py /tmp/probe.so -DJG_PROBE; JPEGGPU_LIB=/tmp/probe.so python tools/probe/sync_stamps.py [photo|cfg2|cfg5] [subseq_bytes])."""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import jpeggpu_amd as jp  # noqa: E402
from jpeggpu_amd import api  # noqa: E402
from tools import jpegsynth  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
sb = int(sys.argv[2]) if len(sys.argv) > 2 else 64
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
data = {"photo": lambda: open(os.path.join(root, "tests", "golden", "IMG_6510.JPG"), "rb").read(),
        "cfg2": lambda: jpegsynth.config(2, seed=0), "cfg5": lambda: jpegsynth.config(5)}[which]()
dev = torch.device("cuda", 0)
dec = jp.Decoder(sb)
info = dec.parse_header(data)
n = dec.get_buffer_size()
tmp = torch.empty(n + 256, dtype=torch.uint8, device=dev)
base = (tmp.data_ptr() + 255) // 256 * 256
planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device=dev) for c in range(info.num_components)]
stream = torch.cuda.Stream()
lib = api.lib()
lib.jpeggpu_probe_read.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
buf = np.zeros((4096, 64), np.uint32)
for it in range(3):
    dec.transfer(base, n, stream.cuda_stream)
    dec.decode([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, stream.cuda_stream)
    stream.synchronize()
    rc = lib.jpeggpu_probe_read(buf.ctypes.data, buf.nbytes, 1)
    assert rc == 0, rc
lay = dec.layout()
nseq = (lay.scans[0].num_subsequences + 239) // 240
b = buf[:nseq].astype(np.int64)
t0 = b[:, 0].min()
last = np.array([np.max(np.nonzero(r)[0]) for r in b])
print(which, "subseq bytes", sb, "workgroups", nseq, "kernel span %.1f us" % ((b.max() - t0) / 100.0))
print("start spread %.1f us; tables %.1f us (median)" % ((b[:, 0].max() - t0) / 100.0, np.median(b[:, 1] - b[:, 0]) / 100.0))
print("speculative pass: median %.1f us, max %.1f us" % (np.median(b[:, 2] - b[:, 1]) / 100.0, (b[:, 2] - b[:, 1]).max() / 100.0))
print("flow iterations per workgroup: median %d, max %d" % (np.median(last - 2), (last - 2).max()))
for i in range(int((last - 2).max())):
    sel = last >= 3 + i
    d = (b[sel, 3 + i] - b[sel, 2 + i]) / 100.0
    print("  iteration %2d: %4d workgroups, median %.2f us, max %.2f us" % (i, sel.sum(), np.median(d), d.max()))
w = int(np.argmax(b.max(axis=1)))
print("slowest workgroup %d: passes (us)" % w, [round((b[w, k + 1] - b[w, k]) / 100.0, 1) for k in range(1, last[w])])
