#!/usr/bin/env python3
"""What a device-side marker scan could save on a MULTI-scan file: host parse time (all of it the
marker walk) against the whole latency of the reference's protocol, for BASELINE.json configs[3] (39 MP 4:4:4, three
scans). Only the LAST scan could skip the host walk: the bytes of every earlier scan must be walked on the host
anyway to find the next SOS."""
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch

import jpeggpu_amd as jp
from tools import jpegsynth

data = jpegsynth.config(4)
pinned = torch.empty(len(data), dtype=torch.uint8).pin_memory()
pinned.numpy()[:] = memoryview(data)
dec = jp.Decoder()
info = dec.parse_header(pinned.data_ptr(), pinned.numel())
n = dec.get_buffer_size()
lay = dec.layout()
tmp = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
base = (tmp.data_ptr() + 255) // 256 * 256
planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device="cuda") for c in range(3)]
ptrs, pit = [p.data_ptr() for p in planes], [p.stride(0) for p in planes]
st = torch.cuda.Stream()
lat, par = [], []
for it in range(40):
    t = time.perf_counter()
    dec.parse_header(pinned.data_ptr(), pinned.numel())
    n = dec.get_buffer_size()
    t1 = time.perf_counter()
    dec.transfer(base, n, st.cuda_stream)
    dec.decode(ptrs, pit, base, n, st.cuda_stream)
    st.synchronize()
    if it >= 5:
        lat.append((time.perf_counter() - t) * 1e3)
        par.append((t1 - t) * 1e3)
print("cfg4 39 MP, %d scans, file %.1f MB: p50 latency %.3f ms, of which host parse %.3f ms (%.0f %%); a device scan of the last scan "
      "alone could save about a third of the parse: %.3f ms" % (lay.num_scans, len(data) / 1e6, statistics.median(lat), statistics.median(par),
                                                               100 * statistics.median(par) / statistics.median(lat), statistics.median(par) / 3))
