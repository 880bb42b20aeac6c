#!/usr/bin/env python3
"""One image decoded on its own, 50 times: the workload for a per-kernel trace of a lone decode
  cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/lone_trace -o t --output-format csv -- python3 tools/probe/lone_trace.py [photo|cfg2|cfg2_nodri|cfg5]"""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
import jpeggpu_amd as jp  # noqa: E402
from tools import jpegsynth  # noqa: E402

root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
which = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
data = {"photo": lambda: open(os.path.join(root, "tests", "golden", "IMG_6510.JPG"), "rb").read(), "cfg2": lambda: jpegsynth.config(2, seed=0),
        "cfg2_nodri": lambda: jpegsynth.encode(4032, 3024, ((2, 2), (1, 1), (1, 1)), True, 0, quality=88, noise=9, seed=0), "cfg5": lambda: jpegsynth.config(5)}[which]()
dec = jp.Decoder()
info = dec.parse_header(data)
n = dec.get_buffer_size()
tmp = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
base = (tmp.data_ptr() + 255) // 256 * 256
planes = [torch.zeros((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device="cuda") for c in range(info.num_components)]
st = torch.cuda.Stream()
for _ in range(50):
    dec.transfer(base, n, st.cuda_stream)
    dec.decode([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, st.cuda_stream)
    st.synchronize()
