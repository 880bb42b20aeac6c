#!/usr/bin/env python3
"""Per-stage device time and p50 of ONE image decoded on its own (library defaults), photo and cfg 2, with and without
the multi-hypothesis speculation (JPEGGPU_MULTI_HYPOTHESIS=0), and checked against the oracle."""
import os, sys, time, statistics
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import numpy as np, torch
import jpeggpu_amd as jp
from oracle import oracle
from tools import jpegsynth
root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..")
inputs = {"photo": open(os.path.join(root, "tests", "golden", "IMG_6510.JPG"), "rb").read(), "cfg2": jpegsynth.config(2, seed=0),
          "cfg2_nodri": jpegsynth.encode(4032, 3024, ((2, 2), (1, 1), (1, 1)), True, 0, quality=88, noise=9, seed=0),
          "cfg5": jpegsynth.config(5),
          "444_dri": jpegsynth.encode(4032, 3024, ((1, 1), (1, 1), (1, 1)), True, 504, quality=88, noise=9, seed=0),
          "422_dri": jpegsynth.encode(4032, 3024, ((2, 1), (1, 1), (1, 1)), True, 252, quality=88, noise=9, seed=0),
          "444_nodri": jpegsynth.encode(4032, 3024, ((1, 1), (1, 1), (1, 1)), True, 0, quality=88, noise=9, seed=0),
          "422_nodri": jpegsynth.encode(4032, 3024, ((2, 1), (1, 1), (1, 1)), True, 0, quality=88, noise=9, seed=0)}
only = sys.argv[1:]
for name, data in inputs.items():
    if only and name not in only:
        continue
    ref = oracle.decode(data)
    for sb in (None,) if os.environ.get("LONE_DEFAULT_ONLY") else (None, 32):
        dec = jp.Decoder(sb)
        pinned = torch.empty(len(data), dtype=torch.uint8).pin_memory(); pinned.numpy()[:] = memoryview(data)
        info = dec.parse_header(pinned.data_ptr(), pinned.numel()); n = dec.get_buffer_size()
        tmp = torch.empty(n + 256, dtype=torch.uint8, device="cuda"); base = (tmp.data_ptr() + 255) // 256 * 256
        planes = [torch.zeros((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device="cuda") for c in range(info.num_components)]
        ptrs = [p.data_ptr() for p in planes]; pit = [p.stride(0) for p in planes]
        st = torch.cuda.Stream(); lat = []
        for it in range(60):
            t = time.perf_counter()
            dec.parse_header(pinned.data_ptr(), pinned.numel()); n = dec.get_buffer_size()
            dec.transfer(base, n, st.cuda_stream); dec.decode(ptrs, pit, base, n, st.cuda_stream); st.synchronize()
            if it >= 10: lat.append((time.perf_counter() - t) * 1e3)
        ok = all(np.array_equal(planes[c].cpu().numpy(), ref.planes[c]) for c in range(ref.ncomp))
        dec.set_profiling(True)
        for _ in range(10):
            dec.decode(ptrs, pit, base, n, st.cuda_stream)
        st.synchronize()
        us = {k: round(v * 1e3, 1) for k, v in dec.stage_ms().items()}
        print(name, "subseq", dec.layout().subsequence_bytes, "hyp", dec.layout().scans[0].hypotheses, "blocks", dec.layout().scans[0].hypothesis_blocks, "tmp MB %.1f" % (n / 1e6), "p50 %.3f ms" % statistics.median(lat), "exact" if ok else "WRONG", us, flush=True)
        dec.cleanup()
