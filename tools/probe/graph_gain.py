"""How much would a HIP graph save on one image? Replays the launch sequence of jpeggpu_decoder_decode
(host-walked image: kernel arguments by value, nothing but launches) from a captured graph and compares
decode + synchronise with the direct call. Probe only."""
import statistics
import sys
import time

sys.path.insert(0, ".")
import torch

import jpeggpu_amd as jp
from tools import jpegsynth

S420 = ((2, 2), (1, 1), (1, 1))
data = jpegsynth.encode(4032, 3024, S420, restart_interval=252, quality=88, noise=9, seed=5)
for sb in (64, 128):
    dec = jp.Decoder(sb)
    info = dec.parse_header(data)
    n = dec.get_buffer_size()
    tmp = torch.empty(n + 256, dtype=torch.uint8, device="cuda")
    base = (tmp.data_ptr() + 255) // 256 * 256
    planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device="cuda") for c in range(3)]
    ptrs, pit = [p.data_ptr() for p in planes], [p.stride(0) for p in planes]
    st = torch.cuda.Stream()
    dec.transfer(base, n, st.cuda_stream)
    st.synchronize()

    def direct():
        dec.decode(ptrs, pit, base, n, st.cuda_stream)
        st.synchronize()

    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=st):
        dec.decode(ptrs, pit, base, n, st.cuda_stream)

    def replay():
        g.replay()
        torch.cuda.synchronize()

    for name, fn in (("direct", direct), ("graph", replay), ("direct", direct), ("graph", replay)):
        for _ in range(10):
            fn()
        ts = []
        for _ in range(60):
            t = time.perf_counter()
            fn()
            ts.append((time.perf_counter() - t) * 1e3)
        print("sb %d %-7s decode+sync p50 %.3f ms  min %.3f" % (sb, name, statistics.median(ts), min(ts)), flush=True)
    dec.cleanup()
