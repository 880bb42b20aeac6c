"""Quick on-box probe: per-stage device time of one 12 MP decode for each subsequence size."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch

import jpeggpu_amd


def main():
    path = os.path.join(ROOT, "tests", "golden", "IMG_6510.JPG")
    data = open(path, "rb").read()
    dev = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0), flush=True)
    for sb in (128, 64, 32):
        dec = jpeggpu_amd.Decoder(sb)
        info = dec.parse_header(data)
        n = dec.get_buffer_size()
        tmp = torch.empty(n + 256, dtype=torch.uint8, device=dev)
        base = (tmp.data_ptr() + 255) // 256 * 256
        planes = [torch.empty((info.sizes_y[c], info.sizes_x[c]), dtype=torch.uint8, device=dev) for c in range(3)]
        st = torch.cuda.current_stream().cuda_stream
        dec.transfer(base, n, st)
        args = ([p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n, st)
        for _ in range(3):
            dec.decode(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            dec.decode(*args)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        dec.set_profiling(True)
        acc = None
        for _ in range(10):
            dec.decode(*args)
            torch.cuda.synchronize()
            ms = dec.stage_ms()
            acc = ms if acc is None else {k: acc[k] + ms[k] for k in ms}
        print("subseq %3d B: %.3f ms/decode (back-to-back)  stages(us): %s  sum %.1f us" % (
            sb, dt * 1e3, {k: round(v * 100, 1) for k, v in acc.items()}, sum(acc.values()) * 100), flush=True)
        dec.cleanup()


if __name__ == "__main__":
    main()
