#!/usr/bin/env python3
"""Randomised soak of the GPU path against the oracle: many synthetic images of random geometry, sampling,
restart interval, quality, table kind and scan structure, decoded through the batch API (random subsequence
size, sync iterations and overlap parts) and through the drop-in call; every plane must be bit-exact.
  python tools/soak_gpu.py [seconds] [seed]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch

    import jpeggpu_amd as jp
    from oracle import oracle
    from tools import jpegsynth

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
    samplings = [((1, 1),) * 3, ((2, 2), (1, 1), (1, 1)), ((2, 1), (1, 1), (1, 1)), ((1, 2), (1, 1), (1, 1)),
                 ((4, 1), (1, 1), (1, 1)), ((1, 1),), ((2, 1), (1, 1), (1, 1), (2, 1)), ((2, 2), (1, 1))]
    t0, rounds, images, sharded = time.time(), 0, 0, 0
    while time.time() - t0 < budget:
        sb = int(rng.choice([32, 64, 128, 256]))
        items, refs, keep = [], [], []
        total_scans = 0
        for _ in range(int(rng.integers(4, 24))):
            ss = samplings[int(rng.integers(len(samplings)))]
            w, h = int(rng.integers(1, 900)), int(rng.integers(1, 700))
            if rng.random() < 0.1:
                w, h = int(rng.integers(1500, 4100)), int(rng.integers(1000, 3100))
            mcus = ((w + 8 * max(s[0] for s in ss) - 1) // (8 * max(s[0] for s in ss)))
            dri = int(rng.choice([0, 0, 1, 2, 7, mcus, mcus * 3 + 1, 100]))
            data = jpegsynth.encode(w, h, ss, interleaved=bool(rng.random() < 0.8) or len(ss) == 1,
                                    restart_interval=dri, quality=int(rng.choice([3, 30, 60, 85, 95, 100])),
                                    optimize=bool(rng.random() < 0.5), noise=int(rng.integers(0, 40)),
                                    fill_bytes=int(rng.choice([0, 0, 0, 1, 3])), seed=int(rng.integers(1 << 30)),
                                    qmax=int(rng.choice([0, 0, 255, 65535])))
            ref = oracle.decode(data)
            # half of the decoders get a fixed subsequence size, the others what the library picks for the images per call
            # they are told to expect (jpeggpu_ext_set_batch_hint, round 5: 0 / 1 = the lone decode's plan)
            dec = jp.Decoder(sb if rng.random() < 0.5 else None)
            dec.set_batch_hint(int(rng.choice([0, 1, 2, 8, 20, 64])))
            dec.set_device_scan(bool(rng.integers(2)))  # batches mix host-walked and device-scanned images
            info = dec.parse_header(data)
            n = dec.get_buffer_size()
            tmp = torch.empty(n + 256, dtype=torch.uint8, device="cuda:0")
            base = (tmp.data_ptr() + 255) // 256 * 256
            planes = [torch.full((info.sizes_y[c], info.sizes_x[c]), 0xCD, dtype=torch.uint8, device="cuda:0")
                      for c in range(info.num_components)]
            dec.transfer(base, n, 0)
            total_scans += dec.layout().num_scans
            items.append((dec, [p.data_ptr() for p in planes], [p.stride(0) for p in planes], base, n))
            keep.append((dec, tmp, planes, data))
            refs.append(ref)
        # a third of the calls as if they filled the chip (read when the batch is created): the full batch's kernels -- one flow
        # iteration, marks, and the tail kernel's parts + the write pass's sequences as one launch (huff_tail_write)
        full = rng.random() < 0.34
        os.environ["JPEGGPU_EXP_KEEP_FLOWS_BELOW"] = "0" if full else str(220000)
        batch = jp.Batch(total_scans)
        scratch = torch.empty(batch.scratch_size, dtype=torch.uint8, device="cuda:0")
        batch.set_items(items)
        if full and rng.random() < 0.25:
            batch.set_fused_tail(False)  # ... or as two launches
        if not full and rng.random() < 0.5:  # a caller's cap (marks + tail kernel); else the library's choice: these calls are small, so
            batch.set_sync_iterations(int(rng.choice([1, 1, 2, 5, 255, 256])))  # every flow stays in the sequence kernel
        batch.set_overlap(int(rng.integers(1, 5)))
        batch.decode(scratch.data_ptr(), 0)
        torch.cuda.synchronize()
        for (dec, tmp, planes, data), ref in zip(keep, refs):
            for c in range(ref.ncomp):
                assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), ("batch", rounds, len(data), c)
                planes[c].fill_(0x3C)
        # the same images through the drop-in call, half of them with the device-side marker scan
        singles, shard_keep = [], []
        for (dec, tmp, planes, data), ref, it in zip(keep, refs, items):
            dec.cleanup()
            dec2 = jp.Decoder(sb)
            dev = bool(rng.integers(2))
            dec2.set_device_scan(dev)
            world = int(rng.choice([1, 1, 2, 3, 5]))
            if world > 1:
                # restart-interval sharding (jpeggpu_ext_set_segment_shard): the first world - 1 shares through
                # decoders of their own, the last one through dec2 below; files that cannot be cut say so
                try:
                    shares = []
                    for r in range(world - 1):
                        d3 = jp.Decoder(sb)
                        d3.set_segment_shard(r, world)
                        d3.parse_header(data)
                        n3 = d3.get_buffer_size()
                        tmp3 = torch.empty(n3 + 256, dtype=torch.uint8, device="cuda:0")
                        base3 = (tmp3.data_ptr() + 255) // 256 * 256
                        d3.transfer(base3, n3, 0)
                        d3.decode(it[1], it[2], base3, n3, 0)
                        shares.append((d3, tmp3))
                    shard_keep.extend(shares)
                    dec2.set_segment_shard(world - 1, world)
                    sharded += 1
                except jp.JpegGpuError as e:
                    assert e.status == jp.Status.NOT_SUPPORTED, e.status
                    dec2.set_segment_shard(0, 1)
            dec2.parse_header(data)
            n2 = dec2.get_buffer_size()
            tmp2 = torch.empty(n2 + 256, dtype=torch.uint8, device="cuda:0")
            base2 = (tmp2.data_ptr() + 255) // 256 * 256
            dec2.transfer(base2, n2, 0)
            dec2.decode(it[1], it[2], base2, n2, 0)
            singles.append((dec2, tmp2, base2))
        torch.cuda.synchronize()
        for (dec, tmp, planes, data), ref, (dec2, tmp2, base2) in zip(keep, refs, singles):
            st = dec2.device_status(base2, 0)
            assert st == jp.Status.SUCCESS, st
            for c in range(ref.ncomp):
                assert np.array_equal(planes[c].cpu().numpy(), ref.planes[c]), ("single", rounds, len(data), c)
            dec2.cleanup()
        for d3, _ in shard_keep:
            d3.cleanup()
        batch.destroy()
        rounds += 1
        images += len(keep)
        if rounds % 20 == 0:  # a run that stays silent for minutes is taken to be hung on the GPU pool
            print("  %d rounds, %d images, %.0f s" % (rounds, images, time.time() - t0), flush=True)
    assert jp.fused_tail_timeouts() == 0, jp.fused_tail_timeouts()
    print("soak ok: %d rounds, %d images (%d of them also decoded as restart-segment shares), %.0f s" % (rounds, images, sharded, time.time() - t0))


if __name__ == "__main__":
    main()
