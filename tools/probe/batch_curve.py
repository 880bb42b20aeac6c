#!/usr/bin/env python3
"""Device time of ONE jpeggpu_ext_decode_batch call by images per call, subsequence size and flow iterations of the
sequence kernel (bench.py's batch_point, swept): what the plan rule of jpeggpu_ext_set_batch_hint is made from.

    python tools/probe/batch_curve.py [--workload cfg2|photo] [--sizes 0,64,128,256] [--images 1,2,4,8,16,32,64] [--sync-iters 0,2,8]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="cfg2")
    ap.add_argument("--sizes", default="0,32,64,128,256")
    ap.add_argument("--images", default="1,2,4,8,16,32,64")
    ap.add_argument("--sync-iters", default="0")
    ap.add_argument("--overlap", default="1")
    ap.add_argument("--hint", type=int, default=-1, help="-1: the images per call")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--stages", action="store_true", help="per-stage device us of the call (HIP events between the launches)")
    a = ap.parse_args()
    import torch

    import bench
    import jpeggpu_amd as jp
    from tools import jpegsynth

    if a.workload == "photo":
        images = [open(bench.PHOTO, "rb").read()]
    else:
        images = [jpegsynth.config(2, seed=s) for s in range(8)]
    dev = torch.device("cuda", 0)
    st = torch.cuda.Stream(device=dev)

    class A:
        pass
    args = A()
    print("workload %s; us per call (us per image) by images per call" % a.workload, flush=True)
    for ov in [int(x) for x in a.overlap.split(",")]:
        for si in [int(x) for x in a.sync_iters.split(",")]:
            for sb in [int(x) for x in a.sizes.split(",")]:
                row = []
                for nb in [int(x) for x in a.images.split(",")]:
                    r = bench.batch_point(args, torch, jp, images, dev, st, nb, a.iters, subseq_bytes=sb, sync_iters=si,
                                          hint=None if a.hint < 0 else a.hint, overlap=ov, stages=a.stages)
                    row.append("%d: %.0f (%.1f) [%dB h%d]" % (nb, r["device_us"], r["device_us_per_image"], r["subsequence_bytes"], r["hypotheses"]))
                    if a.stages:
                        row[-1] += " " + " ".join("%s %.0f" % (k[:6], v) for k, v in r["stage_us"].items())
                print("overlap %d sync_iters %d subseq %3d | %s" % (ov, si, sb, " | ".join(row)), flush=True)


if __name__ == "__main__":
    main()
