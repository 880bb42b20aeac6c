#!/usr/bin/env python3
"""bench.py -- images/s of the baseline-JPEG decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (destuff -> Huffman sync/write -> IDCT, through
jpeggpu_ext_decode_batch by default or the drop-in jpeggpu_decoder_decode with --mode streams)
over one batch of 12 MP 4:2:0 restart-interval JPEGs per GPU (BASELINE.json configs[1]; with N > 1 it
is configs[2]: images sharded by rank, no data-path collective unless --gather, weak scaling).
Inputs (entropy-coded bytes + table blobs) are resident in HBM before the timed region starts.
Rank 0 prints ONE JSON line; see DESIGN.md "Measurement" for the definition of every field.

For N > 1 the driver launches this file with torch.distributed.run, one rank per GPU (RCCL).
"""
import argparse
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime multiplexes streams onto 4 hardware queues by default, which caps the number of
# kernels in flight at 4 (profiles/r01_bench_kernel_stats_4queues.csv). One decode per stream needs more.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="images per GPU per step")
    ap.add_argument("--mode", default="batch", choices=["batch", "streams"],
                    help="batch: jpeggpu_ext_decode_batch, one launch per stage per group of images; "
                         "streams: the drop-in jpeggpu_decoder_decode, one image per call")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams (batch mode: groups of images, default 4; streams mode: default 16)")
    ap.add_argument("--unique", type=int, default=2, help="distinct synthetic images per rank (seeded)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "photo"])
    ap.add_argument("--subseq-bytes", type=int, default=0, help="0 = library default")
    ap.add_argument("--overlap", type=int, default=1,
                    help="jpeggpu_ext_batch_set_overlap: concurrent parts per batch call (for --streams 1)")
    ap.add_argument("--sync-iters", type=int, default=0,
                    help="flow iterations inside the sequence kernel in batch mode (0 = library default, 1)")
    ap.add_argument("--gather", action="store_true", help="RCCL gather of the decoded planes to rank 0 each step")
    ap.add_argument("--latency-iters", type=int, default=50)
    ap.add_argument("--device-scan", type=int, default=0,
                    help="1: the images of the timed batch use jpeggpu_ext_set_device_scan (marker scan inside the timed region)")
    ap.add_argument("--latency-device-scan", type=int, default=1,
                    help="latency probe: jpeggpu_ext_set_device_scan (restart-marker scan on the device instead of the host walk)")
    ap.add_argument("--latency-subseq-bytes", type=int, default=64,
                    help="subsequence size of the single-image latency probe (64 B: shorter serial chain)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--e2e-rounds", type=int, default=6,
                    help="rounds of the PCIe-inclusive measurement (parse + transfer + decode from pinned host memory); 0 = skip")
    ap.add_argument("--host-threads", type=int, default=6, help="threads of jpeggpu_ext_parse_headers in that measurement")
    return ap.parse_args()


def make_images(args, rank, world):
    """Synthetic 12 MP 4:2:0 JPEGs (DRI = one MCU row). Image i of the global batch has seed i mod
    (unique * world) and is decoded by rank i mod world (jpeggpu_amd.shard), so a rank needs only the
    `unique` seeds of its own shard."""
    if args.workload == "photo":
        with open(os.path.join(ROOT, "tests", "golden", "IMG_6510.JPG"), "rb") as f:
            return [f.read()]
    from jpeggpu_amd import shard
    from tools import jpegsynth

    seeds = shard.shard_indices(args.unique * world, rank, world)
    return [jpegsynth.config(2, seed=s) for s in seeds]


class Slot:
    """One image in flight: its decoder, its temporary device memory and its output planes."""

    def __init__(self, torch, jp, data, device, subseq_bytes, planes_flat=None, planes_off=0, device_scan=False):
        self.dec = jp.Decoder(subseq_bytes or None)
        if device_scan:
            self.dec.set_device_scan(True)
        self.data = data
        self.info = self.dec.parse_header(data)
        self.tmp_size = self.dec.get_buffer_size()
        self.tmp = torch.empty(self.tmp_size + 256, dtype=torch.uint8, device=device)
        self.base = (self.tmp.data_ptr() + 255) // 256 * 256
        nc = self.info.num_components
        sizes = [(self.info.sizes_y[c], self.info.sizes_x[c]) for c in range(nc)]
        self.plane_bytes = sum(h * w for h, w in sizes)
        if planes_flat is None:
            planes_flat = torch.empty(self.plane_bytes, dtype=torch.uint8, device=device)
            planes_off = 0
        self.planes, o = [], planes_off
        for h, w in sizes:
            self.planes.append(planes_flat[o:o + h * w].view(h, w))
            o += h * w
        self.ptrs = [p.data_ptr() for p in self.planes]
        self.pitches = [p.stride(0) for p in self.planes]
        self.layout = self.dec.layout()

    def transfer(self, stream):
        self.dec.transfer(self.base, self.tmp_size, stream)

    def decode(self, stream):
        self.dec.decode(self.ptrs, self.pitches, self.base, self.tmp_size, stream)


def algorithmic_bytes(slot):
    """destuff+Huffman pass B_dh = stuffed scan bytes + 128 B per data unit;
    end-to-end B_e2e = stuffed scan bytes + plane bytes. Plus the dominant kernel's own bytes."""
    lay = slot.layout
    stuffed = lay.transferred_bytes
    ndu = sum(lay.scans[s].num_data_units for s in range(lay.num_scans))
    nsub = sum(lay.scans[s].num_subsequences for s in range(lay.num_scans))
    return {
        "stuffed": stuffed,
        "b_dh": stuffed + 128 * ndu,
        "b_e2e": stuffed + slot.plane_bytes,
        # per-kernel algorithmic bytes
        "front": stuffed,
        "destuff": 2 * stuffed,
        # sync_intra: destuffed bytes read once + subsequence->segment map read + 21 B of state written
        "sync_intra": nsub * lay.subsequence_bytes + nsub * 4 + nsub * 21,
        # sync_tail: about one subsequence of bitstream per subsequence (the live flows decay
        # geometrically, summing to ~0.9 lane-passes) + state read and written
        "sync_inter": nsub * lay.subsequence_bytes + nsub * 40,
        "tails": nsub * 16,
        # write pass: destuffed bytes + state read, coefficient buffer written (128 B per data unit)
        "write": nsub * lay.subsequence_bytes + nsub * 24 + 128 * ndu,
        "idct": 128 * ndu + slot.plane_bytes,
    }


def cpu_baseline(args, data):
    """The oracle (scalar C port of the reference's arithmetic) timed on the host, bounded sample."""
    from oracle import oracle

    oracle.decode(data)  # warm (page in the library)
    n, t0 = 0, time.perf_counter()
    while True:
        oracle.decode(data)
        n += 1
        dt = time.perf_counter() - t0
        if dt >= args.cpu_seconds or n >= 64:
            break
    out = {"value": n / dt, "unit": "images/s", "cores": 1, "kind": "port",
           "sample": "%d sequential decodes of one 12 MP image by oracle/jpeg_oracle.c, %.1f s" % (n, dt)}
    # libjpeg-turbo through Pillow, planes only (draft YCbCr, no colour conversion) when available
    try:
        import io

        from PIL import Image, features

        ver = features.version_feature("libjpeg_turbo")
        if ver:
            m, t1 = 0, time.perf_counter()
            while time.perf_counter() - t1 < min(4.0, args.cpu_seconds) and m < 64:
                im = Image.open(io.BytesIO(data))
                im.draft("YCbCr", im.size)
                im.load()
                m += 1
            out["libjpeg_turbo"] = {"value": m / (time.perf_counter() - t1), "unit": "images/s", "cores": 1,
                                    "version": ver, "sample": "%d Pillow decodes (YCbCr, incl. chroma upsampling)" % m}
    except Exception:  # Pillow is optional on the box
        pass
    return out


def pcie_inclusive(args, torch, jp, slots, groups, nstreams):
    """End to end from pinned host memory, never `value`: per group of the batch, parse the headers on a pool
    of host threads (jpeggpu_ext_parse_headers), enqueue the two H2D copies of every image and one batched
    decode on the group's stream, and go on to the next group while that runs. Images/s over `--e2e-rounds`
    rounds of the whole batch."""
    pinned = []
    for s in slots:
        t = torch.empty(len(s.data), dtype=torch.uint8).pin_memory()
        t.numpy()[:] = memoryview(s.data)
        pinned.append(t)
    per_group = [list(range(g, len(slots), nstreams)) for g in range(nstreams)]
    copied = [torch.cuda.Event() for _ in groups]

    def one_round():
        for g, (bt, scratch, st, _) in enumerate(groups):
            copied[g].synchronize()  # the group's previous copies have executed: its decoders' pinned tables are free again
            idx = per_group[g]
            jp.parse_headers([slots[i].dec for i in idx], [pinned[i] for i in idx], num_threads=args.host_threads)
            for i in idx:
                slots[i].dec.transfer(slots[i].base, slots[i].tmp_size, st.cuda_stream)
            copied[g].record(st)
            bt.decode(scratch.data_ptr(), st.cuda_stream)

    one_round()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.e2e_rounds):
        one_round()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    nbytes = sum(len(s.data) for s in slots)
    return {"value": args.e2e_rounds * len(slots) / dt, "unit": "images/s", "host_threads": args.host_threads,
            "rounds": args.e2e_rounds, "h2d_GBs": args.e2e_rounds * sum(s.layout.transferred_bytes for s in slots) / dt / 1e9,
            "file_bytes_per_image": nbytes // len(slots),
            "protocol": "pinned host JPEGs -> parse_headers (thread pool) -> transfer -> decode_batch, %d groups in flight" % nstreams}


def main():
    args = parse_args()
    import torch
    import torch.distributed as dist

    import jpeggpu_amd as jp
    from jpeggpu_amd import shard

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("bench.py --gpus %d must be launched with torch.distributed.run" % args.gpus)
        args.gpus = world
    assert torch.cuda.is_available(), "bench.py needs a HIP device (there is no CPU fallback)"
    # JPEGGPU_BENCH_BACKEND=gloo rehearses the N > 1 control flow on a box with fewer GPUs than ranks
    # (ranks share devices, timing reduction on the CPU); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("JPEGGPU_BENCH_BACKEND", "nccl")
    dev_index = local_rank % torch.cuda.device_count() if backend == "gloo" else local_rank
    torch.cuda.set_device(dev_index)
    device = torch.device("cuda", dev_index)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group(backend)

    images = make_images(args, rank, world)
    if args.streams <= 0:
        args.streams = 4 if args.mode == "batch" else 16
    nstreams = max(1, min(args.streams, args.batch))
    streams = [torch.cuda.Stream(device=device) for _ in range(nstreams)]

    # all planes of the rank's batch live in one flat tensor so that a gather is one collective
    probe = Slot(torch, jp, images[0], device, args.subseq_bytes, device_scan=bool(args.device_scan))
    per_image = probe.plane_bytes
    planes_flat = torch.empty(per_image * args.batch, dtype=torch.uint8, device=device)
    slots = [Slot(torch, jp, images[i % len(images)], device, args.subseq_bytes, planes_flat, i * per_image,
                  device_scan=bool(args.device_scan))
             for i in range(args.batch)]
    probe.dec.cleanup()
    del probe
    for i, s in enumerate(slots):
        s.transfer(streams[i % nstreams].cuda_stream)
    torch.cuda.synchronize()
    gather_list = None
    if backend != "nccl":
        args.gather = False  # the plane gather is an RCCL collective on device buffers
    if world > 1 and args.gather and rank == 0:
        gather_list = [torch.empty_like(planes_flat) for _ in range(world)]

    # batch mode: the rank's images are split into `nstreams` groups, each group is one
    # jpeggpu_ext_decode_batch call (7 launches) on its own stream
    groups = []
    if args.mode == "batch":
        for g in range(nstreams):
            mine = slots[g::nstreams]
            nscans = sum(s.layout.num_scans for s in mine)
            bt = jp.Batch(nscans)
            scratch = torch.empty(bt.scratch_size, dtype=torch.uint8, device=device)
            bt.set_items([(s.dec, s.ptrs, s.pitches, s.base, s.tmp_size) for s in mine])
            if args.sync_iters > 0:
                bt.set_sync_iterations(args.sync_iters)
            bt.set_overlap(args.overlap)
            groups.append((bt, scratch, streams[g], len(mine)))
        groups[0][0].set_profiling(True)
    else:
        slots[0].dec.set_profiling(True)

    def step():
        if args.mode == "batch":
            for bt, scratch, st, _ in groups:
                bt.decode(scratch.data_ptr(), st.cuda_stream)
        else:
            for i, s in enumerate(slots):
                s.decode(streams[i % nstreams].cuda_stream)
        if world > 1 and args.gather:
            for st in streams:
                torch.cuda.current_stream().wait_stream(st)
            shard.gather_planes(planes_flat, rank, world, dst=0, gather_list=gather_list)

    def barrier():
        if world > 1:
            dist.barrier()

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    profiled = groups[0][0] if args.mode == "batch" else slots[0].dec
    profiled.stage_ms()  # drop the warm-up samples, open the measurement window of the timed region
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    ms_per_step = elapsed / args.steps * 1e3
    value = world * args.batch * args.steps / elapsed
    # mean duration per stage over the timed region (HIP events on the launch stream); in batch mode
    # one launch covers `images_per_launch` images
    stage_us = {k: v * 1e3 for k, v in profiled.stage_ms().items()}
    images_per_launch = groups[0][3] if args.mode == "batch" else 1
    ab = algorithmic_bytes(slots[0])

    out = None
    if rank == 0:
        # single-image latency under the reference's protocol (benchmark/benchmark_jpeggpu.hpp:69-108):
        # parse_header + get_buffer_size + transfer + decode + stream sync, wall clock, pinned input
        s0 = Slot(torch, jp, slots[0].data, device, args.latency_subseq_bytes, device_scan=bool(args.latency_device_scan))
        pinned = torch.empty(len(s0.data), dtype=torch.uint8).pin_memory()
        pinned.numpy()[:] = memoryview(s0.data)
        host_ptr, host_n = pinned.data_ptr(), pinned.numel()
        lat, lat_parse, lat_enqueue = [], [], []
        st = streams[0]
        for it in range(args.latency_iters + 3):
            t1 = time.perf_counter()
            s0.dec.parse_header(host_ptr, host_n)
            n = s0.dec.get_buffer_size()
            t2 = time.perf_counter()
            s0.dec.transfer(s0.base, n, st.cuda_stream)
            s0.dec.decode(s0.ptrs, s0.pitches, s0.base, n, st.cuda_stream)
            t3 = time.perf_counter()
            st.synchronize()
            if it >= 3:
                lat.append((time.perf_counter() - t1) * 1e3)
                lat_parse.append((t2 - t1) * 1e3)
                lat_enqueue.append((t3 - t2) * 1e3)
        # device-only latency of one decode, nothing else running
        s0.dec.set_profiling(True)
        for _ in range(10):
            s0.decode(st.cuda_stream)
        st.synchronize()
        solo = {k: v * 1e3 for k, v in s0.dec.stage_ms().items()}  # us, mean of 10 single-image decodes

        dom = max(stage_us, key=stage_us.get)
        dom_bytes = ab[dom] * images_per_launch
        kernel_names = {"front": "front_windows", "destuff": "destuff_kernel", "sync_intra": "huff_sync_intra",
                        "sync_inter": "huff_sync_tail", "tails": "huff_seq_tails", "write": "huff_write",
                        "idct": "idct_kernel"}
        t_pass_us = sum(stage_us[k] for k in ("front", "destuff", "sync_intra", "sync_inter", "tails", "write"))
        roofline = {
            "bound": "hbm", "kernel": kernel_names[dom],
            "achieved": (dom_bytes / (stage_us[dom] * 1e-6) / 1e9) if dom_bytes and stage_us[dom] > 0 else None,
            "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
            "algorithmic_bytes_per_launch": dom_bytes, "avg_launch_us": stage_us[dom],
            "images_per_launch": images_per_launch,
        }
        if roofline["achieved"] is not None:
            roofline["frac"] = roofline["achieved"] / HBM_PEAK_GBS
        pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        all_kernels = {}
        pmc_traffic = {}
        if os.path.exists(pmc):  # HBM bytes per image from separate rocprofv3 --pmc passes (profiles/)
            try:
                with open(pmc) as f:
                    pmc_traffic = json.load(f)
            except Exception:
                pmc_traffic = {}
        for k, us in stage_us.items():
            t = pmc_traffic.get(kernel_names[k], {}).get("per_image_bytes")
            all_kernels[kernel_names[k]] = {
                "avg_launch_us": us, "algorithmic_bytes_per_launch": ab[k] * images_per_launch,
                "achieved_GBs": ab[k] * images_per_launch / (us * 1e-6) / 1e9 if us > 0 else None,
                "traffic_bytes_per_launch": t * images_per_launch if t else None}
        roofline["traffic"] = all_kernels[kernel_names[dom]]["traffic_bytes_per_launch"]
        out = {
            "metric": "images/s (12 MP 4:2:0 baseline JPEG decode, inputs resident in HBM)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,  # BASELINE.md holds no published number for this metric on this hardware
            "dtype": "int16/int32 fixed point, u8 out", "data": "synthetic",
            "config": {"workload": "cfg2: 4032x3024 4:2:0 interleaved baseline JPEG, DRI=252 (one MCU row), "
                                   "%d seeded images per rank" % len(images) if args.workload == "cfg2"
                       else "cfg1 bytes: tests/golden/IMG_6510.JPG (the reference's 12 MP photo)",
                       "images_per_gpu_per_step": args.batch, "mode": args.mode, "streams": nstreams,
                       "subsequence_bytes": slots[0].layout.subsequence_bytes,
                       "device_scan": bool(slots[0].layout.scans[0].device_scan),
                       "stuffed_scan_bytes": ab["stuffed"], "gather": bool(args.gather and world > 1),
                       "parallelism": "image-sharded x%d" % world},
            "roofline": roofline,
            "roofline_pass": {
                "what": "destuff+Huffman pass (front end, destuff, sync_intra, sync_inter, tails, write), "
                        "B_dh = stuffed scan bytes + 128 B per data unit (SURVEY.md 8d)",
                "bytes_per_image": ab["b_dh"], "sum_launch_us_under_load": t_pass_us,
                "images_per_launch": images_per_launch,
                "throughput_GBs": ab["b_dh"] * value / world / 1e9,
                "frac_of_hbm_peak": ab["b_dh"] * value / world / 1e9 / HBM_PEAK_GBS},
            "roofline_e2e": {"bytes_per_image": ab["b_e2e"], "throughput_GBs": ab["b_e2e"] * value / world / 1e9,
                             "frac_of_hbm_peak": ab["b_e2e"] * value / world / 1e9 / HBM_PEAK_GBS},
            "kernels": all_kernels,
            "stage_us_under_load": stage_us, "stage_us_solo": solo,
            "latency_ms": {"protocol": "parse+size+transfer+decode+sync, 1 image, 1 stream, pinned input",
                           "subsequence_bytes": s0.layout.subsequence_bytes,
                           "device_scan": bool(s0.layout.scans[0].device_scan),
                           "p50": statistics.median(lat), "mean": statistics.fmean(lat), "max": max(lat),
                           "p50_host_parse": statistics.median(lat_parse),
                           "p50_host_enqueue": statistics.median(lat_enqueue),
                           "iters": len(lat), "images_per_s_single_stream": 1e3 / statistics.fmean(lat)},
        }
        if args.mode == "batch" and args.e2e_rounds > 0:
            out["pcie_inclusive"] = pcie_inclusive(args, torch, jp, slots, groups, nstreams)
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, slots[0].data)
    if rank == 0:
        s0.dec.cleanup()
    for bt, _, _, _ in groups:
        bt.destroy()
    for s in slots:
        s.dec.cleanup()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(out))


if __name__ == "__main__":
    main()
