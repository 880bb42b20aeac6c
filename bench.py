#!/usr/bin/env python3
"""bench.py -- images/s of the baseline-JPEG decode hot path on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

One "step" = `--rounds` passes of the hot path (destuff -> Huffman sync / write -> IDCT, through
jpeggpu_ext_decode_batch by default or the drop-in jpeggpu_decoder_decode with --mode streams) over the rank's
batch of 12 MP 4:2:0 restart-interval JPEGs (BASELINE.json configs[1]; with N > 1 the images are sharded by
rank as in configs[2], no data-path collective in the timed region, weak scaling). The defaults make a step
2048 images (~75 ms), so the timed region is about 1.5 s.

`value` is the rate with inputs (entropy-coded bytes + table blobs) resident in HBM when the timed region
starts; `value_full_path` is the rate of the whole boundary protocol from pinned host memory (parse_header +
transfer + decode), which the boundary hands over -- never `value`. After the timed loop the planes of every
distinct image are compared with the CPU oracle (`verified`). Rank 0 prints ONE JSON line; DESIGN.md section 4
defines every field.

For N > 1 the driver launches this file with torch.distributed.run, one rank per GPU (RCCL); the line then also
carries `gather`: BASELINE.json configs[2] itself -- 64 images sharded over the ranks, decoded, and collected
on rank 0 with one RCCL gather per round.
"""
import argparse
import hashlib
import json
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

# The HIP runtime multiplexes streams onto 4 hardware queues by default, which caps the number of
# kernels in flight at 4 (profiles/r01_early_kernel_stats_streams_4queues.csv). One decode per stream needs more.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s (spec)
KERNEL_NAMES = {"front": "front_windows", "destuff": "destuff_kernel", "sync_intra": "huff_sync_intra",
                "sync_inter": "huff_sync_tail", "tails": "huff_seq_tails", "write": "huff_write", "idct": "idct_kernel"}


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=256, help="images per GPU per round (4 groups of 64: one launch per stage and group; BASELINE configs[2] is a batch of 64)")
    ap.add_argument("--rounds", type=int, default=8, help="passes over the batch per step (a step = batch * rounds images)")
    ap.add_argument("--mode", default="batch", choices=["batch", "streams"],
                    help="batch: jpeggpu_ext_decode_batch, one launch per stage per group of images; "
                         "streams: the drop-in jpeggpu_decoder_decode, one image per call")
    ap.add_argument("--streams", type=int, default=0,
                    help="HIP streams (batch mode: groups of images, default 4; streams mode: default 16)")
    ap.add_argument("--unique", type=int, default=16, help="distinct synthetic images per rank (seeded)")
    ap.add_argument("--workload", default="cfg2", choices=["cfg2", "photo"])
    ap.add_argument("--subseq-bytes", type=int, default=0, help="0 = the library's choice for batches")
    ap.add_argument("--overlap", type=int, default=1,
                    help="jpeggpu_ext_batch_set_overlap: concurrent parts per batch call (for --streams 1)")
    ap.add_argument("--sync-iters", type=int, default=0,
                    help="flow iterations inside the sequence kernel in batch mode (0 = library default, 1)")
    ap.add_argument("--gather-rounds", type=int, default=10,
                    help="N > 1: rounds of the configs[2] leg (64 images over the ranks + RCCL gather to rank 0); 0 = skip")
    ap.add_argument("--segment-shard-rounds", type=int, default=5,
                    help="N > 1: rounds of the restart-interval-sharding leg (ONE 39 MP image, every rank decodes its share of "
                         "the restart segments, bands gathered on rank 0); 0 = skip")
    ap.add_argument("--latency-iters", type=int, default=200,
                    help="iterations of the reference's own protocol (1 warm-up + 200, benchmark_common.hpp:39); 0 = skip")
    ap.add_argument("--device-scan", type=int, default=0,
                    help="1: the images of the timed batch use jpeggpu_ext_set_device_scan (marker scan inside the timed region)")
    ap.add_argument("--roofline-launches", type=int, default=6,
                    help="serialized launches (one stream, nothing else on the chip) the roofline figures are averaged over; 0 = skip")
    ap.add_argument("--other-configs", type=int, default=10,
                    help="iterations of the latency protocol on BASELINE.json configs[0] (the reference's photo), [3] and [4]; 0 = skip")
    ap.add_argument("--cpu-seconds", type=float, default=6.0, help="budget of each leg of the CPU baseline")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-verify", action="store_true")
    ap.add_argument("--e2e-rounds", type=int, default=16,
                    help="rounds of the full-path measurement (parse + transfer + decode from pinned host memory); 0 = skip")
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
    end-to-end B_e2e = stuffed scan bytes + plane bytes. Plus each kernel's own bytes."""
    lay = slot.layout
    stuffed = lay.transferred_bytes
    ndu = sum(lay.scans[s].num_data_units for s in range(lay.num_scans))
    nsub = sum(lay.scans[s].num_subsequences for s in range(lay.num_scans))
    return {
        "stuffed": stuffed,
        "b_dh": stuffed + 128 * ndu,
        "b_e2e": stuffed + slot.plane_bytes,
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


def _turbo_decode(data):
    import io

    from PIL import Image

    im = Image.open(io.BytesIO(data))
    im.draft("YCbCr", im.size)
    im.load()


def _turbo_worker(args):
    data, seconds = args
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < seconds and n < 256:
        _turbo_decode(data)
        n += 1
    return n, time.perf_counter() - t0


def usable_cores():
    """Host cores this process may really use: the affinity mask, cut to the cgroup's CPU quota (a GPU box hands a
    one-GPU job a share of a big host) and to 64."""
    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            quota, period = f.read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return max(1, min(n, 64))


def cpu_baseline(args, data):
    """CPU decodes of the same 12 MP image on this box's host cores, bounded samples: the oracle (scalar C port
    of the reference's arithmetic, `kind: port`) on one core, and libjpeg-turbo (through Pillow: planes in
    YCbCr, which includes its chroma upsampling) on one core and on every core the process may use."""
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
    try:
        from PIL import features

        ver = features.version_feature("libjpeg_turbo")
        if ver:
            _turbo_decode(data)
            m, dt1 = _turbo_worker((data, min(4.0, args.cpu_seconds)))
            out["libjpeg_turbo_1_thread"] = {
                "value": m / dt1, "unit": "images/s", "cores": 1, "version": ver,
                "sample": "%d Pillow decodes of the same image (YCbCr planes, incl. chroma upsampling), %.1f s" % (m, dt1)}
            cores = usable_cores()
            import multiprocessing as mp

            with mp.get_context("fork").Pool(cores) as pool:
                t1 = time.perf_counter()
                res = pool.map(_turbo_worker, [(data, min(4.0, args.cpu_seconds))] * cores)
                wall = time.perf_counter() - t1
            total = sum(r[0] for r in res)
            out["libjpeg_turbo_all_cores"] = {
                "value": total / wall, "unit": "images/s", "cores": cores, "version": ver,
                "sample": "%d processes, one image stream each, %d decodes in %.1f s" % (cores, total, wall)}
    except Exception as e:  # Pillow is optional on the box
        out["libjpeg_turbo_error"] = repr(e)
    return out


def pcie_inclusive(args, torch, jp, slots, groups, nstreams):
    """Full path from pinned host memory: per group of the batch, parse the headers on a pool of host threads
    (jpeggpu_ext_parse_headers), enqueue the two H2D copies of every image and one batched decode on the
    group's stream, and go on to the next group while that runs. Images/s over `--e2e-rounds` rounds."""
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
            "rounds": args.e2e_rounds, "seconds": dt,
            "h2d_GBs": args.e2e_rounds * sum(s.layout.transferred_bytes for s in slots) / dt / 1e9,
            "file_bytes_per_image": nbytes // len(slots),
            "protocol": "pinned host JPEGs -> parse_headers (thread pool) -> transfer -> decode_batch, %d groups in flight" % nstreams}


def latency_probe(args, torch, jp, data, device, stream, device_scan):
    """The reference's own protocol (benchmark/benchmark_jpeggpu.hpp:69-108): per image parse_header +
    get_buffer_size + transfer + decode + stream sync, wall clock, pinned input, one warm-up. Library defaults
    unless `device_scan`."""
    s0 = Slot(torch, jp, data, device, 0, device_scan=device_scan)
    pinned = torch.empty(len(data), dtype=torch.uint8).pin_memory()
    pinned.numpy()[:] = memoryview(data)
    host_ptr, host_n = pinned.data_ptr(), pinned.numel()
    lat, lat_parse, lat_enqueue = [], [], []
    warm = 3
    for it in range(args.latency_iters + warm):
        t1 = time.perf_counter()
        s0.dec.parse_header(host_ptr, host_n)
        n = s0.dec.get_buffer_size()
        t2 = time.perf_counter()
        s0.dec.transfer(s0.base, n, stream.cuda_stream)
        s0.dec.decode(s0.ptrs, s0.pitches, s0.base, n, stream.cuda_stream)
        t3 = time.perf_counter()
        stream.synchronize()
        if it >= warm:
            lat.append((time.perf_counter() - t1) * 1e3)
            lat_parse.append((t2 - t1) * 1e3)
            lat_enqueue.append((t3 - t2) * 1e3)
    # device-only time of one decode, nothing else running
    s0.dec.set_profiling(True)
    for _ in range(10):
        s0.decode(stream.cuda_stream)
    stream.synchronize()
    solo = {k: v * 1e3 for k, v in s0.dec.stage_ms().items()}  # us, mean of 10 single-image decodes
    out = {"protocol": "parse+size+transfer+decode+sync, 1 image, 1 stream, pinned input (reference benchmark_jpeggpu.hpp:69-108)",
           "subsequence_bytes": s0.layout.subsequence_bytes, "device_scan": bool(s0.layout.scans[0].device_scan),
           "p50": statistics.median(lat), "mean": statistics.fmean(lat), "max": max(lat),
           "p50_host_parse": statistics.median(lat_parse), "p50_host_enqueue": statistics.median(lat_enqueue),
           "iters": len(lat), "images_per_s_single_stream": 1e3 / statistics.fmean(lat), "stage_us_device": solo}
    s0.dec.cleanup()
    return out


def verify(args, slots, images, torch):
    """Planes of one slot per distinct image against the CPU oracle, after the timed loop: the timed launches
    did the work the number claims."""
    from oracle import oracle

    seen, bad = {}, []
    for s in slots:
        key = id(s.data)
        if key in seen:
            continue
        seen[key] = True
        ref = oracle.decode(s.data)
        for c in range(ref.ncomp):
            got = s.planes[c].cpu().numpy()
            if got.shape != ref.planes[c].shape or hashlib.sha256(got.tobytes()).digest() != hashlib.sha256(ref.planes[c].tobytes()).digest():
                bad.append((len(seen) - 1, c))
    return len(seen), bad


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
    # (ranks share devices, collectives on CPU tensors); the driver's runs use nccl (= RCCL).
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
    if args.subseq_bytes == 0 and args.mode == "batch":
        args.subseq_bytes = jp.BATCH_SUBSEQ_BYTES

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

    def one_round():
        if args.mode == "batch":
            for bt, scratch, st, _ in groups:
                bt.decode(scratch.data_ptr(), st.cuda_stream)
        else:
            for i, s in enumerate(slots):
                s.decode(streams[i % nstreams].cuda_stream)

    def step():
        for _ in range(args.rounds):
            one_round()

    def barrier():
        if world > 1:
            dist.barrier()

    def max_over_ranks(x):
        if world == 1:
            return x
        t = torch.tensor([x], dtype=torch.float64, device=device if backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    for p in planes_flat.split(1 << 30):
        p.zero_()  # the planes hold nothing a previous run could have left behind
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    elapsed = max_over_ranks(time.perf_counter() - t0)

    ms_per_step = elapsed / args.steps * 1e3
    images_per_step = world * args.batch * args.rounds
    value = images_per_step * args.steps / elapsed

    # every distinct image of the timed batch against the oracle
    verified, nverified = None, 0
    if not args.no_verify:
        nverified, bad = verify(args, slots, images, torch)
        ok = 0.0 if bad else 1.0
        if world > 1:
            ok = -max_over_ranks(-ok)  # min over ranks
        verified = ok == 1.0
        if bad:
            sys.stderr.write("bench: planes differ from the oracle: %r\n" % (bad[:8],))

    # BASELINE.json configs[2]: 64 images over the ranks, decoded and gathered on rank 0
    gather = None
    if world > 1 and args.gather_rounds > 0 and args.mode == "batch":
        per_rank = max(1, min(64 // world, args.batch))
        mine = slots[:per_rank]
        bt = jp.Batch(sum(s.layout.num_scans for s in mine))
        scratch = torch.empty(bt.scratch_size, dtype=torch.uint8, device=device)
        bt.set_items([(s.dec, s.ptrs, s.pitches, s.base, s.tmp_size) for s in mine])
        bt.set_overlap(min(4, max(1, per_rank // 8)))
        send = planes_flat[: per_rank * per_image]
        on_gpu = backend == "nccl"
        gl = None
        if rank == 0:
            gl = [torch.empty(per_rank * per_image, dtype=torch.uint8, device=device if on_gpu else "cpu") for _ in range(world)]
        st = streams[0]

        def gather_round():
            with torch.cuda.stream(st):
                bt.decode(scratch.data_ptr(), st.cuda_stream)
                shard.gather_planes(send if on_gpu else send.cpu(), rank, world, dst=0, gather_list=gl)

        gather_round()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.gather_rounds):
            gather_round()
        torch.cuda.synchronize()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t1)
        # what arrived on rank 0 is what every rank sent: 64-bit wrapping sums of the buffers, exchanged separately
        def checksum(t):
            return int(t[: t.numel() // 8 * 8].view(torch.int64).sum().item())

        mine_sum = torch.tensor([checksum(send)], dtype=torch.int64, device=device if on_gpu else "cpu")
        sums = [torch.zeros_like(mine_sum) for _ in range(world)]
        dist.all_gather(sums, mine_sum)
        ok = True
        if rank == 0:
            ok = all(checksum(gl[r]) == int(sums[r].item()) for r in range(world))
        gather = {"what": "BASELINE.json configs[2]: 64 x 12 MP 4:2:0 sharded over the ranks (image i -> rank i mod N), "
                          "decoded, planes gathered on rank 0 (one RCCL gather per round)",
                  "images_per_round": per_rank * world, "rounds": args.gather_rounds,
                  "value": per_rank * world * args.gather_rounds / dt, "unit": "images/s",
                  "ms_per_round": dt / args.gather_rounds * 1e3, "bytes_per_rank": per_rank * per_image,
                  "backend": backend, "gathered_buffers_match_senders": ok}
        bt.destroy()

    # Restart-interval sharding of ONE large image (second bullet): 39 MP 4:2:0 (the size of
    # BASELINE.json configs[3]) with one restart interval per MCU row; rank r decodes segments [r n / N, (r + 1) n / N)
    # into its band of the planes (jpeggpu_ext_set_segment_shard), bands gathered on rank 0.
    segment_shard = None
    if world > 1 and args.segment_shard_rounds > 0:
        from tools import jpegsynth

        big = jpegsynth.encode(7216, 5408, ((2, 2), (1, 1), (1, 1)), True, (7216 + 15) // 16, quality=88, noise=9, seed=4242)
        dec = jp.Decoder(jp.BATCH_SUBSEQ_BYTES)
        dec.set_segment_shard(rank, world)
        info = dec.parse_header(big)
        n = dec.get_buffer_size()
        tmp = torch.empty(n + 256, dtype=torch.uint8, device=device)
        base = (tmp.data_ptr() + 255) // 256 * 256
        shapes = [(info.sizes_y[c], info.sizes_x[c]) for c in range(info.num_components)]
        full = [torch.zeros(h, w, dtype=torch.uint8, device=device) for h, w in shapes]
        rows = [dec.shard_rows(c) for c in range(info.num_components)]
        st = streams[0]
        on_gpu = backend == "nccl"
        dec.transfer(base, n, st.cuda_stream)
        all_rows = [None] * world
        dist.all_gather_object(all_rows, rows)

        def shard_round():
            with torch.cuda.stream(st):
                dec.decode([p.data_ptr() for p in full], [p.stride(0) for p in full], base, n, st.cuda_stream)
                band = torch.cat([full[c][a:a + k].reshape(-1) for c, (a, k) in enumerate(rows)])
                return shard.gather_bands(band if on_gpu else band.cpu(), rank, world, dst=0)

        bands = shard_round()
        torch.cuda.synchronize()
        barrier()
        t1 = time.perf_counter()
        for _ in range(args.segment_shard_rounds):
            bands = shard_round()
        torch.cuda.synchronize()
        barrier()
        dt = max_over_ranks(time.perf_counter() - t1)
        ok = None
        if rank == 0:  # the assembled image against this GPU's own decode of the whole file
            got = shard.assemble_bands(bands, all_rows, shapes)
            whole, _ = jp.decode_to_planes(big, device=str(device), subseq_bytes=jp.BATCH_SUBSEQ_BYTES)
            ok = all(bool(torch.equal(got[c].to(device), whole[c])) for c in range(len(shapes)))
        segment_shard = {"what": "ONE 7216x5408 4:2:0 image (39 MP, DRI = one MCU row) over the ranks by restart segments, "
                                 "bands gathered on rank 0", "file_bytes": len(big), "rounds": args.segment_shard_rounds,
                         "value": args.segment_shard_rounds / dt, "unit": "images/s", "ms_per_image": dt / args.segment_shard_rounds * 1e3,
                         "rows_of_plane_0_per_rank": [r[0] for r in all_rows], "backend": backend,
                         "assembled_equals_whole_decode": ok}
        dec.cleanup()

    out = None
    if rank == 0:
        ab = algorithmic_bytes(slots[0])
        # Roofline figures from SERIALIZED launches: one stream, one batched launch per stage, nothing else on the
        # chip -- what `rocprofv3 --kernel-trace --stats -- python3 bench.py --roofline-only`-style runs reproduce
        # (profiles/). In the timed region the launches of the groups overlap and their durations say nothing
        # about a kernel alone.
        roofline, all_kernels, stage_us, images_per_launch = None, {}, {}, 0
        if args.roofline_launches > 0 and args.mode == "batch":
            bt, scratch, st, images_per_launch = groups[0]
            bt.set_profiling(True)
            for _ in range(args.roofline_launches):
                bt.decode(scratch.data_ptr(), st.cuda_stream)
                st.synchronize()
            stage_us = {k: v * 1e3 for k, v in bt.stage_ms().items()}
            bt.set_profiling(False)
            pmc_traffic = {}
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            if os.path.exists(pmc):  # HBM bytes per image from separate rocprofv3 --pmc passes (profiles/README.md)
                try:
                    with open(pmc) as f:
                        pmc_traffic = json.load(f)
                except Exception:
                    pmc_traffic = {}
            for k, us in stage_us.items():
                t = pmc_traffic.get(KERNEL_NAMES[k], {}).get("per_image_bytes")
                all_kernels[KERNEL_NAMES[k]] = {
                    "avg_launch_us": us, "algorithmic_bytes_per_launch": ab[k] * images_per_launch,
                    "achieved_GBs": ab[k] * images_per_launch / (us * 1e-6) / 1e9 if us > 0 else None,
                    "frac_of_hbm_peak": ab[k] * images_per_launch / (us * 1e-6) / 1e9 / HBM_PEAK_GBS if us > 0 else None,
                    "traffic_bytes_per_launch": t * images_per_launch if t else None}
            dom = max(stage_us, key=stage_us.get)
            d = all_kernels[KERNEL_NAMES[dom]]
            roofline = {"bound": "hbm", "kernel": KERNEL_NAMES[dom], "achieved": d["achieved_GBs"], "peak": HBM_PEAK_GBS,
                        "unit": "GB/s", "frac": d["frac_of_hbm_peak"], "traffic": d["traffic_bytes_per_launch"],
                        "algorithmic_bytes_per_launch": d["algorithmic_bytes_per_launch"], "avg_launch_us": d["avg_launch_us"],
                        "images_per_launch": images_per_launch,
                        "measured": "HIP events on the launch stream, %d serialized launches of %d images, one stream"
                                    % (args.roofline_launches, images_per_launch)}
        t_pass_us = sum(stage_us.get(k, 0.0) for k in ("front", "destuff", "sync_intra", "sync_inter", "tails", "write"))
        out = {
            "metric": "images/s, 12 MP 4:2:0 baseline JPEG decode, inputs resident in HBM "
                      "(value_full_path: from pinned host memory, parse + transfer + decode; latency_ms: p50 of the reference's protocol)",
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None,  # BASELINE.md holds no published number for this metric on this hardware
            "dtype": "int16/int32 fixed point, u8 out", "data": "synthetic",
            "verified": verified, "verified_images": nverified,
            "config": {"workload": "cfg2: 4032x3024 4:2:0 interleaved baseline JPEG, DRI=252 (one MCU row), "
                                   "%d seeded images per rank" % len(images) if args.workload == "cfg2"
                       else "cfg1 bytes: tests/golden/IMG_6510.JPG (the reference's 12 MP photo)",
                       "images_per_gpu_per_step": args.batch * args.rounds, "images_per_step": images_per_step,
                       "batch_per_gpu": args.batch, "rounds_per_step": args.rounds,
                       "mode": args.mode, "streams": nstreams,
                       "subsequence_bytes": slots[0].layout.subsequence_bytes,
                       "device_scan": bool(slots[0].layout.scans[0].device_scan),
                       "stuffed_scan_bytes": ab["stuffed"], "parallelism": "image-sharded x%d" % world},
            "roofline": roofline,
            "roofline_pass": {
                "what": "destuff+Huffman pass (front end, destuff, sync_intra, sync_inter, tails, write), "
                        "B_dh = stuffed scan bytes + 128 B per data unit (SURVEY.md 8d)",
                "bytes_per_image": ab["b_dh"], "sum_serialized_launch_us": t_pass_us,
                "images_per_launch": images_per_launch,
                "frac_of_hbm_peak_serialized": (ab["b_dh"] * images_per_launch / (t_pass_us * 1e-6) / 1e9 / HBM_PEAK_GBS)
                if t_pass_us > 0 else None,
                "throughput_GBs": ab["b_dh"] * value / world / 1e9,
                "frac_of_hbm_peak": ab["b_dh"] * value / world / 1e9 / HBM_PEAK_GBS},
            "roofline_e2e": {"bytes_per_image": ab["b_e2e"], "throughput_GBs": ab["b_e2e"] * value / world / 1e9,
                             "frac_of_hbm_peak": ab["b_e2e"] * value / world / 1e9 / HBM_PEAK_GBS},
            "kernels": all_kernels,
        }
        if gather is not None:
            out["gather"] = gather
        if segment_shard is not None:
            out["segment_shard"] = segment_shard
        if args.latency_iters > 0:
            out["latency_ms"] = latency_probe(args, torch, jp, slots[0].data, device, streams[0], device_scan=False)
            out["latency_ms_device_scan"] = latency_probe(args, torch, jp, slots[0].data, device, streams[0], device_scan=True)
        if args.other_configs > 0:
            # the other BASELINE.json configurations under the same per-image protocol (parity at full size: tests/)
            from tools import jpegsynth

            saved, args.latency_iters = args.latency_iters, args.other_configs
            others = {}
            with open(os.path.join(ROOT, "tests", "golden", "IMG_6510.JPG"), "rb") as f:
                photo = f.read()
            for name, blob in (("config1_photo_12MP_420_dri252", photo),
                               ("config4_39MP_444_three_scans", jpegsynth.config(4)),
                               ("config5_12MP_4_components_8_tables_no_dri", jpegsynth.config(5))):
                r = latency_probe(args, torch, jp, blob, device, streams[0], device_scan=False)
                others[name] = {"file_bytes": len(blob), "p50_ms": r["p50"], "p50_host_parse_ms": r["p50_host_parse"],
                                "max_ms": r["max"], "iters": r["iters"], "stage_us_device": r["stage_us_device"]}
            args.latency_iters = saved
            out["other_configs"] = others
        if args.mode == "batch" and args.e2e_rounds > 0:
            out["pcie_inclusive"] = pcie_inclusive(args, torch, jp, slots, groups, nstreams)
            out["value_full_path"] = out["pcie_inclusive"]["value"]
        if not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(args, slots[0].data)
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
