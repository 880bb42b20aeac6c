"""N > 1 path on CPU: world_size 2 over gloo. Each rank takes its shard (image i -> rank i mod 2) through the host side
of the real flow -- parse_header, buffer sizing and plane geometry through the C ABI, the flat plane buffer laid out
as bench.py lays it out -- and only the device decode itself is stood in for by the oracle (there is no GPU here);
rank 0 gathers and checks that every image arrives, in global order, intact. The flow with the HIP decode in it runs
as a -m gpu test (tests/test_gpu_api.py::test_bench_multi_rank_control_flow_over_gloo)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from jpeggpu_amd import shard


def test_shard_indices_partition():
    for n in (0, 1, 7, 64):
        for world in (1, 2, 4, 8):
            got = sorted(i for r in range(world) for i in shard.shard_indices(n, r, world))
            assert got == list(range(n))
            sizes = [len(shard.shard_indices(n, r, world)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    assert shard.shard_indices(64, 3, 8) == list(range(3, 64, 8))  # BASELINE config 3: 8 per GPU
    with pytest.raises(ValueError):
        shard.shard_indices(4, 2, 2)


def _worker(rank, world, port, num_images, result):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import jpeggpu_amd
        from oracle import oracle
        from tools import jpegsynth

        mine = shard.shard_indices(num_images, rank, world)
        per_image, local = None, None
        for k, g in enumerate(mine):
            data = jpegsynth.encode(48, 32, seed=1000 + g)
            dec = jpeggpu_amd.Decoder()
            dec.set_batched(True)
            info = dec.parse_header(data)  # host side of the real flow: geometry, plan, buffer size
            assert dec.get_buffer_size() > 0 and dec.layout().subsequence_bytes in (32, 64, 128, 256)
            sizes = [(info.sizes_y[c], info.sizes_x[c]) for c in range(info.num_components)]
            dec.cleanup()
            if local is None:  # one flat tensor for the rank's planes, image after image, component after component
                per_image = sum(h * w for h, w in sizes)
                local = torch.empty(per_image * len(mine), dtype=torch.uint8)
            d = oracle.decode(data)  # stands in for transfer + decode into the slots of the flat buffer
            o = k * per_image
            for c, (h, w) in enumerate(sizes):
                assert d.planes[c].shape == (h, w)
                local[o:o + h * w] = torch.from_numpy(d.planes[c].reshape(-1))
                o += h * w
        gathered = shard.gather_planes(local, rank, world, dst=0)
        if rank == 0:
            images = shard.unshard(gathered, num_images, world, per_image)
            ok = True
            for g in range(num_images):
                d = oracle.decode(jpegsynth.encode(48, 32, seed=1000 + g))
                want = np.concatenate([p.reshape(-1) for p in d.planes])
                ok &= bool(np.array_equal(images[g].numpy(), want))
            result.put(ok)
        else:
            assert gathered is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_gather_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, 6, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert result.get() is True


def _band_worker(rank, world, port, result):
    """Restart-interval sharding of one image: the library's host parse says which rows a rank owns; the oracle's
    rows stand in for the device decode; rank 0 gathers the bands and puts the image back together."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import jpeggpu_amd
        from oracle import oracle
        from tests import cases

        data = cases.matrix()["dri_row"]
        ref = oracle.decode(data)
        dec = jpeggpu_amd.Decoder()
        dec.set_segment_shard(rank, world)
        info = dec.parse_header(data)
        rows = [dec.shard_rows(c) for c in range(info.num_components)]
        dec.cleanup()
        band = torch.from_numpy(np.concatenate([ref.planes[c][a:a + n].reshape(-1) for c, (a, n) in enumerate(rows)]))
        all_rows = [None] * world
        dist.all_gather_object(all_rows, rows)
        bands = shard.gather_bands(band, rank, world, dst=0)
        if rank == 0:
            planes = shard.assemble_bands(bands, all_rows, [p.shape for p in ref.planes])
            result.put(all(np.array_equal(planes[c].numpy(), ref.planes[c]) for c in range(ref.ncomp)))
        else:
            assert bands is None
        dist.barrier()
    finally:
        dist.destroy_process_group()


def test_two_rank_segment_bands_over_gloo():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    ctx = mp.get_context("spawn")
    result = ctx.SimpleQueue()
    procs = [ctx.Process(target=_band_worker, args=(r, 2, port, result)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    assert result.get() is True


def test_bench_starts_its_own_ranks_when_launched_plainly():
    """`python bench.py --gpus 2 ...` with no WORLD_SIZE in the environment -- the shape of the driver's N = 1 command --
    must become the launcher of its own two ranks (a child torch.distributed.run, no exec). There is no GPU here, so
    each rank gets as far as bench.py's "needs a HIP device" assertion; what is checked is that argument handling did
    not refuse the command, that two ranks really started, and that the launcher passes their exit code on."""
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_PORT=str(port), JPEGGPU_BENCH_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    if torch.cuda.is_available():
        pytest.skip("covered with a device by tests/test_gpu_api.py::test_bench_multi_rank_control_flow_over_gloo")
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", "photo", "--no-cpu",
                        "--steps", "1", "--warmup", "0"], env=env, capture_output=True, text=True, timeout=300)
    assert "must be launched with" not in r.stderr
    assert r.returncode != 0  # the ranks' failure is the launcher's
    assert r.stderr.count("bench.py needs a HIP device") >= 2, r.stderr[-2000:]
