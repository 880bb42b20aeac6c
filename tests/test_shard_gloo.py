"""N > 1 path on CPU: world_size 2 over gloo. Each rank takes its shard (image i -> rank i mod 2),
produces flat planes (the oracle stands in for the device decode here), rank 0 gathers and checks
that every image arrives, in global order, intact."""
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
        from oracle import oracle
        from tools import jpegsynth

        mine = shard.shard_indices(num_images, rank, world)
        flats = []
        for g in mine:
            d = oracle.decode(jpegsynth.encode(48, 32, seed=1000 + g))
            flats.append(np.concatenate([p.reshape(-1) for p in d.planes]))
        per_image = flats[0].size
        local = torch.from_numpy(np.concatenate(flats))
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
