"""Image-level sharding of a batch across the GPUs of one node.

Images are independent, so image i goes to rank i mod world and every rank decodes its own shard
with its own decoders and streams: there is no collective on the data path. The only exchange step
is optional: collecting the decoded planes on one rank (RCCL gather over xGMI, one direct link per
peer), for consumers that want the whole batch in one place.
"""
from typing import List, Optional


def shard_indices(num_images: int, rank: int, world: int) -> List[int]:
    """Global image indices decoded by `rank` (image i -> rank i mod world)."""
    if not (0 <= rank < world):
        raise ValueError("rank %d outside world %d" % (rank, world))
    return list(range(rank, num_images, world))


def gather_planes(local_flat, rank: int, world: int, dst: int = 0, gather_list: Optional[list] = None, group=None):
    """Gather every rank's flat uint8 plane buffer (same size on all ranks) on `dst`.

    Returns the list of per-rank buffers on `dst`, None elsewhere. With world == 1 no collective runs.
    """
    import torch
    import torch.distributed as dist

    if world == 1:
        return [local_flat]
    if rank == dst and gather_list is None:
        gather_list = [torch.empty_like(local_flat) for _ in range(world)]
    dist.gather(local_flat, gather_list if rank == dst else None, dst=dst, group=group)
    return gather_list if rank == dst else None


def unshard(per_rank: list, num_images: int, world: int, bytes_per_image: int):
    """Views of the gathered buffers in global image order: result[i] is image i's flat planes."""
    out = []
    for i in range(num_images):
        r, k = i % world, i // world
        out.append(per_rank[r][k * bytes_per_image:(k + 1) * bytes_per_image])
    return out


def gather_bands(local_band, rank: int, world: int, dst: int = 0, group=None):
    """Restart-interval sharding of ONE image (jpeggpu_ext_set_segment_shard): every rank holds the rows of each
    plane its segments cover, packed into one flat uint8 tensor `local_band` (component after component). Bands
    differ in size by a segment or so, so the sizes are exchanged first and the buffers padded to the largest.
    Returns the list of per-rank bands (trimmed) on `dst`, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return [local_band]
    n = torch.tensor([local_band.numel()], dtype=torch.int64, device=local_band.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(sizes)
    padded = local_band if local_band.numel() == cap else torch.cat(
        [local_band, torch.zeros(cap - local_band.numel(), dtype=local_band.dtype, device=local_band.device)])
    out = [torch.empty(cap, dtype=local_band.dtype, device=local_band.device) for _ in range(world)] if rank == dst else None
    dist.gather(padded, out, dst=dst, group=group)
    return [o[:k] for o, k in zip(out, sizes)] if rank == dst else None


def assemble_bands(per_rank_bands, rows_per_rank, plane_shapes):
    """Planes from the gathered bands: rows_per_rank[r][c] = (first_row, num_rows) of component c on rank r
    (jpeggpu_ext_get_shard_rows), plane_shapes[c] = (height, width). Returns one tensor per component."""
    import torch

    planes = [torch.empty(h * w, dtype=per_rank_bands[0].dtype, device=per_rank_bands[0].device).view(h, w) for h, w in plane_shapes]
    for band, rows in zip(per_rank_bands, rows_per_rank):
        o = 0
        for c, (first, count) in enumerate(rows):
            w = plane_shapes[c][1]
            planes[c][first:first + count] = band[o:o + count * w].view(count, w)
            o += count * w
    return planes
