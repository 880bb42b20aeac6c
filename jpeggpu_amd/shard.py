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
