"""Many independent queries: sharding across GPUs and the result gather.

Queries are independent (own tree, own RNG stream, shared read-only grid), so the path shards
with no data-path collective: query q runs on rank q % world_size.  The one
exchange step is the gather of the fixed-size result slabs at the end of a batch, done with
torch.distributed (backend "nccl" == RCCL over xGMI on the GPU node, "gloo" in the CPU tests).
torch is used here only as plumbing for the process group and the collective; the planner
package itself does not import it.
"""
from typing import List


def shard_queries(total: int, world_size: int, rank: int) -> List[int]:
    """Indices of the queries that `rank` owns: round-robin, q -> q % world_size."""
    if world_size < 1 or not (0 <= rank < world_size):
        raise ValueError(f"bad rank {rank} of {world_size}")
    return list(range(rank, total, world_size))


def owner_of(q: int, world_size: int) -> int:
    return q % world_size


def local_slot(q: int, world_size: int) -> int:
    """Position of query q inside its owner's batch."""
    return q // world_size


class DeviceBlock:
    """Zero-copy view of a device allocation for torch (``__cuda_array_interface__``)."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (int(nbytes),), "typestr": "|u1", "data": (int(ptr), False), "version": 2}


def gather_result_blocks(local_block, group=None):
    """All-gather equally sized uint8 result slabs (1-D tensors) from every rank.

    Returns a (world_size, nbytes) tensor on the same device as `local_block`; row r is rank r's
    slab.  One collective per batch: with RCCL this is a single ncclAllGather over xGMI."""
    import torch
    import torch.distributed as dist

    ws = dist.get_world_size(group)
    out = torch.empty((ws, local_block.numel()), dtype=local_block.dtype, device=local_block.device)
    dist.all_gather_into_tensor(out.view(-1), local_block.contiguous(), group=group)
    return out


def unpack_slab(slab, Q: int, node_stride: int):
    """Split one rank's slab (uint8 numpy array) into (vcost[Q,stride] f64, nodes[Q,stride] u32,
    parent[Q,stride] i32) -- the layout of rrt_batch_result_block."""
    import numpy as np

    a = np.asarray(slab, dtype=np.uint8)
    n8 = Q * node_stride * 8
    n4 = Q * node_stride * 4
    vcost = a[:n8].view(np.float64).reshape(Q, node_stride)
    nodes = a[n8:n8 + n4].view(np.uint32).reshape(Q, node_stride)
    parent = a[n8 + n4:n8 + 2 * n4].view(np.int32).reshape(Q, node_stride)
    return vcost, nodes, parent
