"""Query sharding across the GPUs of one node, and the gather of per-rank hit lists to rank 0.

`dicey hunt` treats queries independently (src/hunter.h:291 loop), so a batch shards with no data-path exchange:
rank r takes a contiguous slice, searches it against its own full index replica, and only the variable-length hit
lists travel — one all_gather of byte counts plus one gather of padded byte tensors (RCCL over xGMI when the backend is
"nccl"; the same code runs on gloo for the CPU tests).
"""
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(nq: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous ceil(nq/world) slices, rank order == query order."""
    per = (nq + world - 1) // world
    lo = min(nq, rank * per)
    return lo, min(nq, lo + per)


def gather_bytes(local: torch.Tensor, dst: int = 0, group=None) -> Optional[List[torch.Tensor]]:
    """Gather 1-D uint8 tensors of different lengths to `dst`, returned in rank order (None elsewhere)."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    n = torch.tensor([local.numel()], dtype=torch.int64, device=local.device)
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n, group=group)
    sizes = [int(s.item()) for s in sizes]
    cap = max(max(sizes), 1)
    padded = torch.zeros(cap, dtype=torch.uint8, device=local.device)
    padded[:local.numel()] = local
    if rank == dst:
        bufs = [torch.empty(cap, dtype=torch.uint8, device=local.device) for _ in range(world)]
        dist.gather(padded, bufs, dst=dst, group=group)
        return [b[:s] for b, s in zip(bufs, sizes)]
    dist.gather(padded, None, dst=dst, group=group)
    return None


class DevicePtrView:
    """Exposes a raw HIP device pointer (owned by libdiceygpu) to torch through __cuda_array_interface__."""

    def __init__(self, ptr: int, nbytes: int):
        self.__cuda_array_interface__ = {"shape": (nbytes,), "typestr": "|u1", "data": (ptr, False), "version": 2}


def device_bytes(ptr: int, nbytes: int, device) -> torch.Tensor:
    if nbytes == 0 or not ptr:
        return torch.empty(0, dtype=torch.uint8, device=device)
    return torch.as_tensor(DevicePtrView(ptr, nbytes), device=device)
