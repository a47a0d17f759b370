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


class PipelinedGather:
    """Gather of per-step hit lists to rank `dst` that overlaps with the next step's kernels, and moves the PAYLOAD, not a capacity.

    submit(k) stages step k's payload (so the library may reuse its buffers), prefixes its length and starts an asynchronous
    all_reduce(MAX) of the length; the gather of step k - 1 is launched right behind it, with the size its own all_reduce has
    agreed on by then (rounded up to 64 KiB) — every rank sends and rank `dst` receives exactly that many bytes.  r03 gathered
    a fixed capacity (1.25 x the first step) every step.  No host synchronisation on a collective that was launched in the same
    call; at most `depth` steps are staged or in flight; finish() launches what is left and waits for everything.  `capacity`
    bounds a step's payload (buffers are allocated once, outside any timed region: all_reduce MAX over the ranks)."""

    ROUND = 1 << 16

    def __init__(self, capacity: int, device, dst: int = 0, depth: int = 2, group=None):
        self.group, self.dst, self.depth = group, dst, max(2, depth)
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        cap = torch.tensor([int(capacity)], dtype=torch.int64, device=device)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX, group=group)
        self.cap = int(cap.item())
        alloc = (self.cap + self.ROUND - 1) // self.ROUND * self.ROUND + 8
        self.send = [torch.zeros(alloc, dtype=torch.uint8, device=device) for _ in range(self.depth)]
        self.recv = ([[torch.empty(alloc, dtype=torch.uint8, device=device) for _ in range(self.world)] for _ in range(self.depth)]
                     if self.rank == dst else None)
        # length prefixes come from pinned host memory with a stream-ordered copy: a pageable source would block the host for
        # a tiny transfer once per step
        self.is_cuda = torch.device(device).type == "cuda"
        self.prefix = [torch.zeros(1, dtype=torch.int64).pin_memory() if self.is_cuda else torch.zeros(1, dtype=torch.int64)
                       for _ in range(self.depth)]
        self.size = [torch.zeros(1, dtype=torch.int64, device=device) for _ in range(self.depth)]
        self.size_work = [None] * self.depth
        self.work = [None] * self.depth
        self.sent = [0] * self.depth       # bytes per rank of the slot's gather (the agreed, rounded size + prefix)
        self.n = 0                          # steps staged
        self.launched = 0                   # steps whose gather has been launched
        self.bytes_received = 0
        self.bytes_moved = 0                # what rank dst took in over the wire, padding included
        self.last_slot = None  # receive buffers of the most recently completed gather (rank dst)

    def _drain(self, slot):
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
            if self.rank == self.dst:  # lengths are read only now: no host synchronisation on the submit path
                for buf in self.recv[slot]:
                    self.bytes_received += int(buf[:8].view(torch.int64).item())
                self.bytes_moved += self.sent[slot] * self.world

    def _launch(self, step):
        """gather of a staged step: its size agreement was started one submit ago"""
        slot = step % self.depth
        self.size_work[slot].wait()
        self.size_work[slot] = None
        n = int(self.size[slot].item())
        n = min((n + self.ROUND - 1) // self.ROUND * self.ROUND, self.send[slot].numel() - 8) + 8
        self.sent[slot] = n
        self.work[slot] = dist.gather(self.send[slot][:n], [r[:n] for r in self.recv[slot]] if self.rank == self.dst else None,
                                      dst=self.dst, group=self.group, async_op=True)
        self.launched = step + 1

    def last_received(self) -> List[bytes]:
        """Rank dst, after finish(): the payload every rank sent in the last completed gather, in rank order."""
        if self.rank != self.dst or self.last_slot is None:
            return []
        out = []
        for buf in self.recv[self.last_slot]:
            n = int(buf[:8].view(torch.int64).item())
            out.append(buf[8:8 + n].cpu().numpy().tobytes())
        return out

    def submit(self, payload):
        """payload: a uint8 tensor, or a sequence of them (copied one after the other: no concatenated temporary)."""
        parts = [payload] if isinstance(payload, torch.Tensor) else list(payload)
        total = sum(int(t.numel()) for t in parts)
        if total > self.cap:
            raise RuntimeError(f"payload of {total} bytes exceeds the agreed capacity {self.cap}")
        slot = self.n % self.depth
        self._drain(slot)
        buf = self.send[slot]
        self.prefix[slot][0] = total  # slot's previous gather has been drained, so its prefix copy is long done
        buf[:8].copy_(self.prefix[slot].view(torch.uint8), non_blocking=self.is_cuda)
        self.size[slot].copy_(self.prefix[slot], non_blocking=self.is_cuda)
        at = 8
        for t in parts:
            # staging buffer on another device (the gloo/CPU path): a blocking copy, so the bytes are there before the gather
            # reads them; same device: an ordinary stream-ordered copy
            buf[at:at + t.numel()] = t.to(buf.device) if t.device != buf.device else t
            at += int(t.numel())
        self.size_work[slot] = dist.all_reduce(self.size[slot], op=dist.ReduceOp.MAX, group=self.group, async_op=True)
        self.n += 1
        while self.launched < self.n - 1:  # the step before this one: its size has been agreed on while this step computed
            self._launch(self.launched)

    def finish(self) -> int:
        while self.launched < self.n:
            self._launch(self.launched)
        for k in range(self.depth):  # oldest first
            self._drain((self.n + k) % self.depth)
        if self.n and self.rank == self.dst:
            self.last_slot = (self.n - 1) % self.depth
        return self.bytes_received
