"""N>1 path on CPU: world_size-2 gloo run of the shard/gather helpers bench.py uses over RCCL."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from dicey_amd.shard import PipelinedGather, gather_bytes, shard_range


def test_shard_range_covers_batch_in_order():
    for nq in (0, 1, 7, 100000, 100001):
        for world in (1, 2, 4, 8):
            parts = [shard_range(nq, r, world) for r in range(world)]
            assert parts[0][0] == 0 and parts[-1][1] == nq
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            assert all(hi - lo <= (nq + world - 1) // world for lo, hi in parts)


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq = 11
    lo, hi = shard_range(nq, rank, world)
    # fake per-rank hit list: one 16-byte record per query plus a rank-dependent tail
    local = torch.tensor([(q * 7 + b) % 256 for q in range(lo, hi) for b in range(16)] + [rank] * (rank * 5), dtype=torch.uint8)
    got = gather_bytes(local, dst=0)
    if rank == 0:
        torch.save([g.clone() for g in got], out)
    else:
        assert got is None
    # pipelined variant: 5 steps of rank- and step-dependent payloads, two in flight
    pg = PipelinedGather(capacity=64 + 10 * rank, device="cpu", depth=2)
    total = 0
    for step in range(5):
        n = 20 + rank * 7 + step
        pg.submit(torch.full((n,), (rank * 16 + step) % 256, dtype=torch.uint8))
        total += n
    got_bytes = pg.finish()
    t = torch.tensor([total], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        assert got_bytes == int(t.item()), (got_bytes, int(t.item()))
        assert pg.cap == 64 + 10 * (world - 1)
        # what travelled is the payload's size class (64 KiB granules + the 8-byte prefix), not the capacity
        assert pg.bytes_moved == 5 * world * (PipelinedGather.ROUND + 8)
        # the contents of the last gather, rank by rank (step 4)
        assert pg.last_received() == [bytes([(r * 16 + 4) % 256]) * (20 + r * 7 + 4) for r in range(world)]
    else:
        assert pg.last_received() == []
    # multi-part payloads are laid out one after the other
    pg2 = PipelinedGather(capacity=40, device="cpu", depth=2)
    pg2.submit([torch.full((3,), rank, dtype=torch.uint8), torch.empty(0, dtype=torch.uint8), torch.full((5,), 100 + rank, dtype=torch.uint8)])
    pg2.finish()
    if rank == 0:
        assert pg2.last_received() == [bytes([r] * 3 + [100 + r] * 5) for r in range(world)]
    dist.barrier()
    dist.destroy_process_group()


def test_gather_bytes_world2_gloo(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = str(tmp_path / "g.pt")
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = torch.load(out)
    assert len(got) == 2
    for rank, g in enumerate(got):
        lo, hi = shard_range(11, rank, 2)
        want = [(q * 7 + b) % 256 for q in range(lo, hi) for b in range(16)] + [rank] * (rank * 5)
        assert g.tolist() == want


def _worker_uneven(rank, world, port):
    """strong-scaling shape: a batch of 13 queries over `world` ranks — ranks beyond the batch get nothing (an empty payload every
    step), the last non-empty rank gets a short shard; payload sizes differ by three orders of magnitude between ranks and steps"""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    nq = 13
    lo, hi = shard_range(nq, rank, world)
    sizes = lambda r, step: (shard_range(nq, r, world)[1] - shard_range(nq, r, world)[0]) * (1 + 40000 * (step % 2)) * 3  # noqa: E731
    pg = PipelinedGather(capacity=4 * 40001 * 3, device="cpu", depth=2)
    sent = 0
    for step in range(6):
        n = sizes(rank, step)
        pg.submit(torch.full((n,), (7 * rank + step) % 256, dtype=torch.uint8))
        sent += n
    got = pg.finish()
    t = torch.tensor([sent], dtype=torch.int64)
    dist.all_reduce(t)
    if rank == 0:
        assert got == int(t.item())
        assert pg.last_received() == [bytes([(7 * r + 5) % 256]) * sizes(r, 5) for r in range(world)]
        assert any(hi2 == lo2 for lo2, hi2 in (shard_range(nq, r, world) for r in range(world))) == (world > 7)
    dist.barrier()
    dist.destroy_process_group()


def test_pipelined_gather_world4_and_world8_uneven_shards():
    for world in (4, 8):
        s = socket.socket()
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
        s.close()
        mp.spawn(_worker_uneven, args=(world, port), nprocs=world, join=True)
