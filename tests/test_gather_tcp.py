"""libdiceygather.so without a GPU: the pipeline that runs over RCCL on a node of GPUs (size agreement one step ahead, exact-size
transfers, ring of three slots, payloads of changing and zero length) over the library's TCP transport, world sizes 2, 4 and 8
as separate processes on 127.0.0.1.  What the root holds after every finish() must be the ranks' last payloads byte for byte, and
the byte / step accounting must add up.  (the RCCL transport itself runs in tests/test_gpu_multirank.py.)"""
import multiprocessing as mp
import os
import socket
import sys

import pytest

from conftest import ROOT

sys.path.insert(0, ROOT)


def payload(rank, step, world):
    """deterministic bytes; some ranks / steps are empty, lengths change every step"""
    if (rank + step) % 5 == 4:
        return b""
    n = 1 + (rank * 7919 + step * 104729) % 70001
    return bytes((rank * 31 + step * 17 + i * 3) & 255 for i in range(min(n, 257))) * (n // 257 + 1)


def _rank_main(world, rank, port, steps, q):
    try:
        import ctypes as C
        from dicey_amd import _gather
        cap = 1 << 10 if rank else 1 << 8      # the agreed capacity is the MAX of what the ranks ask for ...
        c = _gather.Comm(world, rank, max(cap, 300000 if rank == world - 1 else cap), tcp_port=port)
        assert c.max_u64(rank * 10 + 3) == (world - 1) * 10 + 3
        total = 0
        for rnd in range(2):                   # two finish() rounds: the accounting restarts, the ring carries on
            for step in range(steps):
                p = payload(rank, 100 * rnd + step, world)
                buf = C.create_string_buffer(p, max(1, len(p)))
                c.submit(C.addressof(buf) if p else 0, len(p))
                del buf                        # staged: the caller's buffer is free again when submit returns
            got_bytes, got_steps = c.finish()
            assert got_steps == steps
            if rank == 0:
                want = sum(len(payload(r, 100 * rnd + s, world)) for r in range(world) for s in range(steps))
                assert got_bytes == want, (got_bytes, want)
                for r in range(world):
                    assert c.last(r) == payload(r, 100 * rnd + steps - 1, world), (rnd, r)
                total += got_bytes
        c.barrier()
        # a payload above the agreed capacity is refused locally, before anything travels
        big = C.create_string_buffer(400000)
        with pytest.raises(_gather.GatherError):
            c.submit(C.addressof(big), 400000)
        c.close()
        q.put((rank, "ok", total))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, "error", traceback.format_exc() + repr(e)))


@pytest.mark.parametrize("world,steps", [(2, 7), (4, 5), (8, 4), (1, 3)])
def test_gather_pipeline_over_tcp(world, steps):
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_rank_main, args=(world, r, port, steps, q)) for r in range(world)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(timeout=30)
    assert all(r[1] == "ok" for r in res), [r for r in res if r[1] != "ok"]


def test_gather_library_exports_every_symbol_of_its_header():
    import re
    from dicey_amd import _gather
    L = _gather.load()
    hdr = open(os.path.join(ROOT, "include", "dicey_gather.h")).read()
    declared = set(re.findall(r"\b(dg_[a-z0-9_]+)\s*\(", hdr))
    assert declared == set(_gather.SYMBOLS), declared ^ set(_gather.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_rccl_form_refuses_to_open_without_a_device():
    """no CPU stand-in behind the product entry point: dg_comm_open needs a HIP device (skipped on a GPU box)"""
    from dicey_amd import _gather
    if os.path.exists("/dev/kfd"):
        pytest.skip("a GPU box: the RCCL form is exercised by tests/test_gpu_multirank.py")
    with pytest.raises(_gather.GatherError):
        _gather.Comm(1, 0, 1024, unique_id=b"\0" * 128)
