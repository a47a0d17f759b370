"""The N>1 entry points, run for real on whatever the box has: `bench.py --gpus 2` launches its own ranks (as the driver's
torch.distributed.run command would) and the C++ host shards a batch over DICEY_DEVICES.  On a 1-GPU box both ranks /
both device slots use GPU 0 (--same-device, gloo; DICEY_DEVICES=0,0): the control flow, the sharding and the gather are
the multi-GPU ones, only the wires differ."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("batches", [1, 3])
def test_bench_gpus2_spawns_ranks_and_gathers_hit_lists(tmp_path, batches):
    """batches = 1: every step searches the same queries, so the gathered bytes per step equal the ranks' local lists; 3: the rotating
    stream of r04 (step k searches batch k mod 3), where the last step's lists are what the dump holds"""
    dump = str(tmp_path / "gather")
    env = dict(os.environ)
    env.pop("WORLD_SIZE", None)
    env.pop("RANK", None)
    env.pop("LOCAL_RANK", None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--same-device", "--backend", "gloo", "--genome-size", "2e6",
           "--queries", "2000", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--dump-gather", dump, "--batches", str(batches),
           "--detail-out", str(tmp_path / "detail.json"), "--cli-queries", "0"]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1 and p.stdout.rstrip("\n").splitlines()[-1] == lines[0], p.stdout[-2000:]  # the contract line is stdout's last line
    assert len(lines[0]) < 4096
    out = json.loads(lines[0])
    detail = json.load(open(tmp_path / "detail.json"))
    assert detail["value"] == out["value"] and "phases_ms" in detail and "phases_ms" not in out
    assert out["n_gpus"] == 2 and out["scaling"] == "weak"
    assert out["gathered_bytes_per_step"] > 0
    assert out["value"] > 0 and out["steps"] == 3
    total = 0
    for r in range(2):
        local = open(os.path.join(dump, f"local_{r}.bin"), "rb").read()
        got = open(os.path.join(dump, f"gathered_{r}.bin"), "rb").read()
        assert local and got == local, (r, len(local), len(got))
        total += len(local)
    if batches == 1:
        assert abs(out["gathered_bytes_per_step"] - total) < 1e-6 * total + 1  # the same queries every step


def test_bench_rccl_gather_path_with_one_rank(tmp_path):
    """The RCCL form of the gather as bench.py drives it at N > 1 — compact records staged on the library's own stream
    (dg_index_stream as a torch ExternalStream), the asynchronous size agreement, the gather one step behind — with a process group
    of one rank on the `nccl` backend (two ranks cannot share the one GPU of this box under RCCL)."""
    dump = str(tmp_path / "gather1")
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gather-single", "--backend", "nccl", "--genome-size", "2e6", "--queries", "2000",
           "--steps", "4", "--warmup", "1", "--no-cpu-baseline", "--no-extras", "--no-extra-configs", "--batches", "3", "--dump-gather", dump,
           "--detail-out", str(tmp_path / "detail.json")]
    p = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    out = json.loads([ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["gathered_bytes_per_step"] > 0
    local = open(os.path.join(dump, "local_0.bin"), "rb").read()
    got = open(os.path.join(dump, "gathered_0.bin"), "rb").read()
    assert local and got == local
    # what travelled is the payload's 64 KiB size class, not a capacity
    assert out["gather_bytes_moved_per_step"] <= out["gathered_bytes_per_step"] + 65536 + 8


def test_cli_shards_a_batch_over_dicey_devices(tmp_path):
    """`dicey hunt` with DICEY_DEVICES=0,0,0: three host threads, three index replicas, contiguous query shards
    (SURVEY.md 8(e)); stdout and the gz outfile must be byte-identical to the single-device run, capped queries included"""
    import gzip
    import random
    from conftest import genome_text, make_genome, make_queries
    dicey = os.path.join(ROOT, "dicey_amd", "dicey")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dicey_amd", "cli"), "-s"])
    seqs = make_genome(77, 3, 15000)
    fa = tmp_path / "g.fa.gz"
    with gzip.open(fa, "wt") as f:
        for i, s in enumerate(seqs):
            f.write(">c%d\n%s\n" % (i, s))
    assert subprocess.run([dicey, "index", str(fa)], capture_output=True).returncode == 0
    qs = make_queries(3, genome_text(seqs), 101, lens=(12, 20, 26)) + ["ACGTAC", "N" * 15]
    random.Random(1).shuffle(qs)
    qf = tmp_path / "q.fa"
    qf.write_text("".join(">q%d\n%s\n" % (i, q) for i, q in enumerate(qs)))
    for extra in ([], ["-d", "2", "-m", "5"]):
        one = subprocess.run([dicey, "hunt", *extra, "-g", str(fa), str(qf)], capture_output=True, text=True)
        assert one.returncode == 0, one.stderr
        env = dict(os.environ, DICEY_DEVICES="0,0,0")
        many = subprocess.run([dicey, "hunt", *extra, "-g", str(fa), str(qf)], capture_output=True, text=True, env=env)
        assert many.returncode == 0, many.stderr
        assert many.stdout == one.stdout and many.stdout.count("\n") == len(qs)
    out = tmp_path / "o.json.gz"
    r = subprocess.run([dicey, "hunt", "-o", str(out), "-g", str(fa), str(qf)], capture_output=True, text=True,
                       env=dict(os.environ, DICEY_DEVICES="0,0"))
    assert r.returncode == 0 and r.stdout == ""
    one = subprocess.run([dicey, "hunt", "-g", str(fa), str(qf)], capture_output=True, text=True)
    assert gzip.open(out, "rt").read().replace(str(out), "") == one.stdout  # only the outfile field differs


def test_pipelined_gather_on_the_device_backend_single_rank():
    """The RCCL form of PipelinedGather (staging buffer in HBM, length prefix from pinned host memory with a stream-ordered
    copy) — the multi-rank tests above run on gloo, which takes the CPU branch.  One rank, backend nccl, on the test box's GPU:
    payloads of changing length over more steps than the pipeline is deep; what rank 0 holds after finish() is the last payload
    and the byte count is the sum."""
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
from dicey_amd.shard import PipelinedGather
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29631")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
dev = torch.device("cuda:0")
g = PipelinedGather(5000, dev, depth=2)
tot = 0
for step in range(7):
    n = 100 + 613 * step
    a = (torch.arange(n, device=dev) %% 251).to(torch.uint8)
    b = torch.full((step + 1,), step, dtype=torch.uint8, device=dev)
    g.submit([a, b]); tot += n + step + 1
    torch.cuda.current_stream().synchronize()
got = g.finish()
last = g.last_received()
want = bytes((i %% 251) for i in range(100 + 613 * 6)) + bytes([6] * 7)
assert got == tot, (got, tot)
assert len(last) == 1 and last[0] == want, (len(last[0]), len(want))
dist.destroy_process_group()
print("ok")
''' % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-500:], r.stderr[-1500:])


def test_cli_one_process_per_gpu_gathers_over_libdiceygather(tmp_path):
    """`dicey hunt` with DICEY_RANKS / DICEY_RANK: one process per GPU, contiguous query shards, every rank's compact blocks gathered
    to rank 0 by libdiceygather.so, which writes all lines in query order (BASELINE.json north_star).  On the one
    GPU of this box: a single rank over RCCL (communicator, size exchange, staging on the batch's stream — everything but a peer),
    and two / three ranks over the library's TCP test transport (RCCL takes one rank per device).  stdout of rank 0 must equal the
    plain single-process run byte for byte; the other ranks print nothing."""
    import gzip
    import random
    import socket
    from conftest import genome_text, make_genome, make_queries
    dicey = os.path.join(ROOT, "dicey_amd", "dicey")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dicey_amd", "cli"), "-s"])
    seqs = make_genome(78, 3, 15000)
    fa = tmp_path / "g.fa.gz"
    with gzip.open(fa, "wt") as f:
        for i, s in enumerate(seqs):
            f.write(">c%d\n%s\n" % (i, s))
    assert subprocess.run([dicey, "index", str(fa)], capture_output=True).returncode == 0
    qs = make_queries(5, genome_text(seqs), 203, lens=(12, 20, 26)) + ["ACGTAC", "N" * 15]
    random.Random(2).shuffle(qs)
    qf = tmp_path / "q.fa"
    qf.write_text("".join(">q%d\n%s\n" % (i, q) for i, q in enumerate(qs)))
    two = tmp_path / "two.fa"
    two.write_text(">a\n%s\n>b\n%s\n" % (qs[0], qs[1]))
    env0 = {k: v for k, v in os.environ.items() if not k.startswith("DICEY_")}
    for extra, inp in (([], qf), (["-d", "2", "-m", "5"], qf), ([], two)):
        one = subprocess.run([dicey, "hunt", *extra, "-g", str(fa), str(inp)], capture_output=True, text=True, env=env0)
        assert one.returncode == 0, one.stderr
        r1 = subprocess.run([dicey, "hunt", *extra, "-g", str(fa), str(inp)], capture_output=True, text=True,
                            env=dict(env0, DICEY_RANKS="1", DICEY_RANK="0"))
        assert r1.returncode == 0, r1.stderr
        assert r1.stdout == one.stdout
        for nr in (2, 3):
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                port = sk.getsockname()[1]
            ps = [subprocess.Popen([dicey, "hunt", *extra, "-g", str(fa), str(inp)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                                   env=dict(env0, DICEY_RANKS=str(nr), DICEY_RANK=str(r), DICEY_DEVICE="0", DICEY_COMM_TCP=str(port)))
                  for r in range(nr)]
            outs = [p.communicate(timeout=600) for p in ps]
            assert all(p.returncode == 0 for p in ps), [o[1][-500:] for o in outs]
            assert outs[0][0] == one.stdout, (nr, extra)
            assert all(o[0] == "" for o in outs[1:])


def test_cli_ranks_answer_refused_chunks_in_pieces_and_fail_together(tmp_path):
    """in DICEY_RANKS mode a chunk the library refuses with DG_ELIMIT used to end the rank (and leave its
    peers blocked in the next collective).  Now the ranks agree — through the collective that carries the byte counts — to cut
    the chunk into pieces, like run_slice's halves; any other failure travels the same way and every rank exits.  Two ranks over
    the TCP test transport, cap-prone queries (-x 50) under a 1 MB budget: stdout equals the single-process run."""
    import gzip
    import socket
    from conftest import genome_text, make_genome, make_queries
    dicey = os.path.join(ROOT, "dicey_amd", "dicey")
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dicey_amd", "cli"), "-s"])
    seqs = make_genome(79, 2, 20000)
    fa = tmp_path / "g.fa.gz"
    with gzip.open(fa, "wt") as f:
        for i, s in enumerate(seqs):
            f.write(">c%d\n%s\n" % (i, s))
    assert subprocess.run([dicey, "index", str(fa)], capture_output=True).returncode == 0
    base = make_queries(23, genome_text(seqs), 1500, (20,))
    qs = (base * 20)[:30000]
    qf = tmp_path / "capped.fa"
    qf.write_text("".join(">c%d\n%s\n" % (i, q) for i, q in enumerate(qs)))
    env0 = {k: v for k, v in os.environ.items() if not k.startswith("DICEY_")}
    env0.update(DICEY_KMER_K="9", DICEY_CAP_BUDGET_MB="1")
    cmd = [dicey, "hunt", "-x", "50", "-g", str(fa), str(qf)]
    one = subprocess.run(cmd, capture_output=True, text=True, env=env0)
    assert one.returncode == 0, one.stderr[-1500:]
    assert one.stdout.count("\n") == len(qs)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    ps = [subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(env0, DICEY_RANKS="2", DICEY_RANK=str(r), DICEY_DEVICE="0", DICEY_COMM_TCP=str(port)))
          for r in range(2)]
    outs = [p.communicate(timeout=900) for p in ps]
    assert all(p.returncode == 0 for p in ps), [o[1][-800:] for o in outs]
    assert outs[0][0] == one.stdout
    assert outs[1][0] == ""
    # a failure that is not a size refusal (a query the library rejects: distance above its limit) ends BOTH ranks with code 2
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    few = tmp_path / "few.fa"
    few.write_text("".join(">f%d\n%s\n" % (i, q) for i, q in enumerate(qs[:5])))
    bad = [dicey, "hunt", "-d", "9", "-g", str(fa), str(few)]
    ps = [subprocess.Popen(bad, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True,
                           env=dict(env0, DICEY_RANKS="2", DICEY_RANK=str(r), DICEY_DEVICE="0", DICEY_COMM_TCP=str(port)))
          for r in range(2)]
    outs = [p.communicate(timeout=300) for p in ps]
    assert [p.returncode for p in ps] == [2, 2], [o[1][-500:] for o in outs]
    assert outs[0][0] == "" and outs[1][0] == ""
