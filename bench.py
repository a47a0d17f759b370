#!/usr/bin/env python3
"""bench.py — primers/s of the `dicey hunt` hot path on MI355X.

Workload (BASELINE.json configs[1]): 100 000 synthetic 20-mers, edit distance 1, both strands, -m 1000 -x 10000,
against a GRCh38-size genome.  No genome can be downloaded here, so the genome is the deterministic synthetic one of
SURVEY.md §8(d): 24 sequences, 3.1 Gb in total, i.i.d. A/C/G/T at GRCh38 base frequencies, 5 % N in runs
(config.genome says so).  The FM-index is built on the GPU (dg_index_build_device), written in sdsl csa_wt<> layout,
and loaded back UNCHANGED through dg_index_open — the same path a `dicey index` file takes.

A step = one pass of the whole hunt pipeline (prepare, search, select, locate, verify) over the rank's 100 000
queries, which are resident in HBM when the timed region starts; hit records stay in HBM (N=1) or are gathered to
rank 0 over RCCL (N>1, inside the timed region).  One process per GPU; weak scaling (every rank searches its own
100 000 queries against its own index replica).

Prints ONE JSON line (rank 0).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
OCC_LINE_BYTES = 64    # one Occ block
BYTES_PER_EXT = 2 * OCC_LINE_BYTES  # an interval extension reads the block of each interval end
BYTES_PER_TAB_READ = 8  # one K-mer jump-table entry (lo, hi)
BYTES_PER_FILTER_PROBE = 4  # one word of the K-mer presence filter
GRCH38_FREQ = (0.295, 0.205, 0.205, 0.295)  # A C G T


def synth_genome(total_len: int, nchr: int, seed: int, device) -> (torch.Tensor, list):
    """Text SEQ1\\nSEQ2\\n...\\n on the device (uint8) and per-sequence lengths."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    # chromosome lengths: decreasing like a karyotype, summing to total_len
    w = np.linspace(2.0, 0.6, nchr)
    lens = np.maximum(1000, (w / w.sum() * total_len).astype(np.int64))
    text = torch.empty(int(lens.sum()) + nchr, dtype=torch.uint8, device=device)
    lut = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    cut = torch.tensor([GRCH38_FREQ[0], GRCH38_FREQ[0] + GRCH38_FREQ[1], GRCH38_FREQ[0] + GRCH38_FREQ[1] + GRCH38_FREQ[2]],
                       device=device)
    rng = np.random.default_rng(seed)
    at = 0
    for c in range(nchr):
        L = int(lens[c])
        CH = 1 << 27
        for o in range(0, L, CH):
            k = min(CH, L - o)
            r = torch.rand(k, device=device, generator=g)
            text[at + o: at + o + k] = lut[torch.bucketize(r, cut)]
        # 5 % N: one long centromere-like run, a telomere run at each end, plus short gaps
        nn = int(0.05 * L)
        runs = [(0, min(10000, nn // 20)), (L - min(10000, nn // 20), min(10000, nn // 20))]
        big = int(nn * 0.7)
        runs.append((int(L * 0.4), big))
        rest = nn - big - 2 * min(10000, nn // 20)
        ngap = max(1, rest // 50000)
        for _ in range(ngap):
            ln = max(1, rest // ngap)
            runs.append((int(rng.integers(0, max(1, L - ln))), ln))
        for s, ln in runs:
            if ln > 0:
                text[at + s: at + min(L, s + ln)] = 78
        text[at + L] = 10
        at += L + 1
    return text, [int(x) for x in lens]


def synth_queries(text: torch.Tensor, nq: int, m: int, seed: int):
    """80 % sampled from non-N genome positions (half of them with one random edit:
    substitution / insertion / deletion equiprobable, random position), 20 % uniform random ACGT."""
    rng = np.random.default_rng(seed)
    n = text.numel()
    out = []
    n_genome = int(nq * 0.8)
    need = n_genome
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    while need > 0:
        k = int(need * 1.3) + 16
        pos = torch.from_numpy(rng.integers(0, n - m - 1, size=k)).to(text.device)
        win = text[pos[:, None] + torch.arange(m + 1, device=text.device)[None, :]].cpu().numpy()
        ok = np.all((win != 78) & (win != 10), axis=1)
        for wv in win[ok][:need]:
            out.append(wv)
        need = n_genome - len(out)
    qs = []
    for i, wv in enumerate(out):
        q = wv[:m].copy()
        if i % 2 == 0:  # one random edit
            k = int(rng.integers(0, m))
            r = rng.random()
            if r < 1 / 3:
                q[k] = acgt[rng.integers(0, 4)]
            elif r < 2 / 3:
                q = np.concatenate([q[:k], q[k + 1:], wv[m:m + 1]])  # delete, keep length m with the next genome base
            else:
                q = np.concatenate([q[:k], acgt[rng.integers(0, 4, 1)], q[k:]])[:m]  # insert, trim back to m
        qs.append(q)
    for _ in range(nq - n_genome):
        qs.append(acgt[rng.integers(0, 4, m)])
    order = rng.permutation(nq)
    return [bytes(qs[i].tobytes()) for i in order]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--genome-size", type=float, default=3.1e9, help="synthetic genome length (default GRCh38 size class)")
    ap.add_argument("--queries", type=int, default=100000, help="queries per GPU per step")
    ap.add_argument("--qlen", type=int, default=20)
    ap.add_argument("--distance", type=int, default=1)
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--fm9", default="", help="reuse an existing index file instead of building the synthetic one")
    ap.add_argument("--keep-index", action="store_true")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="optional extra measurement after the timed region (N=1 only): the same steps with this many batches in "
                         "flight, one host thread + one handle on the shared index each (dg_index_share).  Off by default so that "
                         "a kernel trace of the default run only holds launches that had the GPU to themselves")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--same-device", action="store_true",
                    help="dry run of the N>1 control flow on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--dump-gather", default="",
                    help="directory: every rank writes the hit-list bytes of its last step (local_<rank>.bin) and rank 0 what the "
                         "gather delivered for every rank (gathered_<rank>.bin); used by the tests")
    a = ap.parse_args()

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # started by hand: launch the ranks ourselves, exactly as the driver would (one process per GPU)
        import socket
        import subprocess
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
               "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        env = dict(os.environ)
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        raise SystemExit(subprocess.call(cmd, env=env))

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if a.gpus != world:
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}; using the launcher's world size", file=sys.stderr)
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if a.same_device:
        local = 0
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a HIP device (there is no CPU path)")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if a.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(a.backend, rank=rank, world_size=world)

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    import dicey_amd
    from dicey_amd import _capi
    from dicey_amd.shard import PipelinedGather, device_bytes
    L = _capi.load()

    # ---------------- genome + index (rank 0 builds, everyone loads the same file unchanged)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    fm9 = a.fm9 or os.path.join(shm, f"dicey_bench_{os.environ.get('MASTER_PORT', 'p')}_{int(a.genome_size)}.fm9")
    meta_path = fm9 + ".meta.json"
    t0 = time.time()
    info = {}
    if rank == 0 and not a.fm9:
        text, lens = synth_genome(int(a.genome_size), 24, seed=1, device=dev)
        torch.cuda.synchronize()
        info["t_genome_s"] = time.time() - t0
        t1 = time.time()
        _capi.check(L, L.dg_index_build_device(C.c_void_p(text.data_ptr()), text.numel(), local, fm9.encode()))
        info["t_build_s"] = time.time() - t1
        # queries for every rank come from the same genome; rank 0 draws them while it still holds the text
        allq = [synth_queries(text, a.queries, a.qlen, seed=42 + r) for r in range(world)]
        json.dump({"lens": lens, "queries": [[q.decode() for q in qs] for qs in allq]}, open(meta_path, "w"))
        del text
        torch.cuda.empty_cache()
    elif rank == 0:
        if not os.path.exists(meta_path):
            raise SystemExit(f"--fm9 needs {meta_path} (sequence lengths + queries)")
    barrier()
    meta = json.load(open(meta_path))
    seqlen = [x + 1 for x in meta["lens"]]  # util.h:201
    queries = [q.encode() for q in meta["queries"][rank if rank < len(meta["queries"]) else 0]]
    t2 = time.time()
    ix = dicey_amd.FmIndex(fm9, device=local)
    st = ix.stats()
    info["t_open_s"] = time.time() - t2

    # ---------------- inputs resident in HBM
    nq = len(queries)
    qbytes = b"".join(queries)
    off = np.zeros(nq + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(q) for q in queries])
    d_q = torch.frombuffer(bytearray(qbytes), dtype=torch.uint8).to(dev)
    d_off = torch.from_numpy(off.view(np.int64)).to(dev)
    sl = (C.c_uint32 * len(seqlen))(*seqlen)
    p = _capi.HuntParams(a.distance, 0, 0, 1000, 10000)

    pipe = {"g": None}

    def step(fetch=0, handle=None):
        rp = C.POINTER(_capi.HuntResult)()
        _capi.check(L, L.dg_hunt_device(handle or ix.handle, C.byref(p), sl, len(seqlen), C.c_void_p(d_q.data_ptr()),
                                        C.c_void_p(d_off.data_ptr()), nq, len(qbytes), fetch, C.byref(rp)))
        R = rp.contents
        res = {"nhits": R.nhits, "ext": R.ctr_ext_steps, "leaves": R.ctr_leaves, "sa": R.ctr_sa_reads, "win": R.ctr_win_bytes, "tab": R.ctr_tab_reads, "probe": R.ctr_filter_probes,
               "ms_total": R.ms_total, "ms_search": R.ms_search, "ms_select": R.ms_select, "ms_locate": R.ms_locate,
               "ms_verify": R.ms_verify}
        if world > 1:  # hit lists to rank 0 over RCCL/xGMI, overlapped with the next step (dicey_amd/shard.py)
            hb = device_bytes(R.d_hits, R.nhits * C.sizeof(_capi.Hit), dev)
            ra = device_bytes(R.d_refalign, R.nhits * R.aln_stride, dev)
            qa = device_bytes(R.d_queryalign, R.nhits * R.aln_stride, dev)
            parts = [hb, ra, qa]  # views of the library's buffers, copied straight into the gather's staging buffer
            if pipe["g"] is None:  # first (warm-up) step: agree on a capacity once
                nbytes = sum(int(t.numel()) for t in parts)
                pipe["g"] = PipelinedGather(int(nbytes * 1.25) + 4096, dev if a.backend == "nccl" else torch.device("cpu"))
            pipe["g"].submit(parts)
            torch.cuda.current_stream().synchronize()  # the library reuses its buffers in the next step, on its own stream
            if a.dump_gather:
                pipe["last_local"] = torch.cat([t.reshape(-1) for t in parts]).cpu().numpy().tobytes()
        L.dg_hunt_result_free(rp)
        return res

    for _ in range(max(a.warmup, 1 if world > 1 else 0)):
        step()
    if pipe["g"] is not None:
        pipe["g"].finish()
        pipe["g"].bytes_received = 0
    barrier()
    t_start = time.perf_counter()
    acc = []
    for _ in range(a.steps):
        acc.append(step())
    gathered = pipe["g"].finish() if pipe["g"] is not None else 0  # every gather completes inside the timed region
    barrier()
    elapsed = time.perf_counter() - t_start
    if a.dump_gather and pipe["g"] is not None:
        os.makedirs(a.dump_gather, exist_ok=True)
        open(os.path.join(a.dump_gather, f"local_{rank}.bin"), "wb").write(pipe.get("last_local", b""))
        if rank == 0:
            for r, payload in enumerate(pipe["g"].last_received()):
                open(os.path.join(a.dump_gather, f"gathered_{r}.bin"), "wb").write(payload)
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---------------- extra, outside the timed region: the same K steps with `--pipeline` batches in flight (one host thread
    # and one handle with its own stream + workspaces per batch, dg_index_share).  The tail kernels of one batch overlap with
    # the search kernel of the other; per-kernel durations are then no longer those of a kernel alone, which is why the
    # headline value and the roofline above come from the one-batch-at-a-time loop.
    pipelined = None
    shared = [ix]
    if world == 1 and a.pipeline > 1:
        import threading
        shared = [ix] + [ix.share() for _ in range(a.pipeline - 1)]
        for h in shared:
            step(handle=h.handle)
        lock, todo = threading.Lock(), [a.steps]

        def worker(h):
            while True:
                with lock:
                    if todo[0] == 0:
                        return
                    todo[0] -= 1
                step(handle=h.handle)
        torch.cuda.synchronize()
        tp = time.perf_counter()
        ths = [threading.Thread(target=worker, args=(h,)) for h in shared]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        torch.cuda.synchronize()
        dtp = time.perf_counter() - tp
        pipelined = {"batches_in_flight": a.pipeline, "value": nq * a.steps / dtp, "unit": "primers/s", "ms_per_step": dtp / a.steps * 1e3,
                     "note": "same K steps, issued from %d host threads on handles sharing one resident index" % a.pipeline}

    # ---------------- cpu baseline + parity spot check at full size (rank 0, N=1 only)
    cpu = None
    cpu_par = None
    parity = None
    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib as O  # the checker / CPU port; never on the measured GPU path
        t3 = time.time()
        orc = O.Index(fm9)
        info["t_oracle_load_s"] = time.time() - t3
        qstr = [q.decode() for q in queries]
        dt, _, _ = orc.hunt_timed(seqlen, qstr[:100], threads=1, distance=a.distance)
        per = max(dt / 100, 1e-6)
        ns = int(min(nq, max(100, a.cpu_seconds / per)))
        dt, octr, _ = orc.hunt_timed(seqlen, qstr[:ns], threads=1, distance=a.distance)
        cpu = {"value": ns / dt, "unit": "primers/s", "cores": 1, "kind": "port",
               "sample": f"first {ns} of the {nq} bench queries, oracle hunt_one (restated hunter.h:291-444) on 1 host thread, "
                         f"{dt:.1f} s, index load excluded", "host_cpus": os.cpu_count(),
               "oracle_ops": octr}
        # the same loop on all host cores over query shards (the reference itself has no threads)
        ncores = min(os.cpu_count() or 1, 64)
        if ncores > 1:
            nsp = int(min(nq, ns * ncores * 0.6))
            dtp, _, _ = orc.hunt_timed(seqlen, qstr[:nsp], threads=ncores, distance=a.distance)
            cpu_par = {"value": nsp / dtp, "unit": "primers/s", "cores": ncores, "kind": "port",
                       "sample": f"first {nsp} bench queries, {ncores} host threads over query shards, {dtp:.1f} s"}
        # parity at full genome size: GPU hits (push order) == oracle hits for a sample
        npar = min(300, nq)
        got = ix.hunt(qstr[:npar], seqlen, distance=a.distance)
        _, ohits = orc.hunt(seqlen, ["s%d" % i for i in range(len(seqlen))], qstr[:npar], distance=a.distance, want_hits=True)
        perq = {}
        for h in ohits:
            perq.setdefault(h[0], []).append(h[1:])
        mism = 0
        for qi, qr in enumerate(got.queries):
            g = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
            mism += g != perq.get(qi, [])
        parity = {"queries": npar, "mismatching": mism, "hits": len(ohits)}

    # ---------------- report
    if rank == 0:
        ms_step = elapsed / a.steps * 1e3
        ext = float(np.mean([r["ext"] for r in acc]))
        ms_search = float(np.mean([r["ms_search"] for r in acc]))
        tab = float(np.mean([r["tab"] for r in acc]))
        probe = float(np.mean([r["probe"] for r in acc]))
        alg_bytes = ext * BYTES_PER_EXT + tab * BYTES_PER_TAB_READ + probe * BYTES_PER_FILTER_PROBE
        # the same launch in SURVEY.md §8(d) units: a backward step on c = 2 L(c) rank ops of 24 B on the sdsl layout
        # (L = Huffman code length in the loaded wavelet tree), small reads by their payload
        cl = st["code_len"]
        avg_l = sum(f * cl.get(ord(ch), 0) for f, ch in zip(GRCH38_FREQ, "ACGT"))
        survey_bytes = ext * 2 * avg_l * 24 + tab * BYTES_PER_TAB_READ + probe * BYTES_PER_FILTER_PROBE
        achieved = alg_bytes / (ms_search * 1e-3) / 1e9 if ms_search > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_k_search.json")
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("workload") == f"{a.queries}x{a.qlen}mer_d{a.distance}_n{int(a.genome_size)}":
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        out = {
            "metric": "primers/sec on GRCh38 edit-dist 1 at 1/2/4/8 GPUs; HBM GB/s vs peak",
            "value": world * nq * a.steps / elapsed,
            "unit": "primers/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u32", "data": "synthetic",
            "config": {"workload": f"dicey hunt, {nq} synthetic {a.qlen}-mers per GPU, edit distance {a.distance}, both strands, "
                                   f"-m 1000 -x 10000 (BASELINE.json configs[1])",
                       "genome": f"synthetic GRCh38-size: 24 sequences, {st['n'] - 1} symbols, i.i.d. ACGT at GRCh38 base "
                                 f"frequencies, 5% N runs, seed 1 (no real genome is available offline)",
                       "index": "sdsl csa_wt<> .fm9 built by dg_index_build_device, loaded unchanged by dg_index_open",
                       "queries_per_gpu": nq, "sharding": f"query-sharded x{world}, full index replica per GPU"},
            "roofline": {"bound": "hbm", "kernel": f"k_search<true,{a.distance}>", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes, "ext_steps_per_launch": ext,
                         "bytes_per_ext_step": BYTES_PER_EXT, "table_reads_per_launch": tab,
                         "bytes_per_table_read": BYTES_PER_TAB_READ, "filter_probes_per_launch": probe,
                         "bytes_per_filter_probe": BYTES_PER_FILTER_PROBE, "kernel_ms": ms_search,
                         "survey_units": {"bytes_per_launch": survey_bytes, "avg_code_len": avg_l,
                                          "achieved": survey_bytes / (ms_search * 1e-3) / 1e9 if ms_search > 0 else 0.0,
                                          "frac": survey_bytes / (ms_search * 1e-3) / 1e9 / HBM_PEAK_GBS if ms_search > 0 else 0.0,
                                          "note": "24 B per rank op, 2 L(c) rank ops per backward step (SURVEY.md 8(d))"},
                         "index_accesses_per_query": (2 * ext + tab + probe) / nq,
                         "index_lines_per_s": (2 * ext + tab + probe) / (ms_search * 1e-3) if ms_search > 0 else 0.0,
                         "gather_ceiling_note": "random 64-B lines over >=16 GiB top out at 19 G lines/s = 1.2 TB/s on this chip "
                                                "(profiles/r01b_gather_bench.jsonl); this kernel is a gather, not a stream"},
            "cpu_baseline": cpu,
            "cpu_baseline_parallel": cpu_par,
            "pipelined": pipelined,
            "parity_sample": parity,
            "phases_ms": {k: float(np.mean([r[k] for r in acc])) for k in ("ms_total", "ms_search", "ms_select", "ms_locate", "ms_verify")},
            "hits_per_step": int(acc[-1]["nhits"]), "leaves_per_step": int(acc[-1]["leaves"]),
            "index": {"n": st["n"], "file_bytes": st["file_bytes"], "hbm_bytes": st["hbm_bytes"],
                      "load_s": st["load_seconds"], "derive_s": st["derive_seconds"]},
            "setup_s": info,
        }
        if world > 1:
            out["gathered_bytes_per_step"] = gathered / max(1, a.steps)
        print(json.dumps(out), flush=True)
    for h in shared[1:]:
        h.close()
    ix.close()
    barrier()
    if rank == 0 and not a.fm9 and not a.keep_index:
        for f in (fm9, meta_path):
            try:
                os.remove(f)
            except OSError:
                pass
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
