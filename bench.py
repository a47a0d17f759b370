#!/usr/bin/env python3
"""bench.py — throughput of the dicey search path on MI355X.

Default workload (BASELINE.json configs[1], `--config hunt_d1`): 100 000 synthetic 20-mers, edit distance 1, both
strands, -m 1000 -x 10000, against a GRCh38-size genome.  No genome can be downloaded here, so the genome is the
deterministic synthetic one of SURVEY.md §8(d): 24 sequences, 3.1 Gb in total, i.i.d. A/C/G/T at GRCh38 base
frequencies, 5 % N in runs (`--genome repeats` plants repeat families on top; config.genome says which).  The FM-index
is built on the GPU (dg_index_build_device), written in sdsl csa_wt<> layout, and loaded back UNCHANGED through
dg_index_open — the same path a `dicey index` file takes.

A step = one pass of the whole pipeline over the rank's batch, which is resident in HBM (hunt) or handed over as host
buffers (search, padlock: their entry points take host buffers) when the timed region starts; hit records stay in HBM
(N=1) or are gathered to rank 0 over RCCL (N>1, inside the timed region).  One process per GPU; weak scaling.
Hunt configurations keep THREE batches in flight per GPU (`--in-flight 3`: dg_hunt_device_submit / dg_hunt_wait on the handle's
lanes; step k is submitted, step k - 2 collected; exactly K batches start and end inside the timed region).  The roofline then
divides the dominant kernel's algorithmic bytes by its BUSY time per launch — the union of the timed launches' intervals on the
lanes' common timeline (HIP events, dg_hunt_result::t_search_*) over the number of launches; `roofline.launch_ms` is the plain
duration of a launch (longer: neighbouring launches overlap), `roofline.one_in_flight` the kernel alone.

Other configurations of BASELINE.json, same contract, one JSON line each:
  --config hunt_d2   configs[3]: 20-mers at edit distance 2 (100 000 per GPU and step)
  --config search    configs[2]: 10 000 primer pairs = 20 000 primers, binding sites incl. one thal() per located hit
  --config padlock   configs[4]: 1 000 genes, per-position arm / probe values (thal, exact and neighbourhood counts)

`--gpus N` without a launcher starts the N ranks itself (python -m torch.distributed.run, 127.0.0.1).
Rank 0 writes the full detail object to bench_detail.json (`--detail-out`) and prints ONE compact JSON line (< 4 KB: the
contract keys, roofline, cpu_baseline, parity sample, a summary per other configuration) as the last line of stdout.
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
METRIC = "primers/sec on GRCh38 edit-dist 1 at 1/2/4/8 GPUs; HBM GB/s vs peak"


def union_of_intervals(iv) -> float:
    """Total length covered by a list of (begin, end) intervals — the time a kernel RAN when its launches overlap (several batches in
    flight): the roofline divides by this per launch, not by the sum of the launches' durations."""
    total = 0.0
    cur_b = cur_e = None
    for b, e in sorted(iv):
        if cur_e is None or b > cur_e:
            if cur_e is not None:
                total += cur_e - cur_b
            cur_b, cur_e = b, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        total += cur_e - cur_b
    return total


def build_id() -> str:
    """Identity of the kernel sources this run was built from (sha256 over dicey_amd/csrc/*.hip|*.hpp and include/dicey_gpu.h, 16 hex
    digits).  Numbers that come from a committed profile round (PMC traffic, instruction counters) are carried into a bench line only
    when the profile was taken on the same sources (tools/summarize_profile.py stamps the files it writes); otherwise they are null."""
    import glob
    import hashlib
    h = hashlib.sha256()
    files = sorted(glob.glob(os.path.join(ROOT, "dicey_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "dicey_amd", "csrc", "*.hpp")))
    for f in files + [os.path.join(ROOT, "include", "dicey_gpu.h")]:
        h.update(os.path.basename(f).encode() + b"\0")
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def profile_of_this_build(name: str, **want):
    """profiles/<name> (JSON) if it was taken on the sources of this build AND its fields equal `want` (exact, no substring match);
    None otherwise."""
    try:
        j = json.load(open(os.path.join(ROOT, "profiles", name)))
    except Exception:
        return None
    if j.get("build_id") != build_id():
        return None
    for k, v in want.items():
        if j.get(k) != v:
            return None
    return j


def workload_tag(units: int, qlen: int, distance: int, hamming: bool, n_frac: float, genome_size: float, genome: str) -> str:
    """What a profile file must have been taken ON to be quoted by a bench line: batch size, query length, distance AND mode
    (Hamming, share of queries with an N), genome.  r05 keyed the profiles by distance and genome only and the Hamming-distance-2
    sub-line carried the edit-distance-2 profile's traffic."""
    return (f"{int(units)}x{int(qlen)}mer_d{int(distance)}" + ("h" if hamming else "") + (f"_n{round(100 * n_frac)}pct" if n_frac > 0 else "") +
            f"_n{int(genome_size)}_{genome}")


def traffic_file(distance: int, hamming: bool, n_frac: float, genome: str) -> str:
    """profiles/<this name>: PMC traffic of the dominant search kernel on exactly this workload (tools/summarize_profile.py)"""
    if distance == 1 and genome == "iid" and not hamming and not n_frac > 0:
        return "traffic_k_search.json"
    return f"traffic_k_search_d{int(distance)}" + ("h" if hamming else "") + (f"_n{round(100 * n_frac)}pct" if n_frac > 0 else "") + f"_{genome}.json"


def kernel_stats_file(tag: str) -> str:
    """profiles/<this name>: rocprofv3 --kernel-trace --stats rows (calls, average / min / max duration) of the bench command on the
    workload `tag`, stamped with the sources' build_id (tools/summarize_profile.py)"""
    return f"kernel_stats_{tag}.json"


def rocprof_average(roof: dict, tag: str) -> dict:
    """Adds the rocprofv3 AVERAGE duration of the roofline's kernel (from a profile of THIS build on THIS workload, else null) and the
    fraction that follows from it, next to the busy-time figure: with several batches in flight a launch lasts longer than the time
    the chip spends on it (launches overlap), and the two divisors are quoted side by side."""
    roof = dict(roof)
    roof["kernel_avg_us_rocprof"] = None
    roof["frac_rocprof_avg"] = None
    ks = profile_of_this_build(kernel_stats_file(tag), workload=tag)
    row = ((ks or {}).get("kernels") or {}).get((roof.get("kernel") or "").split(" (")[0])
    if row and row.get("avg_us"):
        roof["kernel_avg_us_rocprof"] = row["avg_us"]
        roof["kernel_calls_rocprof"] = row.get("calls")
        if roof.get("bound") == "hbm" and roof.get("algorithmic_bytes_per_launch") and roof.get("peak"):
            roof["frac_rocprof_avg"] = roof["algorithmic_bytes_per_launch"] / (row["avg_us"] * 1e-6) / 1e9 / roof["peak"]
    return roof


def cap_enum_roofline(strands: float, qlen: int, distance: int, cap_ms: float) -> dict:
    """k_cap_enum (the capped neighbourhoods of cap-prone primers): a chain of dependent atomics and probes in hash tables that live in
    L2 while a workgroup works on them — bound by LATENCY, not by HBM bytes.  The block says so (`bound: "latency"`, frac null); the
    byte figure is an upper bound of the hash-table bytes and is kept for scale only."""
    leaves = 32 * qlen * qlen + 8 * qlen + 1 if distance >= 2 else 8 * qlen + 1
    tcap = 256
    while tcap < leaves * 5 // 2:
        tcap *= 2
    per_strand = tcap * 16 + leaves * 16 + leaves * 14 * 16 + (leaves + 2) * 4  # table clear, births, <= 14 substring probes per string, events
    cb = strands * per_strand
    ach = cb / (cap_ms * 1e-3) / 1e9 if cap_ms > 0 else 0.0
    return {"bound": "latency", "kernel": "k_cap_enum", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None,
            "hbm_frac_upper_bound": ach / HBM_PEAK_GBS, "traffic": None, "algorithmic_bytes_per_launch": cb, "kernel_ms": cap_ms, "stage": "cap_stage",
            "note": "bound by the latency of dependent atomics and probes (hash tables of 1 MB + 128 KB per workgroup, L2 resident while it works on "
                    "them), NOT by HBM bytes: `achieved` is an upper bound of the hash-table bytes per (query, strand) — table clear + 16 B per leaf "
                    "of the reference's trie (births) + 14 substring probes of 16 B per string (deaths) + the event array — over the stage's "
                    "time, kept for scale; no fraction of the HBM peak is claimed"}


CONTRACT_LINE_LIMIT = 4096  # the driver keeps a bounded tail of stdout: r04's 25.6 KB line was not parsed


def _short(x, n=160):
    return x if not isinstance(x, str) or len(x) <= n else x[:n - 3] + "..."


def _pick(d, keys):
    return {k: _short(d[k]) for k in keys if isinstance(d, dict) and k in d} if isinstance(d, dict) else None


def _sig(d, digits=7):
    """floats of a (flat) block at `digits` significant digits: the contract line's sub-blocks, not its headline keys"""
    if not isinstance(d, dict):
        return d
    return {k: (float("%.*g" % (digits, v)) if isinstance(v, float) and v == v and abs(v) != float("inf") else v) for k, v in d.items()}


def contract_line(out: dict, detail_path: str = "") -> str:
    """The ONE line the driver parses: the contract's keys, roofline + cpu_baseline of the dominant kernel, the parity sample and a
    five-field summary of every other configuration.  Everything else lives in the detail file.  Strict JSON (no NaN), < 4 KB."""
    top = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data")
    line = {k: out.get(k) for k in top}
    cfg = out.get("config") or {}
    line["config"] = {"workload": _short(cfg.get("workload"), 200), "genome": _short(cfg.get("genome_short") or cfg.get("genome"), 120)}
    for k in ("queries_per_gpu", "primers_per_gpu", "genes_per_gpu", "in_flight_batches", "distinct_batches", "sharding"):
        if k in cfg:
            line["config"][k] = _short(cfg[k], 80)
    rk = ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "kernel_ms", "kernel_ms_is", "launch_ms",
          "kernel_avg_us_rocprof", "frac_rocprof_avg", "traffic_over_algorithmic", "lines_per_strand", "stage")
    line["roofline"] = _pick(out.get("roofline"), rk)
    if isinstance(line["roofline"], dict) and isinstance(line["roofline"].get("kernel_ms_is"), str):
        line["roofline"]["kernel_ms_is"] = "busy time per launch (union of overlapping launches)" if line["roofline"]["kernel_ms_is"].startswith("busy") \
            else "average launch duration (HIP events)"
    line["cpu_baseline"] = _pick(out.get("cpu_baseline"), ("value", "unit", "cores", "kind", "cpu_model", "sample"))
    # the same loop on every physical core of the box, beside the single-thread figure the reference corresponds to
    line["cpu_baseline_parallel"] = _pick(out.get("cpu_baseline_parallel"), ("value", "unit", "cores", "kind"))
    # the witness of the headline: the same pipeline over at least a second (20 steps x 0.19 ms is a 3.7 ms measurement)
    line["sustained"] = _pick(out.get("sustained"), ("value", "unit", "seconds", "steps", "ms_per_step"))
    line["parity_sample"] = _pick(out.get("parity_sample"), ("queries", "mismatching", "hits"))
    for k in ("value_one_in_flight", "value_with_d2h", "host_to_host_pipelined", "cli_end_to_end_10M", "cli_end_to_end_d2_1M"):
        if isinstance(out.get(k), dict) and "value" in out[k]:
            line.setdefault("delivery", {})[k] = float("%.6g" % out[k]["value"])
    for k in sorted(out):
        if k.startswith("summary_"):
            line[k] = out[k]
    for k in ("gathered_bytes_per_step", "gather_bytes_moved_per_step"):
        if out.get(k) is not None:
            line[k] = out[k]
    line["build_id"] = out.get("build_id")
    line["detail"] = detail_path
    for k in ("roofline", "cpu_baseline", "cpu_baseline_parallel", "sustained"):
        line[k] = _sig(line.get(k))
    rf = line.get("roofline")
    if isinstance(rf, dict) and isinstance(rf.get("achieved"), float) and rf.get("peak") and rf.get("frac") is not None:
        rf["frac"] = rf["achieved"] / rf["peak"]  # (of the rounded figure: frac == achieved / peak holds to the last digit)
    s = json.dumps(line, allow_nan=False)
    for drop in ("delivery", "detail", "build_id", "cpu_baseline_parallel"):  # never over the limit: optional blocks go first
        if len(s) < CONTRACT_LINE_LIMIT:
            break
        line.pop(drop, None)
        s = json.dumps(line, allow_nan=False)
    if len(s) >= CONTRACT_LINE_LIMIT:
        for k in [k for k in line if k.startswith("summary_")]:
            line[k] = _pick(line[k], ("value", "unit", "frac"))
        s = json.dumps(line, allow_nan=False)
    assert len(s) < CONTRACT_LINE_LIMIT, len(s)
    return s


def _finite(x):
    """NaN / Infinity are not JSON: None in their place (json.dumps(allow_nan=False) then holds for the whole object)"""
    if isinstance(x, float):
        return x if x == x and x not in (float("inf"), float("-inf")) else None
    if isinstance(x, dict):
        return {k: _finite(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [_finite(v) for v in x]
    if isinstance(x, (np.floating, np.integer)):
        return _finite(x.item())
    return x


def emit(out: dict, detail_out: str = "") -> str:
    """Full detail -> a side file (bench_detail.json next to this script, also under gpurun_out/ when that exists; --detail-out
    overrides), then the compact contract line as the LAST line of stdout."""
    out = _finite(out)
    paths = [detail_out] if detail_out else [os.path.join(ROOT, "bench_detail.json")] + \
        ([os.path.join(ROOT, "gpurun_out", "bench_detail.json")] if os.path.isdir(os.path.join(ROOT, "gpurun_out")) else [])
    written = ""
    for p_ in paths:
        try:
            with open(p_, "w") as f:
                json.dump(out, f, allow_nan=False)
                f.write("\n")
            written = written or os.path.relpath(p_, ROOT)
        except OSError:
            pass
    s = contract_line(out, written)
    sys.stdout.flush()
    print(s, flush=True)
    return s


def synth_genome(total_len: int, nchr: int, seed: int, device, repeats: bool = False) -> (torch.Tensor, list):
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
    nruns = []
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
        nruns.append((at, L, runs))
        text[at + L] = 10
        at += L + 1
    if repeats:
        plant_repeats(text, lens, seed, device, g)
    for at, L, runs in nruns:  # the N runs go on top of everything else
        for s, ln in runs:
            if ln > 0:
                text[at + s: at + min(L, s + ln)] = 78
    return text, [int(x) for x in lens]


def plant_repeats(text, lens, seed, device, g):
    """Interspersed repeat families with copy numbers and divergences in the range of a human genome (about 45 % of GRCh38
    is repeats): an Alu-like 300-mer in ~1.1 M copies, a MIR-like 260-mer, L1-like 6 kb elements (mostly 5'-truncated),
    a few dozen smaller families, microsatellites, and segmental duplications (long near-identical copies).  Every copy
    carries its own substitutions (2-30 %); copy numbers scale with the genome length."""
    n = text.numel()
    scale = n / 3.1e9
    rng = np.random.default_rng(seed + 1000)
    acgt = torch.tensor([65, 67, 71, 84], dtype=torch.uint8, device=device)
    lens_np = np.asarray(lens, dtype=np.int64)
    starts = np.concatenate([[0], np.cumsum(lens_np + 1)[:-1]])

    def place(cons, copies, div_lo, div_hi, truncate=False):
        L = int(cons.numel())
        CH = max(1, (1 << 26) // L)
        for o in range(0, copies, CH):
            k = min(CH, copies - o)
            c = rng.integers(0, len(lens), size=k)
            room = lens_np[c] - L - 1
            ok = room > 0
            c, room = c[ok], room[ok]
            k = len(c)
            if not k:
                continue
            pos = torch.from_numpy(starts[c] + (rng.random(k) * room).astype(np.int64)).to(device)
            body = cons[None, :].repeat(k, 1)
            div = torch.from_numpy(rng.uniform(div_lo, div_hi, size=k).astype(np.float32)).to(device)
            mut = torch.rand((k, L), device=device, generator=g) < div[:, None]
            rnd = acgt[torch.randint(0, 4, (k, L), device=device, generator=g)]
            body = torch.where(mut, rnd, body)
            idx = pos[:, None] + torch.arange(L, device=device)[None, :]
            if truncate:  # 5'-truncated copies: only the last `keep` bases are inserted
                keep = torch.from_numpy(np.minimum(L, rng.geometric(1.0 / 900, size=k) + 100)).to(device)
                live = torch.arange(L, device=device)[None, :] >= (L - keep)[:, None]
                text[idx[live]] = body[live]
            else:
                text[idx.reshape(-1)] = body.reshape(-1)

    def consensus(L):
        return acgt[torch.randint(0, 4, (L,), device=device, generator=g)]

    place(consensus(300), int(1.1e6 * scale), 0.02, 0.18)      # Alu-like
    place(consensus(260), int(0.5e6 * scale), 0.10, 0.30)      # MIR-like, older
    place(consensus(6000), int(0.5e6 * scale), 0.02, 0.20, truncate=True)  # L1-like
    for _ in range(40):                                         # smaller families (DNA transposons, LTRs)
        place(consensus(int(rng.integers(150, 1500))), int(rng.integers(2000, 40000) * scale), 0.03, 0.25)
    for _ in range(max(1, int(120 * scale))):                   # microsatellites: 5000 loci per motif
        unit = consensus(int(rng.integers(1, 7)))
        place(unit.repeat(int(rng.integers(5, 40))), max(1, int(5000 * min(1.0, scale * 10))), 0.0, 0.05)
    for _ in range(int(300 * scale) + 1):                       # segmental duplications: long copies, 0.5-5 % diverged
        L = int(rng.integers(10000, 200000))
        if n <= L + 2:
            continue
        src = int(rng.integers(0, n - L - 1))
        place(text[src:src + L].clone(), int(rng.integers(1, 4)), 0.005, 0.05)
    for s, l in zip(starts, lens_np):  # separators survive
        text[int(s) + int(l)] = 10
    bad = (text != 65) & (text != 67) & (text != 71) & (text != 84) & (text != 10)
    if bool(bad.any()):  # a duplicated segment may have carried a separator along: back to a base
        text[bad] = 65


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


def synth_query_batch(text: torch.Tensor, nq: int, m: int, seed: int) -> np.ndarray:
    """The same mixture as synth_queries (80 % genome windows, half of them with one random substitution / deletion / insertion,
    20 % uniform random), drawn with array operations: one call per batch of the rotating stream (r04).  Returns uint8 [nq, m]."""
    rng = np.random.default_rng(seed)
    n = text.numel()
    n_genome = int(nq * 0.8)
    rows = []
    need = n_genome
    while need > 0:
        k = int(need * 1.3) + 16
        pos = torch.from_numpy(rng.integers(0, n - m - 1, size=k)).to(text.device)
        win = text[pos[:, None] + torch.arange(m + 1, device=text.device)[None, :]].cpu().numpy()
        ok = np.all((win != 78) & (win != 10), axis=1)
        rows.append(win[ok][:need])
        need -= len(rows[-1])
    win = np.concatenate(rows, axis=0)
    acgt = np.frombuffer(b"ACGT", dtype=np.uint8)
    col = np.arange(m)[None, :]
    k = rng.integers(0, m, size=n_genome)[:, None]
    kind = rng.integers(0, 3, size=n_genome)[:, None]
    kind = np.where((np.arange(n_genome) % 2 == 0)[:, None], kind, 3)      # every second window stays as it is
    base = acgt[rng.integers(0, 4, size=n_genome)][:, None]
    src = np.where(kind == 1, col + (col >= k), np.where(kind == 2, col - (col > k), col))  # deletion pulls the next genome base in
    q = np.take_along_axis(win, src, axis=1)
    q = np.where(((kind == 0) | (kind == 2)) & (col == k), base, q)
    rnd = acgt[rng.integers(0, 4, size=(nq - n_genome, m))]
    allq = np.concatenate([q, rnd], axis=0)
    return np.ascontiguousarray(allq[rng.permutation(nq)])


COMP = bytes.maketrans(b"ACGT", b"TGCA")


def synth_primer_pairs(text: torch.Tensor, npairs: int, seed: int):
    """primer lengths uniform 18-25, forward primer sampled from the genome, reverse primer = reverse
    complement of a site 80-2000 bp downstream."""
    rng = np.random.default_rng(seed)
    n = text.numel()
    prim = []
    while len(prim) < 2 * npairs:
        k = 4096
        pos = rng.integers(0, n - 3000, size=k)
        win = text[torch.from_numpy(pos).to(text.device)[:, None] + torch.arange(2100, device=text.device)[None, :]].cpu().numpy()
        for j in range(k):
            l1, l2, d = int(rng.integers(18, 26)), int(rng.integers(18, 26)), int(rng.integers(80, 2000))
            w = win[j, :d + l2 + 1].tobytes()
            fw, rv = w[:l1], w[d:d + l2]
            if b"N" in fw + rv or b"\n" in w:
                continue
            prim += [fw.decode(), rv.translate(COMP)[::-1].decode()]
            if len(prim) >= 2 * npairs:
                break
    return prim


def synth_exons(text: torch.Tensor, ngenes: int, seed: int):
    """configs[4] shape: a gene = 4-12 exons of 100-600 nt inside a 100 kb window, one strand."""
    rng = np.random.default_rng(seed)
    n = text.numel()
    exons, gene_of = [], []
    for gid in range(ngenes):
        base = int(rng.integers(0, max(1, n - 200000)))
        strand = int(rng.integers(0, 2))
        for _ in range(int(rng.integers(4, 13))):
            p = base + int(rng.integers(0, 100000))
            ln = int(rng.integers(100, 601))
            if p + ln >= n:
                continue
            s = bytes(text[p:p + ln].cpu().numpy().tobytes())
            if b"\n" in s:
                continue
            exons.append((s.translate(COMP)[::-1] if strand else s).decode("latin-1"))
            gene_of.append(gid)
    return exons, gene_of


def host_cpu_info():
    """physical cores (distinct (physical id, core id) pairs) and the model string of /proc/cpuinfo"""
    model, cores, phys, core = "", set(), None, None
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and not model:
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                phys = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
            elif not ln.strip():
                if phys is not None and core is not None:
                    cores.add((phys, core))
                phys = core = None
    except OSError:
        pass
    return model, (len(cores) or (os.cpu_count() or 1))


def thal_counter_block(cfg, kernel, thal_calls, stage_ms):
    """k_site_wave against the fp64 vector peak and the LDS: instruction counters of the committed PMC pass
    on this workload (tools/prof_thal.sh -> profiles/thal_counters.json), scaled to this run's thal() count and stage time.
    fp64 FLOP/s is an UPPER bound (every lane of every fp64 wavefront instruction counted as active; an FMA as two)."""
    tc = profile_of_this_build("thal_counters.json")  # tools/prof_thal.sh + tools/summarize_thal.py, stamped with the sources' build_id
    k = ref_calls = None
    src = "thal_counters.json"
    try:
        tc = tc[cfg]
        ks = [v for n, v in tc["kernels"].items() if kernel in n]  # (every instantiation of the kernel: k_thal_self_wave<true> / <false>)
        k = {c: sum(v.get(c, 0.0) for v in ks) for c in ks[0]}
        ref_calls = (tc["bench"].get("site_stage") or {}).get("thal_calls_per_step") or \
            (tc["bench"]["arm_thal_per_step"] + tc["bench"]["probe_thal_per_step"])
    except Exception:
        k = None
    if k is None:
        return None
    if not thal_calls or not stage_ms:
        return None
    scale = thal_calls / ref_calls
    f64 = (k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_TRANS_F64"]) * scale
    flops = (k["SQ_INSTS_VALU_ADD_F64"] + k["SQ_INSTS_VALU_MUL_F64"] + 2 * k["SQ_INSTS_VALU_FMA_F64"] + k["SQ_INSTS_VALU_TRANS_F64"]) * 64 * scale
    t = stage_ms * 1e-3
    valu_peak = 256 * 4 * 2.4e9 / 4  # wavefront instructions per second: 1 024 SIMDs, one wave64 instruction per four cycles
    return {"source": "profiles/%s (rocprofv3 --pmc on this workload and on this build's sources), scaled by thal() calls" % src,
            "valu_wave_instructions_per_s": k["SQ_INSTS_VALU"] * scale / t, "valu_issue_peak_per_s": valu_peak,
            "fp64_wave_instructions_per_thal": f64 / thal_calls, "valu_wave_instructions_per_thal": k["SQ_INSTS_VALU"] * scale / thal_calls,
            "lds_wave_instructions_per_thal": k["SQ_INSTS_LDS"] * scale / thal_calls,
            "fp64_TFLOPs_upper_bound": flops / t / 1e12, "fp64_vector_peak_TFLOPs": 78.6, "fp64_frac_upper_bound": flops / t / 78.6e12,
            "valu_issue_frac": k["SQ_INSTS_VALU"] * scale / t / valu_peak,
            "lds_bank_conflict_cycles_over_active": k["SQ_LDS_BANK_CONFLICT"] / k["SQ_LDS_IDX_ACTIVE"],
            "note": "the kernel issues VALU instructions at about two thirds of the chip's rate, one in nine of them fp64: bound by "
                    "instruction issue of the loop-candidate evaluation (compares, selects, address arithmetic), not by the fp64 units"}


def padlock_replay(R, e, ln, armlen=20, mingc=0.4, maxgc=0.6, tmdiff=2, max_nb=2, exp=1):
    """padlock.h:321-428 + :506 on dg_padlock_scan's arrays (arm mode, edit distance 1, non-overlapping; spacer / barcode GC left out):
    the offsets inside exon e at which the reference accepts a probe"""
    o = int(R["pos_off"][e])
    acc = []
    k = 0
    while k < ln - 2 * armlen + 1:
        g1, g2, pg = R["arm_gc"][o + k], R["arm_gc"][o + k + armlen], R["probe_gc"][o + k]
        ok = mingc <= g1 <= maxgc and not (R["arm_tm"][o + k] > 93 + g1 - 675.0 / armlen) and mingc <= g2 <= maxgc
        ok = ok and not (R["arm_tm"][o + k + armlen] > 93 + g2 - 675.0 / armlen) and abs(R["arm_tm"][o + k] - R["arm_tm"][o + k + armlen]) <= tmdiff
        ok = ok and mingc <= pg <= maxgc
        if ok:
            lo = 81.5 + pg - 675.0 / (2 * armlen)
            ok = lo <= R["probe_tm"][o + k] <= lo + 10
        ok = ok and R["arm_count"][o + k] <= exp and R["arm_count"][o + k + armlen] <= exp
        ok = ok and R["arm_nbcount"][o + k] <= max_nb and R["arm_nbcount"][o + k + armlen] <= max_nb
        if ok:
            acc.append(k)
            k += 2 * armlen - 1
        k += 1
    return acc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=6)
    ap.add_argument("--config", default="hunt_d1", choices=["hunt_d1", "hunt_d2", "search", "padlock"],
                    help="which BASELINE.json configuration to run (default: configs[1], the one the metric is quoted on)")
    ap.add_argument("--genome", default="iid", choices=["iid", "repeats"],
                    help="synthetic genome: i.i.d. bases (SURVEY 8(d)), or the same with planted repeat families")
    ap.add_argument("--genome-size", type=float, default=3.1e9, help="synthetic genome length (default GRCh38 size class)")
    ap.add_argument("--queries", type=int, default=0,
                    help="units per GPU per step (default: 100000 queries / 20000 primers / 1000 genes, by --config)")
    ap.add_argument("--qlen", type=int, default=20)
    ap.add_argument("--distance", type=int, default=-1, help="override the configuration's distance")
    ap.add_argument("--hamming", action="store_true", help="hunt configs: -n (substitutions only, neighbors.h:57-66 without the indel branches)")
    ap.add_argument("--n-frac", type=float, default=0.0,
                    help="hunt configs: this fraction of the queries gets one 'N' at a random position (hunter.h:306-307, util.h:208-219: "
                         "such queries leave the flat kernels for the general k_search)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="target CPU time of the cpu_baseline sample")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the delivery measurements after the timed region (hunt configs, N=1)")
    ap.add_argument("--parity-queries", type=int, default=-1, help="size of the full-size parity sample (default 300; 1000 for hunt_d2)")
    ap.add_argument("--no-extra-configs", action="store_true",
                    help="default run only (hunt_d1, i.i.d. genome, N=1): skip the compact sub-lines of the other configurations "
                         "(extra_configs: hunt_d1_repeats, hunt_d2, hunt_d2_25mers, search, padlock), which are measured by re-running this script")
    ap.add_argument("--extra-budget-s", type=float, default=330.0, help="wall-clock budget of the extra_configs block")
    ap.add_argument("--big-table", action="store_true",
                    help="open the index with DG_OPEN_BIG_TABLE (K-mer table one order larger: 199 GB instead of 90 GB resident on the "
                         "GRCh38-size genome, search kernel ~5 %% faster); the default is the library's default layout")
    ap.add_argument("--fm9", default="", help="reuse an existing index file instead of building the synthetic one")
    ap.add_argument("--keep-index", action="store_true")
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1, hunt configs: weak = every rank searches its own --queries per step (the default, what the driver's scaling "
                         "run measures); strong = the N ranks share ONE batch of --queries by contiguous ranges (shard_range), e.g. "
                         "--config hunt_d2 --queries 10000000 --scaling strong for BASELINE configs[3]")
    ap.add_argument("--gather-single", action="store_true",
                    help="testing aid for a 1-GPU box: a process group of ONE rank on the chosen backend, and every hunt step stages and "
                         "gathers its hit list exactly as ranks of an N > 1 job do (the RCCL path: staging on the library's stream, the "
                         "size agreement, the gather) — n_gpus stays 1")
    ap.add_argument("--in-flight", type=int, default=3, choices=(1, 2, 3, 4, 5, 6, 7, 8),
                    help="hunt configs: batches in flight per GPU in the timed region: 2 or 3 = dg_hunt_device_submit / dg_hunt_wait on the handle's "
                         "lanes (step k is submitted, step k - n + 1 collected), 1 = dg_hunt_device, one batch at a time (r01-r04a); "
                         "more than 3 need a library built with -DDG_NEXTRA=n-1 (tools/r05_call13.sh: measured, no gain)")
    ap.add_argument("--stagger-us", type=float, default=0.0,
                    help="hunt configs with batches in flight: the first submissions of the timed region (and of the sustained pass) are this many "
                         "microseconds apart instead of back to back, so that the lanes do not start their search kernels together (measurement "
                         "of the lanes' lock-step, DESIGN; the waits lie INSIDE the timed region)")
    ap.add_argument("--batches", type=int, default=16,
                    help="hunt configs: distinct query batches resident in HBM that the warm-up and timed steps cycle through (step k "
                         "searches batch k mod B; hunter.h:291 searches every query once, so the headline never replays a batch within "
                         "B launches); 1 = the replayed-batch run of r01-r03, which the default run still reports as value_same_batch")
    ap.add_argument("--cli-queries", type=int, default=10000000,
                    help="hunt_d1 default run (N = 1): size of the large `dicey hunt` run timed after the bench released its index "
                         "(BASELINE configs[3]'s 10 M queries through the binary: FASTA in, one JSON line per query out); 0 = skip")
    ap.add_argument("--pipeline", type=int, default=1,
                    help="optional extra measurement after the timed region (hunt, N=1 only): the same steps with this many batches in "
                         "flight, one host thread + one handle on the shared index each (dg_index_share).  Off by default so that "
                         "a kernel trace of the default run only holds launches that had the GPU to themselves")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for dry runs)")
    ap.add_argument("--same-device", action="store_true",
                    help="dry run of the N>1 control flow on a 1-GPU box: every rank uses cuda:0 (needs --backend gloo)")
    ap.add_argument("--detail-out", default="",
                    help="where the full detail object goes (default: bench_detail.json next to this script, and gpurun_out/ when present); "
                         "stdout's last line is the compact contract line (< 4 KB) either way")
    ap.add_argument("--dump-gather", default="",
                    help="directory: every rank writes the hit-list bytes of its last step (local_<rank>.bin) and rank 0 what the "
                         "gather delivered for every rank (gathered_<rank>.bin); used by the tests")
    a = ap.parse_args()

    # BASELINE configs[3] at its size on ONE GPU: `--config hunt_d2 --queries 10000000 --scaling strong --gpus 1` is one
    # batch of --queries cut into chunks of 100 000 (the per-GPU batch of every other hunt line), a.in_flight of them in flight, every
    # chunk distinct and all of them started and drained inside the timed region: steps = chunks, each searching its own queries.
    a.total_queries = 0
    if a.scaling == "strong" and a.gpus == 1 and "WORLD_SIZE" not in os.environ and a.config in ("hunt_d1", "hunt_d2") and a.queries > 131072:
        chunk = 100000
        a.total_queries = (a.queries + chunk - 1) // chunk * chunk
        a.batches = a.total_queries // chunk
        a.steps = a.batches
        a.warmup = min(a.warmup, 6)
        a.queries = chunk

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
    if a.gather_single and world == 1:
        import socket
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ["MASTER_ADDR"] = "127.0.0.1"
    if world > 1 or a.gather_single:
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

    cfg = a.config
    distance = a.distance if a.distance >= 0 else {"hunt_d1": 1, "hunt_d2": 2, "search": 1, "padlock": 1}[cfg]
    units = a.queries or {"hunt_d1": 100000, "hunt_d2": 100000, "search": 20000, "padlock": 1000}[cfg]
    qseed = {"hunt_d1": 42, "hunt_d2": 44, "search": 43, "padlock": 45}[cfg]

    # ---------------- genome + index (rank 0 builds, everyone loads the same file unchanged)
    shm = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    fm9 = a.fm9 or os.path.join(shm, f"dicey_bench_{os.environ.get('MASTER_PORT', 'p')}_{a.genome}_{int(a.genome_size)}.fm9")
    meta_path = fm9 + f".{cfg}.{units}.meta.json"
    batches_path = fm9 + f".{cfg}.{units}.q{a.qlen}.b{a.batches}.npy"
    cli_big_path = fm9 + f".cli{a.cli_queries}.q{a.qlen}.npy"
    t0 = time.time()
    info = {}
    host_text = None
    have_meta = os.path.exists(meta_path) and (cfg not in ("hunt_d1", "hunt_d2") or a.batches <= 1 or os.path.exists(batches_path))
    if rank == 0 and not (a.fm9 and have_meta):
        text, lens = synth_genome(int(a.genome_size), 24, seed=1, device=dev, repeats=a.genome == "repeats")
        torch.cuda.synchronize()
        info["t_genome_s"] = time.time() - t0
        if not a.fm9:
            t1 = time.time()
            _capi.check(L, L.dg_index_build_device(C.c_void_p(text.data_ptr()), text.numel(), local, fm9.encode()))
            info["t_build_s"] = time.time() - t1
        # the inputs of every rank come from the same genome; rank 0 draws them while it still holds the text
        meta = {"lens": lens}
        if cfg in ("hunt_d1", "hunt_d2"):
            meta["queries"] = [[q.decode() for q in synth_queries(text, units, a.qlen, seed=qseed + r)] for r in range(world)]
            if os.environ.get("DICEY_BENCH_SORT"):  # experiment: queries in the order of their forward filter window (page locality)
                meta["queries"] = [sorted(qs, key=lambda q: q[1:]) for qs in meta["queries"]]
            # the other batches of the rotating stream (batch 0 is the list above, the r01-r03 batch): [world, B - 1, units, qlen]
            if a.batches > 1:
                more = np.stack([np.stack([synth_query_batch(text, units, a.qlen, seed=qseed + r + 1000 * b) for b in range(1, a.batches)])
                                 for r in range(world)])
                np.save(batches_path, more)
            # the large CLI run's queries (distinct draws, 100 000 at a time), while the text is here
            if cfg == "hunt_d1" and world == 1 and a.cli_queries > 0 and not a.no_extras and not a.no_extra_configs:
                big = np.concatenate([synth_query_batch(text, 100000, a.qlen, seed=5000 + i) for i in range((a.cli_queries + 99999) // 100000)])
                np.save(cli_big_path, big[:a.cli_queries])
        elif cfg == "search":
            meta["primers"] = [synth_primer_pairs(text, units // 2, seed=qseed + 100 * r) for r in range(world)]
        else:
            meta["exons"] = [synth_exons(text, units, seed=qseed + 100 * r) for r in range(world)]
        json.dump(meta, open(meta_path, "w"))
        if cfg == "search" and not a.no_cpu_baseline and world == 1:
            host_text = text.cpu().numpy().tobytes()  # the checker's amplicon sequences come from the text
        del text
        torch.cuda.empty_cache()
    elif rank == 0 and cfg == "search" and not a.no_cpu_baseline and world == 1:
        text, _ = synth_genome(int(a.genome_size), 24, seed=1, device=dev, repeats=a.genome == "repeats")
        host_text = text.cpu().numpy().tobytes()
        del text
        torch.cuda.empty_cache()
    barrier()
    meta = json.load(open(meta_path))
    seqlen = [x + 1 for x in meta["lens"]]  # util.h:201
    t2 = time.time()
    ix = dicey_amd.FmIndex(fm9, device=local, big_table=a.big_table)
    st = ix.stats()
    info["t_open_s"] = time.time() - t2
    sl = (C.c_uint32 * len(seqlen))(*seqlen)
    cpu_model, phys_cores = host_cpu_info()
    genome_desc = (f"synthetic GRCh38-size: 24 sequences, {st['n'] - 1} symbols, " +
                   ("i.i.d. ACGT at GRCh38 base frequencies" if a.genome == "iid" else
                    "i.i.d. background at GRCh38 base frequencies with planted repeat families (Alu-like 300-mer x 1.1 M copies, MIR-like, "
                    "L1-like 6 kb, 40 smaller families, microsatellites, segmental duplications; 0.5-30 % divergence per copy)") +
                   ", 5% N runs, seed 1 (no real genome is available offline)")
    genome_short = (f"synthetic GRCh38-size ({st['n'] - 1} symbols, 24 sequences), " +
                    ("i.i.d. ACGT at GRCh38 base frequencies" if a.genome == "iid" else "i.i.d. background + planted repeat families") + ", 5% N runs, seed 1")
    base_out = {"metric": METRIC, "unit": "primers/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "higher_is_better": True,
                "scaling": a.scaling, "vs_baseline": None, "data": "synthetic"}
    pipe = {"g": None}
    shared = [ix]
    th = None

    def timed(step, flush=None):
        """flush: steps that keep a batch in flight hand in what is outstanding (a list of step results); called behind the warm-up
        and inside the timed region, so that exactly K batches start and end there"""
        for _ in range(max(a.warmup, 1 if world > 1 else 0)):
            step()
        if flush:
            flush()
        if pipe["g"] is not None:
            pipe["g"].finish()
            pipe["g"].bytes_received = 0
            pipe["g"].bytes_moved = 0
        barrier()
        t_start = time.perf_counter()
        acc = [step() for _ in range(a.steps)]
        if flush:
            acc = [r for r in acc + flush() if r is not None]
        gathered = pipe["g"].finish() if pipe["g"] is not None else 0  # every gather completes inside the timed region
        pipe.setdefault("moved", pipe["g"].bytes_moved if pipe["g"] is not None else 0)  # the headline's (later passes gather too)
        barrier()
        elapsed = time.perf_counter() - t_start
        if world > 1:
            t = torch.tensor([elapsed], dtype=torch.float64, device=dev if a.backend == "nccl" else "cpu")
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            elapsed = float(t.item())
        return acc, elapsed, gathered

    class CxxGather:
        """The hunt configurations on the RCCL backend: libdiceygather.so (dicey_amd/csrc/gather.hip, include/dicey_gather.h) — the
        C++ host's gather.  A step's whole answer ([hit counts | query words | records], ONE block in HBM: dg_hunt_result::d_block) is
        staged on the stream its batch ran on and travels one step behind as exact-size ncclSend / ncclRecv on the communicator's own
        stream: no host synchronisation anywhere on the submit path (r04 synchronised torch's stream every step)."""

        def __init__(self, capacity):
            from dicey_amd import _gather
            uid = [_gather.Comm.unique_id() if rank == 0 else None]
            if world > 1:
                dist.broadcast_object_list(uid, src=0)
            self.c = _gather.Comm(world, rank, int(capacity), root=0, device=local, unique_id=uid[0])
            self.bytes_received = 0
            self.bytes_moved = 0

        def submit_block(self, ptr, nbytes, stream):
            self.c.submit(ptr, nbytes, stream)

        def finish(self):
            b, _ = self.c.finish()
            self.bytes_received += b
            self.bytes_moved += b  # exact-size transfers: what travels is the payload
            return self.bytes_received

        def last_received(self):
            return [self.c.last(r) for r in range(world)] if rank == 0 else []

    def gather_block(R):
        """hunt, backend nccl: the batch's compact block to rank 0 through the C++ gather"""
        nb = int(R.d_block_bytes) if R is not None else 0
        if pipe["g"] is None:
            # every rank opens the communicator here, in its first (warm-up) step.  A rank that cannot (library missing, RCCL refusing the
            # communicator) says so on stderr and the job falls back to the torch.distributed gather of r04 — agreed by all ranks, named
            # in config.gather: a scaling run that dies in its first step measures nothing
            err = ""
            try:
                g_ = CxxGather(int(max(nb, 8 * units) * 1.5) + 65536)
            except Exception as e:
                g_, err = None, repr(e)[:300]
            bad = torch.tensor([1 if g_ is None else 0], dtype=torch.int32, device=dev)
            if world > 1:
                dist.all_reduce(bad, op=dist.ReduceOp.MAX)
            if int(bad.item()):
                if err:
                    print(f"bench.py rank {rank}: libdiceygather could not be opened ({err}); gathering through torch.distributed", file=sys.stderr)
                if g_ is not None:
                    g_.c.close()
                pipe["cxx_off"] = err or "another rank could not open libdiceygather"
            else:
                pipe["g"] = g_
                pipe["cxx_on"] = True
        if pipe.get("cxx_off"):
            parts = [device_bytes(R.d_block, nb, dev)] if nb else [torch.empty(0, dtype=torch.uint8, device=dev)]
            return gather_parts(parts)
        pipe["g"].submit_block(R.d_block if nb else 0, nb, R.stream if nb else 0)
        if a.dump_gather:
            pipe["last_local"] = device_bytes(R.d_block, nb, dev).cpu().numpy().tobytes() if nb else b""

    def gather_parts(parts):
        """result lists to rank 0 over RCCL/xGMI, overlapped with the next step (dicey_amd/shard.py)"""
        if pipe["g"] is None:  # first (warm-up) step: agree on a capacity once
            nbytes = sum(int(t.numel()) for t in parts)
            pipe["g"] = PipelinedGather(int(nbytes * 1.5) + 65536,  # a bound, not what travels: the gather moves each step's agreed payload size
                                        dev if a.backend == "nccl" else torch.device("cpu"))
        # The staging copies read the library's result buffers, which its next batch overwrites: they must be done before the next
        # dg_hunt_device.  torch's stream is synchronised for that (the copies are the only work on it: a few microseconds).  Staging
        # on the library's own stream (dg_index_stream as a torch ExternalStream) would need no host synchronisation, but torch's wheel
        # brings its own HIP runtime and a stream handle of the system runtime behind libdiceygpu is not a stream to it (r04: the
        # process dies); a C++ host that links one runtime can do it.
        pipe["g"].submit(parts)
        torch.cuda.current_stream().synchronize()
        if a.dump_gather:
            pipe["last_local"] = torch.cat([t.reshape(-1) for t in parts]).cpu().numpy().tobytes()

    out = None
    gathered = 0
    cli_job = None
    # =====================================================================================================================
    if cfg in ("hunt_d1", "hunt_d2"):
        strong = a.scaling == "strong" and world > 1
        if strong:  # one batch for the whole job, cut into contiguous ranges in rank order (an empty range is a legal shard)
            from dicey_amd.shard import shard_range
            lo_q, hi_q = shard_range(len(meta["queries"][0]), rank, world)
            queries = [q.encode() for q in meta["queries"][0][lo_q:hi_q]]
        else:
            queries = [q.encode() for q in meta["queries"][rank if rank < len(meta["queries"]) else 0]]
        def with_n(arr, seed_):
            """one 'N' at a random position in a.n_frac of the rows of a uint8 [n, m] array (in place)"""
            if a.n_frac > 0 and arr.size:
                r_ = np.random.default_rng(seed_)
                rows = np.nonzero(r_.random(arr.shape[0]) < a.n_frac)[0]
                arr[rows, r_.integers(0, arr.shape[1], size=rows.size)] = ord("N")
            return arr
        if a.n_frac > 0 and queries:
            qa_ = with_n(np.frombuffer(b"".join(queries), dtype=np.uint8).reshape(len(queries), -1).copy(), 7000 + rank)
            queries = [bytes(r_) for r_ in qa_]
        nq = len(queries)
        qbytes = b"".join(queries)
        off = np.zeros(nq + 1, dtype=np.uint64)
        off[1:] = np.cumsum([len(q) for q in queries])
        d_q = torch.frombuffer(bytearray(qbytes), dtype=torch.uint8).to(dev)  # inputs resident in HBM
        d_off = torch.from_numpy(off.view(np.int64)).to(dev)
        # max_query_len: the bench knows its primers' length, like a primer-design caller does — the library then sizes the batch
        # without reading the offsets back (a host round trip per new buffer; r04a: 0.17 ms of a 0.51 ms step on unseen buffers)
        ham = int(a.hamming)
        p = _capi.HuntParams(distance, ham, 0, 1000, 10000, a.qlen, 0)
        p_compact = _capi.HuntParams(distance, ham, 0, 1000, 10000, a.qlen, _capi.DG_HUNT_COMPACT)
        # The stream of distinct batches (r04): B batches resident in HBM, each with its own bytes and offsets
        # buffer; step k searches batch k mod B, warm-up included, so no batch recurs within B launches.
        dev_batches = [(d_q, d_off, len(qbytes))]
        if a.batches > 1:
            more = np.load(batches_path, mmap_mode="r")
            more = more[0][:, lo_q:hi_q] if strong else more[rank if rank < more.shape[0] else 0]
            for bi in range(more.shape[0]):
                arr = with_n(np.array(more[bi]), 7100 + 16 * rank + bi)
                dev_batches.append((torch.from_numpy(arr.reshape(-1).copy()).to(dev), torch.from_numpy(off.view(np.int64).copy()).to(dev),
                                    int(arr.size)))
        rot = {"k": 0, "on": True}

        def step(fetch=0, handle=None, params=None):
            bq, bo, bbytes = dev_batches[rot["k"] % len(dev_batches)] if rot["on"] else dev_batches[0]
            rot["k"] += 1
            if nq == 0:  # strong scaling: a rank behind the end of the batch still takes part in every gather
                if (world > 1 or a.gather_single) and not fetch:
                    if a.backend == "nccl":
                        gather_block(None)
                    else:
                        gather_parts([torch.empty(0, dtype=torch.uint8, device=dev)])
                return {k_: 0 for k_ in ("nhits", "ext", "leaves", "sa", "win", "tab", "probe", "ops_per_hit", "ms_total", "ms_search",
                                         "ms_search_flat", "ms_select", "ms_locate", "ms_verify", "ms_cap", "cap_dev", "cap_host", "cap_patterns", "t0", "t1", "gen", "form", "vform")}
            rp = C.POINTER(_capi.HuntResult)()
            _capi.check(L, L.dg_hunt_device(handle or ix.handle, C.byref(params or p_compact), sl, len(seqlen), C.c_void_p(bq.data_ptr()),
                                            C.c_void_p(bo.data_ptr()), nq, bbytes, fetch, C.byref(rp)))
            R = rp.contents
            res = {"nhits": R.nhits, "ext": R.ctr_ext_steps, "leaves": R.ctr_leaves, "sa": R.ctr_sa_reads, "win": R.ctr_win_bytes,
                   "tab": R.ctr_tab_reads, "probe": R.ctr_filter_probes, "ops_per_hit": R.ops_per_hit, "ms_total": R.ms_total, "ms_search": R.ms_search,
                   "ms_search_flat": R.ms_search_flat, "ms_select": R.ms_select, "ms_locate": R.ms_locate, "ms_verify": R.ms_verify,
                   "ms_cap": R.ms_cap, "cap_dev": R.cap_queries_device, "cap_host": R.cap_queries_host, "cap_patterns": R.cap_patterns,
                   "t0": R.t_search_begin_ms, "t1": R.t_search_end_ms, "gen": R.t_base_gen, "form": R.flat_kernel_form, "vform": R.verify_kernel_form}
            if (world > 1 or a.gather_single) and not fetch and a.backend == "nccl" and R.compact:
                gather_block(R)
            elif (world > 1 or a.gather_single) and not fetch:
                if R.compact:  # ABI 5 records: position, packed word, d operation words = 12 bytes per hit at distance 1
                    parts = [device_bytes(R.d_hits, R.nhits * 4 * (2 + R.ops_per_hit), dev)]
                else:
                    parts = [device_bytes(R.d_hits, R.nhits * C.sizeof(_capi.Hit), dev)]
                    if R.ops_per_hit and R.nhits:  # alignment description (dicey_gpu.h): 4 bytes per unit of distance
                        parts.append(device_bytes(R.d_ops, R.nhits * R.ops_per_hit * 4, dev))
                gather_parts(parts)
            L.dg_hunt_result_free(rp)
            return res

        # Two batches in flight (r04): step k is handed to one of the handle's two lanes (dg_hunt_device_submit: own stream, workspaces
        # and helper thread each), then step k - 1 is collected (dg_hunt_wait) and, at N > 1, its records go to the gather.  One
        # batch's launch-bound tail — locate, verify, the summary's read-back, the host's turn-around — runs beside the next
        # batch's search kernel; every batch is still one complete pass of the path, and all K start and end inside the timed region.
        inflight = []

        def collect(tk):
            rp = C.POINTER(_capi.HuntResult)()
            _capi.check(L, L.dg_hunt_wait(tk, C.byref(rp)))
            R = rp.contents
            res = {"nhits": R.nhits, "ext": R.ctr_ext_steps, "leaves": R.ctr_leaves, "sa": R.ctr_sa_reads, "win": R.ctr_win_bytes,
                   "tab": R.ctr_tab_reads, "probe": R.ctr_filter_probes, "ops_per_hit": R.ops_per_hit, "ms_total": R.ms_total, "ms_search": R.ms_search,
                   "ms_search_flat": R.ms_search_flat, "ms_select": R.ms_select, "ms_locate": R.ms_locate, "ms_verify": R.ms_verify,
                   "ms_cap": R.ms_cap, "cap_dev": R.cap_queries_device, "cap_host": R.cap_queries_host, "cap_patterns": R.cap_patterns,
                   "t0": R.t_search_begin_ms, "t1": R.t_search_end_ms, "gen": R.t_base_gen, "form": R.flat_kernel_form, "vform": R.verify_kernel_form}
            if (world > 1 or a.gather_single) and a.backend == "nccl" and R.compact:
                gather_block(R)  # staged on the lane's own stream (R.stream): ordered before that lane's next batch
            elif world > 1 or a.gather_single:
                # the lane that ran this batch stays idle until the next step's submit: its result buffers are staged before that
                gather_parts([device_bytes(R.d_hits, R.nhits * 4 * (2 + R.ops_per_hit), dev)] if R.compact else
                             [device_bytes(R.d_hits, R.nhits * C.sizeof(_capi.Hit), dev)] +
                             ([device_bytes(R.d_ops, R.nhits * R.ops_per_hit * 4, dev)] if R.ops_per_hit and R.nhits else []))
            L.dg_hunt_result_free(rp)
            return res

        def step_pipe():
            if nq == 0:
                return step()
            if a.stagger_us > 0 and inflight and len(inflight) < a.in_flight:  # the pipeline is filling: space the lanes' starts
                t_ = time.perf_counter() + a.stagger_us * 1e-6
                while time.perf_counter() < t_:
                    pass
            bq, bo, bbytes = dev_batches[rot["k"] % len(dev_batches)]
            rot["k"] += 1
            tk = C.c_void_p()
            _capi.check(L, L.dg_hunt_device_submit(ix.handle, C.byref(p_compact), sl, len(seqlen), C.c_void_p(bq.data_ptr()), C.c_void_p(bo.data_ptr()),
                                                   nq, bbytes, 0, C.byref(tk)))
            inflight.append(tk)
            return collect(inflight.pop(0)) if len(inflight) >= a.in_flight else None

        def flush_pipe():
            out_ = []
            while inflight:
                out_.append(collect(inflight.pop(0)))
            return out_

        if a.in_flight >= 2:
            acc, elapsed, gathered = timed(step_pipe, flush_pipe)
        else:
            acc, elapsed, gathered = timed(step)
        # the stages' times (HIP events between the kernels, DG_HUNT_PHASE_TIMES) are taken in a pass of their own behind the timed
        # region: every event record is a marker packet in the stream, and the timed steps run without them (the batch total and the
        # search kernel's own time — what roofline.kernel_ms is — are measured in every step)
        p_phase = _capi.HuntParams(distance, ham, 0, 1000, 10000, a.qlen, _capi.DG_HUNT_COMPACT | _capi.DG_HUNT_PHASE_TIMES)
        acc_ph = []
        if rank == 0:
            gp_saved, pipe["g"] = pipe["g"], None
            try:
                for _ in range(max(3, min(a.steps, 10))):
                    rp = C.POINTER(_capi.HuntResult)()
                    bq, bo, bbytes = dev_batches[rot["k"] % len(dev_batches)]
                    rot["k"] += 1
                    if nq:
                        _capi.check(L, L.dg_hunt_device(ix.handle, C.byref(p_phase), sl, len(seqlen), C.c_void_p(bq.data_ptr()),
                                                        C.c_void_p(bo.data_ptr()), nq, bbytes, 0, C.byref(rp)))
                        R = rp.contents
                        acc_ph.append({"ms_total": R.ms_total, "ms_search": R.ms_search, "ms_search_flat": R.ms_search_flat,
                                       "ms_select": R.ms_select, "ms_locate": R.ms_locate, "ms_verify": R.ms_verify})
                        L.dg_hunt_result_free(rp)
            finally:
                pipe["g"] = gp_saved

        # ---------------- extras, outside the timed region (N=1): what delivery costs
        extras = {}
        if world == 1 and nq and not a.no_extras:
            # the headline's witness: the SAME pipeline (rotating batches, a.in_flight in flight, results left in HBM) over at least one
            # second of wall clock, drained inside the timed span — K = 20 steps of 0.19 ms are a 3.7 ms measurement
            n_sus = int(max(a.steps, min(200000, 1.15 / max(elapsed / a.steps, 1e-6))))
            stepf, flushf = (step_pipe, flush_pipe) if a.in_flight >= 2 else (step, None)
            for _ in range(3):
                stepf()
            if flushf:
                flushf()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(n_sus):
                stepf()
            if flushf:
                flushf()
            torch.cuda.synchronize()
            dts = time.perf_counter() - tp
            extras["sustained"] = {"value": nq * n_sus / dts, "unit": "primers/s", "seconds": dts, "steps": n_sus, "ms_per_step": dts / n_sus * 1e3,
                                   "note": "the timed region's pipeline, unchanged, for >= 1 s: every step a complete pass over its batch, all "
                                           "steps started and drained inside the span"}
        if world == 1 and len(dev_batches) > 1:
            # the replayed-batch figure of r01-r03 beside the headline: the same K steps on batch 0 only (its lines stay in the caches)
            rot["on"] = False
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            acc_same = [step() for _ in range(a.steps)]
            torch.cuda.synchronize()
            dts = time.perf_counter() - tp
            extras["value_same_batch"] = {"value": nq * a.steps / dts, "unit": "primers/s", "ms_per_step": dts / a.steps * 1e3,
                                          "kernel_ms": float(np.mean([r["ms_search_flat"] for r in acc_same])),
                                          "note": "the r01-r03 measurement: one batch replayed K times (its 310 MB of index lines are "
                                                  "re-read every step); the headline cycles through %d distinct batches" % len(dev_batches)}
            rot["on"] = True
        if world == 1 and a.in_flight >= 2:
            # the r01-r04a form beside the headline: one batch at a time through dg_hunt_device (rotating batches)
            for _ in range(2):
                step()
            torch.cuda.synchronize()
            tp = time.perf_counter()
            acc_one = [step() for _ in range(a.steps)]
            torch.cuda.synchronize()
            dts = time.perf_counter() - tp
            extras["value_one_in_flight"] = {"value": nq * a.steps / dts, "unit": "primers/s", "ms_per_step": dts / a.steps * 1e3,
                                             "kernel_ms": float(np.mean([r["ms_search_flat"] for r in acc_one])),
                                             "note": "dg_hunt_device, one batch at a time: the kernel's duration without another lane's kernels beside it"}
        pipelined = None
        if world == 1 and not a.no_extras:
            for _ in range(2):
                step(fetch=1, params=p_compact)
            torch.cuda.synchronize()
            tp = time.perf_counter()
            for _ in range(a.steps):
                rlast = step(fetch=1, params=p_compact)
            torch.cuda.synchronize()
            dtf = time.perf_counter() - tp
            words = 2 + int(rlast["ops_per_hit"])
            extras["value_with_d2h"] = {"value": nq * a.steps / dtf, "unit": "primers/s", "ms_per_step": dtf / a.steps * 1e3,
                                        "note": "same steps (rotating batches) with fetch=1 and DG_HUNT_COMPACT: per hit the text position, one "
                                                "packed word and d operation words (dg_chit_unpack / dg_hit_rows rebuild every DnaHit), per "
                                                "query one word of flags + its hit count, in ONE copy into a pinned block of the library's pool",
                                        "bytes_per_step": int(rlast["nhits"]) * 4 * words + 8 * nq}
            # host to host: dg_hunt_submit / dg_hunt_wait on ONE handle, as many batches in flight as the timed region keeps, on the library's internal lanes —
            # query bytes of 16 distinct host batches go up, compact results come back, every step; one Python thread drives it
            try:
                host_batches = []
                hoff = (C.c_uint64 * (nq + 1))(*[int(x) for x in off])
                for bq, _, bbytes in dev_batches:
                    host_batches.append(C.create_string_buffer(bq.cpu().numpy().tobytes(), bbytes))

                def submit(k):
                    t = C.c_void_p()
                    _capi.check(L, L.dg_hunt_submit(ix.handle, C.byref(p_compact), sl, len(seqlen), host_batches[k % len(host_batches)], hoff, nq,
                                                    C.byref(t)))
                    return t

                def wait(t):
                    rp2 = C.POINTER(_capi.HuntResult)()
                    _capi.check(L, L.dg_hunt_wait(t, C.byref(rp2)))
                    nh = rp2.contents.nhits
                    L.dg_hunt_result_free(rp2)
                    return nh
                depth = max(2, a.in_flight)
                for k in range(0, 4 * depth, depth):  # warm EVERY lane (a lane exists only once that many batches are in flight: workspaces, pinned blocks)
                    ts = [submit(k + i) for i in range(depth)]
                    for t in ts:
                        wait(t)
                torch.cuda.synchronize()
                nst = max(a.steps, 2 * len(host_batches))
                tp = time.perf_counter()
                inflight = [submit(i) for i in range(depth)]
                for k in range(nst):
                    nh = wait(inflight[k])
                    if len(inflight) < nst:
                        inflight.append(submit(len(inflight)))
                dth = time.perf_counter() - tp
                extras["host_to_host_pipelined"] = {
                    "value": nq * nst / dth, "unit": "primers/s", "ms_per_step": dth / nst * 1e3, "steps": nst, "hits_last_step": int(nh),
                    "batches_in_flight": depth,
                    "note": "dg_hunt_submit / dg_hunt_wait on one handle (ABI 5: %d batches in flight on the library's internal "
                            "lanes), DG_HUNT_COMPACT, max_query_len given: query bytes of %d distinct host batches up, compact results "
                            "down into pinned blocks; one Python thread drives it" % (depth, len(host_batches))}
            except Exception as e:  # never lose the headline over an extra
                extras["host_to_host_pipelined"] = {"error": repr(e)}
            cli_job = (queries, distance)  # measured in the common tail, after this process has released its own index
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
        cpu = cpu_par = parity = None
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O  # the checker / CPU port; never on the measured GPU path
            t3 = time.time()
            orc = O.Index(fm9)
            info["t_oracle_load_s"] = time.time() - t3
            qstr = [q.decode() for q in queries]
            probe_n = 100 if distance < 2 else 2
            dt, _, _ = orc.hunt_timed(seqlen, qstr[:probe_n], threads=1, distance=distance, hamming=a.hamming)
            per = max(dt / probe_n, 1e-6)
            # three repeats of a third of the budget each, median reported (r02: one run, 25 % spread between lines)
            ns = int(min(nq, max(probe_n, a.cpu_seconds / 3.0 / per)))
            runs = []
            for _ in range(3 if distance < 2 else 1):
                dt, octr, _ = orc.hunt_timed(seqlen, qstr[:ns], threads=1, distance=distance, hamming=a.hamming)
                runs.append(dt)
            dt = sorted(runs)[len(runs) // 2]
            cpu = {"value": ns / dt, "unit": "primers/s", "cores": 1, "kind": "port",
                   "sample": f"first {ns} of the {nq} bench queries, oracle hunt_one (restated hunter.h:291-444) on 1 host thread, "
                             f"median of {len(runs)} runs ({', '.join('%.2f' % x for x in runs)} s), index load excluded",
                   "runs_primers_per_s": [ns / x for x in runs],
                   "host_cpus": os.cpu_count(), "host_physical_cores": phys_cores,
                   "cpu_model": cpu_model, "oracle_ops": octr}
            # the same loop on every physical core over query shards (the reference itself has no threads)
            if phys_cores > 1:
                nsp = int(min(nq, max(phys_cores, ns * phys_cores * 0.6)))
                dtp, _, _ = orc.hunt_timed(seqlen, qstr[:nsp], threads=phys_cores, distance=distance, hamming=a.hamming)
                cpu_par = {"value": nsp / dtp, "unit": "primers/s", "cores": phys_cores, "kind": "port", "cpu_model": cpu_model,
                           "sample": f"first {nsp} bench queries, {phys_cores} host threads (one per physical core) over query shards, {dtp:.1f} s"}
            # parity at full genome size: GPU hits (push order) == oracle hits for a sample; at distance 2 the checker
            # enumerates neighbourhoods with its hash-set form (tested equal to the literal restatement, tests/test_oracle.py)
            npar = min(a.parity_queries if a.parity_queries >= 0 else (1000 if distance < 2 else 300), nq)
            if npar:
                O.fast_neighbors(distance >= 2)
                try:
                    t4 = time.time()
                    got = ix.hunt(qstr[:npar], seqlen, distance=distance, hamming=a.hamming)
                    # the checker on the host's cores: contiguous slices of the sample on up to 32 threads (its loop is single-threaded
                    # like the reference's; a cap-firing 25-mer at distance 2 costs it 2.5 s)
                    from concurrent.futures import ThreadPoolExecutor
                    nthr = max(1, min(32, phys_cores, npar // 4 or 1))
                    per_t = (npar + nthr - 1) // nthr
                    names_ = ["s%d" % i for i in range(len(seqlen))]

                    def _slice(t_):
                        lo_, hi_ = t_ * per_t, min(npar, (t_ + 1) * per_t)
                        if lo_ >= hi_:
                            return []
                        _, hh = orc.hunt(seqlen, names_, qstr[lo_:hi_], distance=distance, hamming=a.hamming, want_hits=True)
                        return [(h[0] + lo_,) + tuple(h[1:]) for h in hh]
                    with ThreadPoolExecutor(nthr) as ex:
                        ohits = [h for part in ex.map(_slice, range(nthr)) for h in part]
                finally:
                    O.fast_neighbors(False)
                perq = {}
                for h in ohits:
                    perq.setdefault(h[0], []).append(h[1:])
                mism = 0
                for qi, qr in enumerate(got.queries):
                    gh = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
                    mism += gh != perq.get(qi, [])
                parity = {"queries": npar, "mismatching": mism, "hits": len(ohits), "seconds": time.time() - t4}

        if rank == 0:
            mean = lambda k: float(np.mean([r[k] for r in acc]))  # noqa: E731
            ext, tab, probe = mean("ext"), mean("tab"), mean("probe")
            flat = mean("ms_search_flat")
            # distance 1: k_search1s = the flat search with the select stage inside (r03); DICEY_NO_FUSED_SELECT gives k_search1p
            # which instantiation ran is the library's word (dg_hunt_result::flat_kernel_form of the last timed step), not a guess: on a
            # repeat-bearing genome the generic kernels stay on and k_search1s runs without the take stage (<true, false>)
            form = int(acc[-1].get("form", 0)) if acc else 0
            ind = "false" if a.hamming else "true"
            k1 = {1: f"k_search1p<{ind}>", 2: f"k_search1s<{ind}, false>", 3: f"k_search1s<{ind}, true>"}.get(form, f"k_search1s<{ind}, true>")
            k2 = {4: "k_search2p<false, false>", 5: "k_search2p<true, false>", 6: "k_search2p<false, true>", 7: "k_search2p<true, true>"}.get(form, "k_search2p<true, true>")
            # the general kernel (queries with N, above 31 nt, Hamming distance >= 2, distance >= 3): dominant when the flat kernels took
            # less than half of the search phase (phase events of the pass behind the timed region)
            ms_phase = float(np.mean([r["ms_search"] for r in acc_ph])) if acc_ph else mean("ms_search")
            generic_dominant = flat <= 0 or (acc_ph and float(np.mean([r["ms_search_flat"] for r in acc_ph])) < 0.5 * ms_phase)
            kernel = f"k_search<{'false' if a.hamming else 'true'}, {distance}>" if generic_dominant else (k1 if distance == 1 else k2)
            kernel_ms = ms_phase if generic_dominant else flat
            # Two batches in flight: the launches of neighbouring batches overlap, and the sum of their durations counts that time twice.
            # The time the kernel RAN is the union of the launches' intervals on the lanes' common timeline (dg_hunt_result::t_search_*,
            # HIP events of the timed steps); per launch it is what the roofline divides by.  launch_ms keeps the plain duration.
            launch_ms = kernel_ms
            busy = None
            if a.in_flight >= 2 and flat > 0 and not generic_dominant:
                by_gen = {}
                for r in acc:
                    if r.get("gen"):
                        by_gen.setdefault(r["gen"], []).append((r["t0"], r["t1"]))
                n_iv = sum(len(v) for v in by_gen.values())
                if n_iv >= max(2, (len(acc) * 4) // 5):
                    union = sum(union_of_intervals(iv) for iv in by_gen.values())
                    busy = {"launches": n_iv, "union_ms": union, "sum_of_durations_ms": float(sum(e - s_ for v in by_gen.values() for s_, e in v))}
                    kernel_ms = union / n_iv
            alg_bytes = ext * BYTES_PER_EXT + tab * BYTES_PER_TAB_READ + probe * BYTES_PER_FILTER_PROBE
            # the same launch in SURVEY.md §8(d) units: a backward step on c = 2 L(c) rank ops of 24 B on the sdsl layout
            # (L = Huffman code length in the loaded wavelet tree), small reads by their payload
            cl = st["code_len"]
            avg_l = sum(f * cl.get(ord(ch), 0) for f, ch in zip(GRCH38_FREQ, "ACGT"))
            survey_bytes = ext * 2 * avg_l * 24 + tab * BYTES_PER_TAB_READ + probe * BYTES_PER_FILTER_PROBE
            achieved = alg_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0
            # HBM bytes per launch of this kernel on this workload from the PMC passes of the profile round (tools/summarize_profile.py:
            # FETCH_SIZE corrected by the calibration factor + WRITE_SIZE, the row of exactly this kernel instantiation) — carried only
            # when that round profiled the sources this library was built from (build_id); a stale profile gives null, never a number
            traffic = fabric_reads = traffic_round = write_bytes = None
            wtag = workload_tag(units, a.qlen, distance, a.hamming, a.n_frac, a.genome_size, a.genome)
            tname = traffic_file(distance, a.hamming, a.n_frac, a.genome)
            tj = profile_of_this_build(tname, workload=wtag, kernel=kernel)
            if tj is not None:
                traffic = tj.get("hbm_bytes_per_launch")
                write_bytes = tj.get("write_bytes_per_launch")
                fabric_reads = (tj.get("tcc") or {}).get("ea_rdreq")
                traffic_round = tj.get("round")
            out = dict(base_out)
            out.update({
                "value": (len(meta["queries"][0]) if strong else world * nq) * a.steps / elapsed, "ms_per_step": elapsed / a.steps * 1e3, "dtype": "u32",
                "config": {"workload": f"dicey hunt, {nq} synthetic {a.qlen}-mers per GPU, {'Hamming' if a.hamming else 'edit'} distance {distance}, both strands, "
                                       + (f"{a.n_frac:.0%} of the queries with one N, " if a.n_frac > 0 else "") +
                                       f"-m 1000 -x 10000 (BASELINE.json configs[{1 if cfg == 'hunt_d1' else 3}]" +
                                       (" with -n" if a.hamming else "") + ")",
                           "genome": genome_desc, "genome_short": genome_short,
                           "index": "sdsl csa_wt<> .fm9 built by dg_index_build_device, loaded unchanged by dg_index_open",
                           "workload_tag": wtag, "traffic_file": tname,
                           **({"total_queries": a.total_queries,
                               "chunks": f"{a.total_queries} distinct queries as {a.batches} chunks of {nq}, {a.in_flight} in flight; every chunk is searched once, "
                                         "inside the timed region (steps = chunks)"} if a.total_queries else {}),
                           "queries_per_gpu": nq, "sharding": f"query-sharded x{world}, full index replica per GPU",
                           "distinct_batches": len(dev_batches), "in_flight_batches": a.in_flight, "stagger_us": a.stagger_us,
                           "results": "the batch's compact block (DG_HUNT_COMPACT: 8 B per query + 8 + 4 d B per hit) left in HBM (N = 1) / gathered to rank 0 "
                                      "over RCCL by the C++ gather, libdiceygather.so (N > 1)",
                           "in_flight": (f"{a.in_flight} batches per GPU (dg_hunt_device_submit / dg_hunt_wait on the handle's lanes: step k is submitted, "
                                         f"then step k - {a.in_flight - 1} collected; K batches start and end inside the timed region)" if a.in_flight >= 2
                                         else "1 batch per GPU (dg_hunt_device)"),
                           "stream": f"step k searches batch k mod {len(dev_batches)} of {len(dev_batches)} distinct batches resident in HBM "
                                     "(seeds 42 + 1000 b), warm-up included" if len(dev_batches) > 1 else "one batch replayed every step"},
                "roofline": {"bound": "hbm", "kernel": kernel, "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": achieved / HBM_PEAK_GBS, "traffic": traffic,
                             "traffic_over_algorithmic": (traffic / alg_bytes) if (traffic and alg_bytes) else None,
                             "write_bytes_per_launch": write_bytes,
                             "algorithmic_bytes_per_launch": alg_bytes, "ext_steps_per_launch": ext,
                             "bytes_per_ext_step": BYTES_PER_EXT, "table_reads_per_launch": tab,
                             "bytes_per_table_read": BYTES_PER_TAB_READ, "filter_probes_per_launch": probe,
                             "table_reads_note": "K-mer table entries (8 B) and, since r04, reads of the preceding-characters array (2 B per suffix of a "
                                                 "narrow interval, <= 32 B) counted together at 8 B each",
                             "bytes_per_filter_probe": BYTES_PER_FILTER_PROBE, "kernel_ms": kernel_ms,
                             "kernel_ms_is": ("busy time per launch: union of the timed launches' intervals (HIP events on the lanes' common timeline) / launches — "
                                              "launches of neighbouring batches overlap with two in flight" if busy else "average launch duration (HIP events around the launch)"),
                             "launch_ms": launch_ms, "busy": busy,
                             "survey_units": {"bytes_per_launch": survey_bytes, "avg_code_len": avg_l,
                                              "achieved": survey_bytes / (kernel_ms * 1e-3) / 1e9 if kernel_ms > 0 else 0.0,
                                              "frac": survey_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0,
                                              "note": "24 B per rank op, 2 L(c) rank ops per backward step (SURVEY.md 8(d))"},
                             # the kernel is a gather of 64-byte lines: what the memory system delivered (TCC_EA0_RDREQ of the
                             # committed PMC pass for this workload) against this launch's duration
                             "fabric_reads_per_launch": fabric_reads, "traffic_from_profile_round": traffic_round,
                             "fabric_reads_per_s": (fabric_reads / (kernel_ms * 1e-3)) if (fabric_reads and kernel_ms > 0) else None,
                             "lines_per_strand": (fabric_reads / (2 * nq)) if fabric_reads else None,
                             # what bounds this kernel: L2-to-fabric requests per second against the chip's measured rate for independent
                             # random 64-byte lines (gather_reference below, 4 GiB footprint; the byte fraction above is what the contract
                             # asks for and falls whenever the kernel is taught to need fewer bytes)
                             # r05: the reference is the kernel-SHAPED gather (tools/microbench/gather_bench filter2: four copies of 2^27
                             # lines, a lane per (strand, position), eight loads per lane, the lines of one operation kind and filter
                             # field shared — 12 lines per strand, 8 wavefronts per SIMD, nothing but the loads): 42.3 G lines/s,
                             # profiles/r05_gather_filter2.jsonl.  r04 compared with 26.5 G/s of a microbenchmark that was slower
                             # than the kernel it was meant to bound (its address arithmetic and 38 lines per wave instruction)
                             "request_rate": ({"fabric_requests_per_s": fabric_reads / (kernel_ms * 1e-3), "reference_requests_per_s": 42.3e9,
                                               "frac": fabric_reads / (kernel_ms * 1e-3) / 42.3e9,
                                               "reference": "gather_bench filter2, 12 lines per strand (profiles/r05_gather_filter2.jsonl)"}
                                              if (fabric_reads and kernel_ms > 0) else None),
                             # two batches in flight: the launches of neighbouring batches overlap each other and the other batch's
                             # locate / verify kernels, so a launch lasts longer than it does alone; its duration alone (the
                             # value_one_in_flight pass right behind the timed region: same batches, dg_hunt_device) and what follows from it
                             "one_in_flight": ({"kernel_ms": extras["value_one_in_flight"]["kernel_ms"],
                                                "achieved": alg_bytes / (extras["value_one_in_flight"]["kernel_ms"] * 1e-3) / 1e9,
                                                "frac": alg_bytes / (extras["value_one_in_flight"]["kernel_ms"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                                                "request_rate_frac": (fabric_reads / (extras["value_one_in_flight"]["kernel_ms"] * 1e-3) / 42.3e9) if fabric_reads else None}
                                               if extras.get("value_one_in_flight", {}).get("kernel_ms") else None),
                             "index_accesses_per_query": (2 * ext + tab + probe) / nq,
                             "index_accesses_per_s": (2 * ext + tab + probe) / (kernel_ms * 1e-3) if kernel_ms > 0 else 0.0,
                             # the same launch in three conventions, side by side: this build's units (above), SURVEY 8(d)
                             # units, and rank operations only (no byte is counted for a filter probe or a table entry)
                             "frac_by_convention": {
                                 "own_units": achieved / HBM_PEAK_GBS,
                                 "survey_units": survey_bytes / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0,
                                 "rank_ops_only": ext * 2 * avg_l * 24 / (kernel_ms * 1e-3) / 1e9 / HBM_PEAK_GBS if kernel_ms > 0 else 0.0},
                             # measured reference rates of this chip for random 64-byte lines (tools/microbench/gather_bench, r03): NOT
                             # ceilings for this kernel — its own line rate (fabric_reads_per_s) is above the uniform-random rate of
                             # its footprint because two thirds of its lines are filter lines inside 4 x 8.6 GB
                             "gather_reference": {
                                 "uniform_random_lines_Glines_per_s": {"4GiB": 26.5, "16GiB": 19.7, "64GiB": 19.1, "170GiB": 19.0},
                                 "with_200k_lanes_only": {"4GiB": 21.8, "16GiB": 16.7, "170GiB": 16.0},
                                 "filter_geometry_4_copies_of_8.6GB": {"6_lines_per_strand": 20.8, "12_lines_per_strand": 9.4,
                                                                         "24_lines_per_strand": 11.3},
                                 "kernel_shaped_filter2_Glines_per_s": {"12_lines_per_strand": 42.3, "15_lines_per_strand": 41.8},
                                 "source": "docs/history/profiles/r03a_gather_matrix.jsonl, docs/history/profiles/r03a_gather_filter.jsonl, profiles/r05_gather_filter2.jsonl"}},
                "cpu_baseline": cpu, "cpu_baseline_parallel": cpu_par, "pipelined": pipelined, "parity_sample": parity,
                "phases_ms": ({k: float(np.mean([r[k] for r in acc_ph])) for k in ("ms_total", "ms_search", "ms_search_flat", "ms_select", "ms_locate", "ms_verify")}
                              if acc_ph else {k: mean(k) for k in ("ms_total", "ms_search", "ms_search_flat", "ms_select", "ms_locate", "ms_verify")}),
                "ms_total_timed_steps": mean("ms_total"),
                "hits_per_step": int(acc[-1]["nhits"]), "leaves_per_step": int(acc[-1]["leaves"]),
            })
            out.update(extras)
            if a.total_queries:
                out["seconds"] = elapsed
                out["queries"] = a.total_queries
            # Capped neighbourhoods (neighbors.h:50) are settled in front of the batch: on the device (k_cap_enum) for A/C/G/T primers up to
            # 29 nt at edit distance <= 2, on the host otherwise.  When that stage is the longest of the step its kernel is the dominant one.
            cap_ms = mean("ms_cap")
            if cap_ms > 0:
                out["cap_stage"] = {"ms": cap_ms, "queries_on_device": int(mean("cap_dev")), "queries_on_host": int(mean("cap_host")),
                                    "explicit_patterns": int(mean("cap_patterns")),
                                    "note": "host wall clock of the stage (classification of the batch, k_cap_enum, the read-back of the pattern count)"}
            if cap_ms > max(out["phases_ms"][k_] for k_ in ("ms_search", "ms_select", "ms_locate", "ms_verify")) and mean("cap_dev") > 0:
                out["roofline_search"] = out["roofline"]
                out["roofline"] = cap_enum_roofline(2 * mean("cap_dev"), a.qlen, distance, cap_ms)
            # The roofline block above describes the search kernel.  When another stage takes longer (the repeat-bearing genome:
            # hundreds of hits per query), the step's dominant kernel is that stage's, and it gets its own block under the same key;
            # the search kernel's block moves to roofline_search.
            ph = out["phases_ms"]
            stage = max(("ms_search", "ms_select", "ms_locate", "ms_verify"), key=lambda k_: ph[k_])
            if stage in ("ms_locate", "ms_verify") and ph[stage] > 0:
                hits, sa = mean("nhits"), mean("sa")
                oph = int(acc[-1].get("ops_per_hit", 0))
                if stage == "ms_locate":
                    dom_kernel = "k_locate_topk<576u>"
                    dom_bytes = 4.0 * sa + 16.0 * hits
                    terms = {"sa_or_minima_words_read": sa, "bytes_per_word": 4, "hit_seeds_written": hits, "bytes_per_seed": 16,
                             "note": "the stage is four kernels (k_locate, k_locate_small, k_locate_topk<576u>, k_locate_topk<1152u>); the named one is "
                                     "its longest on a repeat-rich genome, the bytes are the whole stage's"}
                else:
                    vf = int(acc[-1].get("vform", 0)) if acc else 0
                    dom_kernel = f"k_verify_memo<{vf >> 8}, {vf & 255}>" if vf else ("k_verify_memo<7, 1>" if distance <= 1 else "k_verify_memo<13, 1>")
                    dom_bytes = mean("win") + hits * (16.0 + 8.0 + 4.0 * oph)
                    terms = {"window_bytes": mean("win"), "hits": hits, "bytes_per_hit": 16 + 8 + 4 * oph,
                             "note": "window bytes as the reference extracts them (hunter.h:371), 16 B seed in, 8 B compact record + 4 B "
                                     "per unit of distance out; the kernel itself takes the <= 2d context characters of a hit from its seed "
                                     "(FmView::sax) or the text and aligns once per distinct window"}
                dom_ach = dom_bytes / (ph[stage] * 1e-3) / 1e9
                # HBM bytes of the stage's kernel from the PMC passes of a profile of THIS build on THIS workload (else null)
                sj = profile_of_this_build("traffic_stage_" + wtag + ".json", workload=wtag, kernel=dom_kernel)
                out["roofline_search"] = out["roofline"]
                out["roofline"] = {"bound": "hbm", "kernel": dom_kernel, "achieved": dom_ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": dom_ach / HBM_PEAK_GBS, "traffic": sj.get("hbm_bytes_per_launch") if sj else None,
                                   "traffic_over_algorithmic": (sj["hbm_bytes_per_launch"] / dom_bytes) if (sj and dom_bytes) else None,
                                   "fabric_reads_per_launch": ((sj.get("tcc") or {}).get("ea_rdreq")) if sj else None,
                                   "algorithmic_bytes_per_launch": dom_bytes,
                                   "kernel_ms": ph[stage], "kernel_ms_is": "average launch duration (HIP events around the stage, one batch at a time)",
                                   "stage": stage, "terms": terms,
                                   "note": "dominant stage of this step by HIP events on the index stream (phases_ms); the stage's launches "
                                           "are timed together"}
            out["roofline"] = rocprof_average(out["roofline"], wtag)
            if "roofline_search" in out:
                out["roofline_search"] = rocprof_average(out["roofline_search"], wtag)
    # =====================================================================================================================
    elif cfg == "search":
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import p3config
        th = dicey_amd.Thal(p3config.config_dir(), device=local)
        prim = meta["primers"][rank if rank < len(meta["primers"]) else 0]
        buf = "".join(prim).encode()
        poff = (C.c_uint64 * (len(prim) + 1))()
        t = 0
        for i, s_ in enumerate(prim):
            poff[i] = t
            t += len(s_)
        poff[len(prim)] = t
        sp = _capi.SearchParams(distance, 0, 10000, 10000, 15, 45.0)

        def step():
            rp = C.POINTER(_capi.SearchResult)()
            _capi.check(L, L.dg_search_sites(ix.handle, th._h, C.byref(sp), sl, len(seqlen), buf, poff, len(prim), C.byref(rp)))
            R = rp.contents
            res = {"nsites": R.nsites, "nhits": R.nhits, "ms_device": R.ms_device, "ms_fm": R.ms_fm_search, "ms_site": R.ms_site_stage,
                   "ext": R.ctr_ext_steps, "tab": R.ctr_tab_reads, "probe": R.ctr_filter_probes}
            if world > 1:  # the binding sites of every rank's primers go to rank 0, where the amplicon pairing runs
                nb = R.nsites * C.sizeof(_capi.Site)
                sites = torch.frombuffer(bytearray(C.string_at(R.sites, nb)), dtype=torch.uint8) if nb else torch.empty(0, dtype=torch.uint8)
                gather_parts([sites.to(dev if a.backend == "nccl" else "cpu")])
            L.dg_search_result_free(rp)
            return res

        acc, elapsed, gathered = timed(step)
        cpu = parity = None
        if rank == 0 and world == 1 and not a.no_cpu_baseline and host_text is not None:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            if O.ref_libs() is not None:
                orc = O.Index(fm9)
                names = ["s%d" % i for i in range(len(seqlen))]
                ns = 4
                tc = time.time()
                orc.search(seqlen, names, host_text, "".join(">p%d\n%s\n" % (i, s_) for i, s_ in enumerate(prim[:ns])))
                per = (time.time() - tc) / ns
                ns = int(min(len(prim), max(4, a.cpu_seconds / max(per, 1e-3)))) & ~1
                tc = time.time()
                ojs, _ = orc.search(seqlen, names, host_text, "".join(">p%d\n%s\n" % (i, s_) for i, s_ in enumerate(prim[:ns])))
                dtc = time.time() - tc
                cpu = {"value": ns / dtc, "unit": "primers/s", "cores": 1, "kind": "reference", "cpu_model": cpu_model,
                       "sample": f"first {ns} primers, restated silica.h:429-640 calling the reference's own thal.h (oracle/_ref), 1 host thread, {dtc:.1f} s"}
                orc.close()
                # parity at full size: the binding sites of the same primers from the GPU path (dg_search_sites) against the checker's
                # "primers" records — chromosome, position, strand, Tm and MatchTm as doubles (the JSON number round-trips), site sequence
                try:
                    want_ = {}
                    for r_ in json.loads(ojs)["data"]["primers"]:
                        want_.setdefault(r_["Name"], []).append((r_["Chrom"], r_["Pos"], r_["Ori"], float(r_["Tm"]), float(r_["MatchTm"]), r_["Genome"]))
                    sites_, mt_, _, _ = dicey_amd.search_sites(ix, th, prim[:ns], seqlen, kmer=15, distance=distance, max_locations=10000,
                                                               max_neighborhood=10000, cut_temp=45.0)
                    got_ = {}
                    for s_ in sites_:
                        got_.setdefault("p%d" % s_["primer"], []).append((names[s_["ref"]], s_["pos"] + 1, "forward" if s_["on_for"] else "reverse",
                                                                          float(s_["temp"]), float(s_["perf_temp"]), s_["genome"]))
                    bad_ = sum(sorted(got_.get("p%d" % i, [])) != sorted(want_.get("p%d" % i, [])) for i in range(ns))
                    parity = {"queries": ns, "mismatching": int(bad_), "hits": sum(len(v) for v in want_.values()),
                              "what": "binding sites per primer (chromosome, position, strand, Tm, MatchTm, site sequence) vs the checker's JSON"}
                except Exception as e:
                    parity = {"error": repr(e)[:200]}
        if rank == 0:
            mean = lambda k: float(np.mean([r[k] for r in acc]))  # noqa: E731
            ext, tab, probe = mean("ext"), mean("tab"), mean("probe")
            alg_bytes = ext * BYTES_PER_EXT + tab * BYTES_PER_TAB_READ + probe * BYTES_PER_FILTER_PROBE
            ms_fm = mean("ms_fm")
            achieved = alg_bytes / (ms_fm * 1e-3) / 1e9 if ms_fm > 0 else 0.0
            out = dict(base_out)
            out.update({
                "metric": "primers/sec, `dicey search` binding sites (FM search of the k-mer neighbourhoods + thal() of every located hit)",
                "value": world * len(prim) * a.steps / elapsed, "ms_per_step": elapsed / a.steps * 1e3, "dtype": "u32 + f64",
                "config": {"workload": f"dicey search, {len(prim) // 2} primer pairs = {len(prim)} primers (18-25 nt) per GPU, -k 15 -d {distance} -c 45 "
                                       f"-m 10000 (BASELINE.json configs[2]); binding-site stage, host buffers in and out",
                           "genome": genome_desc, "genome_short": genome_short, "primers_per_gpu": len(prim), "sharding": f"primer-sharded x{world}, full index replica per GPU"},
                # the step IS k_site_wave (one wavefront per located hit: f64 thal() DP in LDS): bound by VALU instruction issue, so the
                # roofline block is that kernel against the chip's issue rate (1 024 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction);
                # instructions per thal() come from the PMC pass of THIS build (null when the committed pass is of other sources)
                "roofline": (lambda tcb: {"bound": "valu", "kernel": "k_site_wave", "achieved": (tcb["valu_wave_instructions_per_s"] / 1e9) if tcb else None,
                                          "peak": 614.4, "unit": "G wave-instructions/s", "frac": tcb["valu_issue_frac"] if tcb else None, "traffic": None,
                                          "kernel_ms": mean("ms_site"), "thal_calls_per_launch": mean("nhits"),
                                          "thal_per_s": mean("nhits") / (mean("ms_site") * 1e-3) if mean("ms_site") > 0 else 0.0,
                                          "valu_wave_instructions_per_thal": tcb["valu_wave_instructions_per_thal"] if tcb else None,
                                          "note": "HBM is not the limit (window + primer bytes in, 32 B out per hit; the DP stays in LDS)"})(
                    thal_counter_block("search", "k_site_wave", mean("nhits"), mean("ms_site"))),
                "roofline_fm_search": {"bound": "hbm", "kernel": "k_search1s<true, false> (FM search of the 15-mer neighbourhoods)", "achieved": achieved,
                                       "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                                       "algorithmic_bytes_per_launch": alg_bytes, "kernel_ms": ms_fm},
                "site_stage": {"kernel": "k_site_wave", "ms": mean("ms_site"), "thal_calls_per_step": mean("nhits"),
                               "thal_per_s": mean("nhits") / (mean("ms_site") * 1e-3) if mean("ms_site") > 0 else 0.0,
                               "binding_sites_per_step": mean("nsites"),
                               "fp64_and_lds": thal_counter_block("search", "k_site_wave", mean("nhits"), mean("ms_site"))},
                "cpu_baseline": cpu, "parity_sample": parity,
                "phases_ms": {"ms_device": mean("ms_device"), "ms_fm_search": ms_fm, "ms_site_stage": mean("ms_site")},
            })
    # =====================================================================================================================
    else:  # padlock
        sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
        import p3config
        th = dicey_amd.Thal(p3config.config_dir(), device=local)
        exons, gene_of = meta["exons"][rank if rank < len(meta["exons"]) else 0]
        ebytes = [e.encode("latin-1") for e in exons]

        def step():
            t1 = time.perf_counter()
            R = dicey_amd.padlock_scan(ix, th, ebytes)
            res = {"npos": int(R["pos_off"][-1]), "arm_thal": int(R["n_arm_thal"]), "probe_thal": int(R["n_probe_thal"]),
                   "arms_counted": int(R["n_arms_counted"]), "ms": (time.perf_counter() - t1) * 1e3}
            if world > 1:
                parts = [torch.from_numpy(R[k].view(np.uint8).copy()) for k in ("arm_tm", "probe_tm", "arm_count", "arm_nbcount")]
                gather_parts([p_.to(dev if a.backend == "nccl" else "cpu") for p_ in parts])
            return res

        acc, elapsed, gathered = timed(step)
        cpu = parity = None
        if rank == 0 and world == 1 and not a.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib as O
            if O.ref_libs() is not None:
                orc = O.Index(fm9)
                ng = 1
                while True:
                    sel = [e for e in range(len(exons)) if gene_of[e] < ng]
                    bars = "".join(">%d\nACGTTGCAACGTTGCAACGT\n" % i for i in range(len(sel)))
                    tc = time.time()
                    otsv = orc.padlock(["e%d" % e for e in sel], [exons[e] for e in sel], "", bars, input_fasta=True, spacerleft="GC",
                                       spacerright="GC", anchor="GCGCGCATATGCGCGCATAT")[0]
                    dtc = time.time() - tc
                    if dtc >= a.cpu_seconds / 2 or ng >= units:
                        break
                    ng = min(units, max(ng + 1, int(ng * a.cpu_seconds / max(dtc, 0.1))))
                pos = sum(len(exons[e]) for e in sel)
                cpu = {"value": ng / dtc, "unit": "genes/s", "cores": 1, "kind": "reference", "cpu_model": cpu_model,
                       "positions_per_s": pos / dtc,
                       "sample": f"{len(sel)} exons of the first {ng} genes ({pos} positions), restated padlock.h:321-520 calling the reference's own "
                                 f"thal.h (oracle/_ref), 1 host thread, {dtc:.1f} s"}
                # parity at full size: the probe positions the checker accepted for these genes against a replay of the reference's
                # per-position decisions (padlock.h:321-428, :506) on the arrays dg_padlock_scan returned for the whole batch
                try:
                    Rp = dicey_amd.padlock_scan(ix, th, ebytes)
                    rows_ = [ln.split("\t") for ln in otsv.rstrip("\n").split("\n")[1:] if ln]
                    want_ = sorted((r_[0], int(r_[3].split(":")[1]) - 1) for r_ in rows_)
                    got_ = sorted(("e%d" % e, k_) for e in sel for k_ in padlock_replay(Rp, e, len(exons[e])))
                    bad_genes = {gene_of[int(n_[1:])] for n_, k_ in set(want_) ^ set(got_)}
                    parity = {"queries": ng, "mismatching": len(bad_genes), "hits": len(want_), "exons": len(sel),
                              "what": "accepted probe positions per gene: checker's TSV vs the reference's decisions replayed on the GPU arrays"}
                except Exception as e:
                    parity = {"error": repr(e)[:200]}
        if rank == 0:
            mean = lambda k: float(np.mean([r[k] for r in acc]))  # noqa: E731
            npos = mean("npos")
            win_bytes = mean("arm_thal") * 20 + mean("probe_thal") * 40 + mean("arms_counted") * 20
            out = dict(base_out)
            out.update({
                "metric": "genes/sec, `dicey padlock` per-position scan (arm / probe thal, exact and neighbourhood counts of surviving arms)",
                "unit": "genes/s", "value": world * units * a.steps / elapsed, "ms_per_step": elapsed / a.steps * 1e3, "dtype": "f64 + u32",
                "config": {"workload": f"dicey padlock, {units} synthetic genes = {len(exons)} exons = {int(npos)} arm windows per GPU, armlen 20, "
                                       f"-d {distance} (BASELINE.json configs[4]; GTF parsing and TSV writing are host work outside this step)",
                           "genome": genome_desc, "genome_short": genome_short, "genes_per_gpu": units, "sharding": f"gene-sharded x{world}, full index replica per GPU"},
                # what bounds the scan: the thal() wave kernel against the chip's VALU issue rate (counters of the PMC pass of THIS build
                # on this workload, scaled by thal() calls; null when the committed pass is of other sources)
                "roofline": (lambda tcb: {"bound": "valu", "kernel": "k_thal_self_wave", "achieved": (tcb["valu_wave_instructions_per_s"] / 1e9) if tcb else None,
                                          "peak": 614.4, "unit": "G wave-instructions/s", "frac": tcb["valu_issue_frac"] if tcb else None, "traffic": None,
                                          "kernel_ms": elapsed / a.steps * 1e3, "thal_calls_per_launch": mean("arm_thal") + mean("probe_thal"),
                                          "thal_per_s": (mean("arm_thal") + mean("probe_thal")) / (elapsed / a.steps),
                                          "valu_wave_instructions_per_thal": tcb["valu_wave_instructions_per_thal"] if tcb else None,
                                          "hbm_window_bytes_per_launch": win_bytes,
                                          "note": "kernel_ms is the whole step (thal of arm and probe windows + the counts of surviving arms); HBM is "
                                                  "not the limit (window bytes in, 8 B out per thal())"})(
                    thal_counter_block("padlock", "k_thal_self_wave", mean("arm_thal") + mean("probe_thal"), elapsed / a.steps * 1e3)),
                "thal_stage": {"kernel": "k_thal_self_wave", "thal_per_s": (mean("arm_thal") + mean("probe_thal")) / (elapsed / a.steps),
                               "fp64_and_lds": thal_counter_block("padlock", "k_thal_self_wave", mean("arm_thal") + mean("probe_thal"),
                                                                  elapsed / a.steps * 1e3)},
                "positions_per_s": world * npos * a.steps / elapsed, "arm_thal_per_step": mean("arm_thal"), "probe_thal_per_step": mean("probe_thal"),
                "arms_counted_per_step": mean("arms_counted"), "cpu_baseline": cpu, "parity_sample": parity,
            })

    # ---------------- common tail
    if a.dump_gather and pipe["g"] is not None:
        os.makedirs(a.dump_gather, exist_ok=True)
        pipe["g"].finish()  # the passes behind the timed region gathered too: what is compared is the last payload every rank staged
        open(os.path.join(a.dump_gather, f"local_{rank}.bin"), "wb").write(pipe.get("last_local", b""))
        if rank == 0:
            for r, payload in enumerate(pipe["g"].last_received()):
                open(os.path.join(a.dump_gather, f"gathered_{r}.bin"), "wb").write(payload)
    if rank == 0 and out is not None and (world > 1 or a.gather_single):
        out["config"]["gather"] = ("libdiceygather.so: C++ over RCCL, exact-size ncclSend / ncclRecv one step behind the size exchange" if pipe.get("cxx_on")
                                   else "torch.distributed (dicey_amd/shard.py)" + (" — FALLBACK: " + pipe["cxx_off"] if pipe.get("cxx_off") else ""))
    if rank == 0 and out is not None:
        out["index"] = {"n": st["n"], "file_bytes": st["file_bytes"], "hbm_bytes": st["hbm_bytes"],
                        "load_s": st["load_seconds"], "derive_s": st["derive_seconds"]}
        out["setup_s"] = info
        if world > 1 or a.gather_single:
            out["gathered_bytes_per_step"] = gathered / max(1, a.steps)
            if pipe["g"] is not None:
                out["gather_bytes_moved_per_step"] = pipe.get("moved", 0) / max(1, a.steps)
    if isinstance(pipe["g"], CxxGather):
        pipe["g"].c.close()  # the communicator goes before the index whose streams its staging copies were queued on
    for h in shared[1:]:
        h.close()
    if th is not None:
        th.close()
    ix.close()
    if rank == 0 and out is not None:
        if cli_job is not None:  # the process seam on a GPU this process no longer occupies (r02f and earlier measured it while
            # the bench still held its 199 GB index, which left the CLI a smaller table and no long filter)
            out["cli_end_to_end_after_release"] = cli_end_to_end(fm9, meta, cli_job[0], cli_job[1])  # right after 199 GB were freed
            time.sleep(8)  # the driver wipes released VRAM at ~32 GiB/s; allocations of the next process wait for it
            out["cli_end_to_end"] = cli_end_to_end(fm9, meta, cli_job[0], cli_job[1])
            if os.path.exists(cli_big_path):
                try:
                    big = np.load(cli_big_path)
                    os.remove(cli_big_path)
                    chk = None
                    if not a.no_cpu_baseline:
                        sys.path.insert(0, os.path.join(ROOT, "tests"))
                        import oracle_lib as O
                        chk = O.Index(fm9)
                    # (every run below starts 5 s after the previous process left: the 98 GB it released are wiped by then — r06's final
                    #  call saw `text + suffix array` take 4.3 s instead of 0.3 s in a process started a second after another one's exit)
                    time.sleep(5)
                    out["cli_end_to_end_1M"] = cli_end_to_end(fm9, meta, big[:1000000], cli_job[1], oracle=chk, seqlen=seqlen)
                    time.sleep(5)
                    out["cli_end_to_end_10M" if len(big) == 10000000 else "cli_end_to_end_%d" % len(big)] = \
                        cli_end_to_end(fm9, meta, big, cli_job[1], oracle=chk, seqlen=seqlen)
                    # the same seam at edit distance 2 (BASELINE configs[3]'s distance): 59 hits and ~11 KB of JSON per query on this genome,
                    # i.e. 110 GB for 10 M queries — the reference would write the same — so this run takes the first 1 M (11 GB)
                    if chk is not None:
                        O.fast_neighbors(True)
                    time.sleep(5)
                    try:
                        out["cli_end_to_end_d2_1M"] = cli_end_to_end(fm9, meta, big[:1000000], 2, oracle=chk, seqlen=seqlen, bytes_per_query=12000)
                    finally:
                        if chk is not None:
                            O.fast_neighbors(False)
                except Exception as e:
                    out["cli_end_to_end_10M"] = {"error": repr(e)[:300]}
        if (world == 1 and cfg == "hunt_d1" and a.genome == "iid" and not a.no_extra_configs and not a.no_extras and not a.queries
                and a.distance < 0 and a.qlen == 20):
            out["extra_configs"] = run_extra_configs(a, fm9)
            # compact top-level summaries of the sub-lines (the driver's record keeps top-level keys whole)
            for name_ in ("hunt_d1_repeats", "hunt_d2", "hunt_d2_10M", "hunt_d2_25mers", "hunt_d2_hamming", "hunt_d1_nmix", "search", "padlock"):
                sub_ = out["extra_configs"].get(name_)
                if isinstance(sub_, dict) and "value" in sub_:
                    rf_ = sub_.get("roofline") or {}
                    out["summary_" + name_] = {"value": float("%.6g" % sub_["value"]), "unit": sub_.get("unit"),
                                               "ms_per_step": float("%.6g" % sub_["ms_per_step"]) if sub_.get("ms_per_step") else None,
                                               "dominant_kernel": _short(rf_.get("kernel"), 48), "bound": rf_.get("bound"),
                                               "frac": float("%.4g" % rf_["frac"]) if rf_.get("frac") else None,
                                               "traffic": float("%.4g" % rf_["traffic"]) if rf_.get("traffic") else None,
                                               "parity": _pick(sub_.get("parity_sample"), ("queries", "mismatching"))}
                    if sub_.get("queries"):  # a run over a stated number of queries (configs[3] at its size): how many, in how long
                        out["summary_" + name_].update({"queries": sub_["queries"], "seconds": float("%.4g" % sub_["seconds"])})
        out["build_id"] = build_id()
        emit(out, a.detail_out)
    barrier()
    if rank == 0 and not a.fm9 and not a.keep_index:
        for f in (fm9, meta_path):
            try:
                os.remove(f)
            except OSError:
                pass
    if world > 1 or a.gather_single:
        dist.destroy_process_group()


def run_extra_configs(a, fm9):
    """The other configurations of BASELINE.json as compact sub-lines of the default run, so that the driver's own bench record
    carries them: this script re-run once per configuration (fresh process, fresh GPU state; the i.i.d. index file is reused,
    the repeat-bearing genome builds its own), after this process released its index.  Every sub-line keeps value, unit,
    ms_per_step, roofline of its dominant kernel, cpu_baseline and parity_sample; the headline keys are untouched."""
    import subprocess
    t_start = time.time()
    plan = [("hunt_d1_repeats", ["--config", "hunt_d1", "--genome", "repeats", "--steps", "30", "--warmup", "8", "--cpu-seconds", "6", "--parity-queries", "300"], False),
            ("hunt_d2", ["--config", "hunt_d2", "--steps", "30", "--warmup", "8", "--cpu-seconds", "4", "--parity-queries", "300"], True),
            # cap-prone primers: 25-mers at distance 2 — the maxNeighborhood cap can fire, so every strand is
            # enumerated on the host in the reference's order first (nbhd_host.hpp) and searched as explicit patterns
            ("hunt_d2_25mers", ["--config", "hunt_d2", "--qlen", "25", "--queries", "2000", "--steps", "1", "--warmup", "1", "--cpu-seconds", "6",
                                "--parity-queries", "100"], True),
            # what the general kernel k_search serves: Hamming distance 2 (neighbors.h:57-66 without the indel
            # branches) and queries with an N (hunter.h:306-307, util.h:208-219)
            # BASELINE configs[3] at its stated size on one GPU: 10 M distinct 20-mers at edit distance 2, 100 chunks, three in flight
            ("hunt_d2_10M", ["--config", "hunt_d2", "--queries", "10000000", "--scaling", "strong", "--cpu-seconds", "2", "--parity-queries", "300"], True),
            ("hunt_d2_hamming", ["--config", "hunt_d2", "--hamming", "--steps", "30", "--warmup", "8", "--cpu-seconds", "4", "--parity-queries", "300"], True),
            ("hunt_d1_nmix", ["--config", "hunt_d1", "--n-frac", "0.05", "--steps", "30", "--warmup", "8", "--cpu-seconds", "4", "--parity-queries", "300"], True),
            ("search", ["--config", "search", "--steps", "2", "--warmup", "1", "--cpu-seconds", "6"], True),
            ("padlock", ["--config", "padlock", "--steps", "3", "--warmup", "1", "--cpu-seconds", "6"], True)]
    keep = ("metric", "value", "unit", "ms_per_step", "steps", "warmup", "dtype", "roofline", "roofline_search", "cpu_baseline", "parity_sample", "cap_stage",
            "phases_ms", "hits_per_step", "site_stage", "positions_per_s", "value_with_d2h", "seconds", "queries")
    res = {}
    for name, args, reuse in plan:
        left = a.extra_budget_s - (time.time() - t_start)
        if left < 25:
            res[name] = {"skipped": "extra_configs budget of %.0f s spent" % a.extra_budget_s}
            continue
        dpath = fm9 + ".detail_%s.json" % name
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--no-extra-configs", "--no-extras", "--detail-out", dpath] + args
        if reuse:
            cmd += ["--fm9", fm9]
        t0 = time.time()
        try:
            r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=max(30.0, min(left, 240.0 if name == "hunt_d2_10M" else 150.0)))
            if r.returncode != 0 or not os.path.exists(dpath):
                res[name] = {"error": "exit code %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
                continue
            d = json.load(open(dpath))  # the sub-run's full detail (its stdout carries only the compact line)
            os.remove(dpath)
            sub = {k_: d[k_] for k_ in keep if k_ in d}
            sub["workload"] = d["config"]["workload"]
            sub["genome"] = "repeats" if "repeats" in args else "iid"
            sub["wall_s"] = time.time() - t0
            if isinstance(sub.get("cpu_baseline"), dict):
                sub["cpu_baseline"].pop("oracle_ops", None)
            res[name] = sub
        except subprocess.TimeoutExpired:
            res[name] = {"error": "timed out after %.0f s" % (time.time() - t0)}
        except Exception as e:  # a sub-line never takes the headline down
            res[name] = {"error": str(e)[:300]}
    import glob
    for f in glob.glob(fm9 + ".*.meta.json") + glob.glob(fm9 + ".*.npy"):  # the sub-runs' query sets next to the reused index
        if not f.startswith(fm9 + ".hunt_d1."):
            try:
                os.remove(f)
            except OSError:
                pass
    res["wall_s"] = time.time() - t_start
    return res


def cli_end_to_end(fm9, meta, queries, distance, oracle=None, seqlen=None, bytes_per_query=700, extra_env=None):
    """The process seam: `dicey hunt -g <genome> <queries.fa>` on the same queries, wall clock of the whole process (index
    open + derivation, search, JSON for every query).  hunt reads only <genome>.fai and the .fm9 next to the genome.
    queries: list of bytes, or a uint8 array [n, m] (the large runs).  oracle (large runs): three slices of 100 lines of THIS run's
    output are compared byte for byte with the checker's JSON for those queries."""
    import subprocess
    binary = os.path.join(ROOT, "dicey_amd", "dicey")
    if not os.path.exists(binary):
        return {"error": "dicey_amd/dicey not built"}
    d = os.path.dirname(fm9)
    base = os.path.join(d, "dicey_cli_genome_%d.fa" % os.getpid())
    made = []
    nq = len(queries)
    try:
        with open(base + ".gz.fai", "w") as f:
            offs = 0
            for i, ln in enumerate(meta["lens"]):
                f.write("s%d\t%d\t%d\t60\t61\n" % (i, ln, offs))
                offs += ln + ln // 60 + 10
        made.append(base + ".gz.fai")
        open(base + ".gz", "wb").write(b"\x1f\x8b placeholder: hunt reads only the .fai and the .fm9 next to it")
        made.append(base + ".gz")
        os.symlink(fm9, base + ".fm9")
        made.append(base + ".fm9")
        qf = base + ".queries.fa"
        made.append(qf)
        if isinstance(queries, np.ndarray):  # ">q0000000\nACGT...\n" rows assembled as one byte matrix
            m = queries.shape[1]
            rows = np.empty((nq, 10 + m + 1), dtype=np.uint8)
            rows[:, 0] = ord(">")
            rows[:, 1] = ord("q")
            idx = np.arange(nq, dtype=np.int64)
            for k in range(7):
                rows[:, 8 - k] = ord("0") + (idx // 10 ** k) % 10
            rows[:, 9] = 10
            rows[:, 10:10 + m] = queries
            rows[:, 10 + m] = 10
            rows.tofile(qf)
            name_of = lambda i: "q%07d" % i  # noqa: E731
            seq_of = lambda i: queries[i].tobytes().decode()  # noqa: E731
        else:
            with open(qf, "w") as f:
                for i, q in enumerate(queries):
                    f.write(">q%06d\n%s\n" % (i, q.decode()))
            name_of = lambda i: "q%06d" % i  # noqa: E731
            seq_of = lambda i: queries[i].decode()  # noqa: E731
        outp = base + ".out.jsonl"
        made.append(outp)
        try:  # ~650 output bytes per query land in /dev/shm, i.e. in memory: a box short of it gets /dev/null (and no parity slices)
            avail_kb = [int(ln.split()[1]) for ln in open("/proc/meminfo") if ln.startswith("MemAvailable")][0]
            if avail_kb * 1024 < 3 * bytes_per_query * nq + (8 << 30):
                outp = "/dev/null"
                oracle = None
        except Exception:
            pass
        t = time.time()
        with open(outp, "wb") as o:
            r = subprocess.run([binary, "hunt", "-d", str(distance), "-g", base + ".gz", qf], stdout=o, stderr=subprocess.PIPE, timeout=900,
                               env=dict(os.environ, DICEY_TIMING="1", **(extra_env or {})))
        dt = time.time() - t
        out_bytes = os.path.getsize(outp) if outp != "/dev/null" else None
        lines = 0
        checked = None
        if oracle is not None and nq >= 1000:
            want = {}
            for s0 in (0, nq // 2 - 50, nq - 100):
                ids = list(range(s0, s0 + 100))
                js, _ = oracle.hunt(seqlen, ["s%d" % i for i in range(len(seqlen))], [seq_of(i) for i in ids], qnames=[name_of(i) for i in ids],
                                    genome=base + ".gz", distance=distance)
                for i, ln in zip(ids, js.split("\n")[:-1]):
                    want[i] = (ln + "\n").encode()
            bad = 0
            with open(outp, "rb") as f:
                for i, ln in enumerate(f):
                    lines += 1
                    if i in want and ln != want[i]:
                        bad += 1
            checked = {"lines_compared_with_the_checker": len(want), "differing": bad}
        else:
            with open(outp, "rb") as f:
                while True:
                    blk = f.read(1 << 24)
                    if not blk:
                        break
                    lines += blk.count(b"\n")
        phases = {}
        for ln in r.stderr.decode(errors="replace").splitlines():  # "dicey timing: <phase>   <ms> ms" from the library's open
            if ln.startswith("dicey timing:") and ln.rstrip().endswith("ms"):
                name, _, val = ln[len("dicey timing:"):].rstrip()[:-2].rstrip().rpartition(" ")
                try:
                    phases[name.strip()] = float(val)
                except ValueError:
                    pass
        res = {"value": nq / dt, "unit": "primers/s", "seconds": dt, "queries": nq, "exit_code": r.returncode, "json_lines": lines,
               "output_bytes": out_bytes, "index_open_phases_ms": phases,
               "note": "one process on a free GPU: index open + derivation of the HBM layouts, FASTA in, the whole input in chunks with two "
                       "batches in flight, one JSON line per query to a file in /dev/shm"}
        if checked is not None:
            res["parity_slices"] = checked
        return res
    except Exception as e:  # an extra, never fatal for the bench line
        return {"error": str(e)[:200]}
    finally:
        for f in made:
            try:
                os.remove(f)
            except OSError:
                pass


if __name__ == "__main__":
    main()
