#!/usr/bin/env python3
"""Scale check of `dicey padlock` (BASELINE.json configs[4] shape: 1000 genes) — not the headline bench.  Exons are cut
from the same synthetic GRCh38-size genome as bench.py; dg_padlock_scan returns the per-position values for all of them
(thal of arms and probes, exact and neighbourhood counts of surviving arms).  CPU beside it: the oracle's restated
padlock.h loop calling the reference's own thal.h on a bounded sample of the same exons, whose accepted probe positions
must equal a replay of the reference's decisions on the GPU arrays."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import bench, dicey_amd, p3config
from dicey_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--genome-size", type=float, default=3.1e9)
ap.add_argument("--genes", type=int, default=1000)
ap.add_argument("--fm9", default="")
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--cpu-genes", type=int, default=3)
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = _capi.load()
t0 = time.time()
text, lens = bench.synth_genome(int(a.genome_size), 24, seed=1, device=dev)
fm9 = a.fm9 or "/dev/shm/dicey_padlock_bench.fm9"
if not a.fm9:
    _capi.check(L, L.dg_index_build_device(C.c_void_p(text.data_ptr()), text.numel(), 0, fm9.encode()))
rng = np.random.default_rng(45)
n = text.numel()
comp = bytes.maketrans(b"ACGT", b"TGCA")
exons, gene_of = [], []
for g in range(a.genes):  # a gene: 4-12 exons of 100-600 nt inside a 100 kb window, one strand
    base = int(rng.integers(0, n - 200000)); strand = int(rng.integers(0, 2))
    for _ in range(int(rng.integers(4, 13))):
        p = base + int(rng.integers(0, 100000)); l = int(rng.integers(100, 601))
        s = bytes(text[p:p + l].cpu().numpy().tobytes())
        if b"\n" in s: continue
        exons.append(s.translate(comp)[::-1] if strand else s); gene_of.append(g)
del text; torch.cuda.empty_cache()
ix = dicey_amd.FmIndex(fm9)
th = dicey_amd.Thal(p3config.config_dir())
t1 = time.time(); R = dicey_amd.padlock_scan(ix, th, exons); dt1 = time.time() - t1
t1 = time.time(); R = dicey_amd.padlock_scan(ix, th, exons); dt = time.time() - t1
npos = int(R["pos_off"][-1])

def replay(e, armlen=20, mingc=0.4, maxgc=0.6, tmdiff=2, max_nb=2, exp=1):
    """padlock.h:321-428 + :506 on the arrays (arm mode, edit distance 1, non-overlapping); spacer/barcode GC left out"""
    o = int(R["pos_off"][e]); ln = len(exons[e]); acc = []
    k = 0
    while k < ln - 2 * armlen + 1:
        g1, g2, pg = R["arm_gc"][o + k], R["arm_gc"][o + k + armlen], R["probe_gc"][o + k]
        ok = mingc <= g1 <= maxgc and not (R["arm_tm"][o + k] > 93 + g1 - 675.0 / armlen) and mingc <= g2 <= maxgc
        ok = ok and not (R["arm_tm"][o + k + armlen] > 93 + g2 - 675.0 / armlen) and abs(R["arm_tm"][o + k] - R["arm_tm"][o + k + armlen]) <= tmdiff
        ok = ok and mingc <= pg <= maxgc
        if ok:
            lo = 81.5 + pg - 675.0 / (2 * armlen); ok = lo <= R["probe_tm"][o + k] <= lo + 10
        ok = ok and R["arm_count"][o + k] <= exp and R["arm_count"][o + k + armlen] <= exp
        ok = ok and R["arm_nbcount"][o + k] <= max_nb and R["arm_nbcount"][o + k + armlen] <= max_nb
        if ok:
            acc.append(k); k += 2 * armlen - 1
        k += 1
    return acc

cpu = None
if not a.no_cpu:
    try:
        import oracle_lib as O
        if O.ref_libs() is None: raise RuntimeError("oracle/_ref not built")
        orc = O.Index(fm9)
        sel = [e for e in range(len(exons)) if gene_of[e] < a.cpu_genes]
        names = ["e%d" % e for e in sel]
        bars = "".join(">%d\nACGTTGCAACGTTGCAACGT\n" % i for i in range(len(sel)))
        tc = time.time()
        tsv, js, err, rc = orc.padlock(names, [exons[e].decode() for e in sel], "", bars, input_fasta=True, spacerleft="GC", spacerright="GC",
                                       anchor="GCGCGCATATGCGCGCATAT")
        dtc = time.time() - tc
        rows = [ln.split("\t") for ln in tsv.rstrip("\n").split("\n")[1:]]
        want = sorted((r[0], int(r[3].split(":")[1]) - 1) for r in rows)
        got = sorted(("e%d" % e, k) for e in sel for k in replay(e))
        pos = sum(len(exons[e]) for e in sel)
        cpu = {"value": pos / dtc, "unit": "exon positions/s", "cores": 1, "kind": "reference",
               "sample": f"{len(sel)} exons of {a.cpu_genes} genes ({pos} positions), restated padlock.h:321-520 calling the reference thal.h, {dtc:.1f} s",
               "accepted_probes": len(want), "gpu_replay_identical": want == got}
    except Exception as ex:
        cpu = {"error": str(ex)}
print(json.dumps({"cpu_baseline": cpu, "workload": f"dicey padlock scan, {a.genes} genes, {len(exons)} exons, {npos} arm windows, armlen 20, d=1, genome {int(a.genome_size)}",
                  "seconds_first": dt1, "seconds": dt, "positions_per_s": npos / dt, "genes_per_s": a.genes / dt, "arm_thal": int(R["n_arm_thal"]),
                  "probe_thal": int(R["n_probe_thal"]), "arms_counted": int(R["n_arms_counted"]), "setup_s": t1 - t0}))
if not a.fm9: os.remove(fm9)
