#!/usr/bin/env python3
"""Scale check of `dicey search` binding-site discovery (BASELINE.json configs[2] shape: primer pairs on a GRCh38-size
genome) — not the headline bench.  Builds the same synthetic genome/index as bench.py, samples primer pairs from it and
times dg_search_sites (FM search of the 15-mer neighbourhoods + one thal() per located hit, all on the GPU)."""
import argparse, ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench, dicey_amd
from dicey_amd import _capi

ap = argparse.ArgumentParser()
ap.add_argument("--genome-size", type=float, default=3.1e9)
ap.add_argument("--pairs", type=int, default=2000)
ap.add_argument("--dump", default="", help="write the site list (JSON lines) here, to diff two runs")
ap.add_argument("--no-cpu", action="store_true")
ap.add_argument("--fm9", default="", help="reuse an index of the same synthetic genome (bench.py --keep-index)")
a = ap.parse_args()
dev = torch.device("cuda", 0)
L = _capi.load()
t0 = time.time()
text, lens = bench.synth_genome(int(a.genome_size), 24, seed=1, device=dev)
fm9 = a.fm9 or "/dev/shm/dicey_search_bench.fm9"
if not a.fm9:
    _capi.check(L, L.dg_index_build_device(C.c_void_p(text.data_ptr()), text.numel(), 0, fm9.encode()))
rng = np.random.default_rng(43)
prim = []
n = text.numel()
comp = bytes.maketrans(b"ACGT", b"TGCA")
while len(prim) < 2 * a.pairs:
    p = int(rng.integers(0, n - 3000)); l1, l2, d = int(rng.integers(18, 26)), int(rng.integers(18, 26)), int(rng.integers(80, 2000))
    w = bytes(text[p:p + d + l2 + 1].cpu().numpy().tobytes())
    fw, rv = w[:l1], w[d:d + l2]
    if b"N" in fw + rv or b"\n" in w: continue
    prim += [fw.decode(), rv.translate(comp)[::-1].decode()]
host_text = text.cpu().numpy().tobytes()  # the oracle's amplicon sequences come from the text
del text; torch.cuda.empty_cache()
seqlen = [x + 1 for x in lens]
ix = dicey_amd.FmIndex(fm9); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import p3config
th = dicey_amd.Thal(p3config.config_dir())
t1 = time.time()
sites, mt, fl, nh = dicey_amd.search_sites(ix, th, prim, seqlen)
dt = time.time() - t1
t1 = time.time()
sites, mt, fl, nh = dicey_amd.search_sites(ix, th, prim, seqlen)
dt2 = time.time() - t1
# CPU baseline: the oracle's restated silica.h driver on the REFERENCE's own thal() (oracle/_ref), one thread, bounded sample
cpu = None
if a.dump:
    with open(a.dump, "w") as f:
        for s_ in sites:
            s_ = dict(s_); s_["temp"] = float(s_["temp"]).hex(); s_["perf_temp"] = float(s_["perf_temp"]).hex(); s_["pseq"] = prim[s_["primer"]]
            f.write(json.dumps(s_, sort_keys=True) + "\n")
try:
    if a.no_cpu: raise RuntimeError("skipped")
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O
    if O.ref_libs() is not None:
        orc = O.Index(fm9)
        ns = 40
        fa = "".join(">p%d\n%s\n" % (i, s) for i, s in enumerate(prim[:ns]))
        tc = time.time()
        js, rc = orc.search(seqlen, ["s%d" % i for i in range(len(seqlen))], host_text, fa)
        dtc = time.time() - tc
        orc.close()
        nthal = sum(1 for _ in ())  # thal calls are not counted by the oracle; hits per primer are the same as on the GPU
        cpu = {"value": ns / dtc, "unit": "primers/s", "cores": 1, "kind": "reference",
               "sample": f"first {ns} primers, restated silica.h:429-573 calling the reference thal.h (oracle/_ref), {dtc:.1f} s"}
except Exception as e:  # the checker is optional here
    cpu = {"error": str(e)}
if hasattr(L, "dg_debug_wave_profile"):
    buf = (C.c_ulonglong * 8)(); L.dg_debug_wave_profile(buf)
    v = list(buf); np_ = max(1, v[5])
    print("PROFILE per pair (cycles): rowpass %.0f openings %.0f reduce+write %.0f end+trace %.0f | rounds/pair %.1f | whole hit %.0f | pairs %d" % (v[0]/np_, v[1]/np_, v[2]/np_, v[3]/np_, v[4]/np_, v[6]/np_, v[5]), file=sys.stderr)
print(json.dumps({"cpu_baseline": cpu, "workload": f"dicey search sites, {len(prim)} primers (18-25 nt), k=15, d=1, genome {int(a.genome_size)}",
                  "seconds_first": dt, "seconds": dt2, "primers_per_s": len(prim) / dt2, "thal_calls": nh, "thal_per_s": nh / dt2,
                  "sites": len(sites), "ms_device": dicey_amd.search_sites.last_ms_device, "setup_s": t1 - t0}))
if not a.fm9:
    os.remove(fm9)
