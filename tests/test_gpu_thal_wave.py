"""The wave-per-pair thal kernel (thal_wave.hpp) against the sequential lane-per-pair kernel (thal.hpp) — the latter is
what the emulator and the golden vectors pin to the reference.  The library picks its kernels once per process, so the
sequential side runs in a child process with DICEY_DEBUG_THAL_REDO=1 (every pair / hit is handed back and recomputed)."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu

CHILD = r'''
import json, os, random, sys
sys.path.insert(0, %(root)r); sys.path.insert(0, os.path.join(%(root)r, "tests")); sys.path.insert(0, os.path.join(%(root)r, "tests", "golden"))
import dicey_amd, p3config
from conftest import make_genome, genome_text, revcomp
rng = random.Random(11)
th = dicey_amd.Thal(p3config.config_dir())
pairs = []
for _ in range(6000):
    l1, l2 = rng.randint(1, 48), rng.randint(1, 48)
    a = "".join(rng.choice("ACGT") for _ in range(l1))
    if rng.random() < 0.6:   # mostly near-complementary pairs: long duplexes with loops and bulges
        b = list(revcomp(a))
        for _ in range(rng.randint(0, 4)):
            k = rng.randrange(len(b)); r = rng.random()
            if r < 0.4: b[k] = rng.choice("ACGTN")
            elif r < 0.7: b.insert(k, rng.choice("ACGT"))
            elif len(b) > 1: del b[k]
        b = "".join(b)[:48]
    else:
        b = "".join(rng.choice("ACGTN") for _ in range(l2))
    pairs.append((a, b))
pairs += [("ACGT" * 6, "ACGT" * 6), ("A" * 30, "T" * 30), ("GC" * 12, "GC" * 12), ("N" * 10, "ACGTACGTAC"), ("A", "T"), ("ACGTTGCA" * 7, "TGCAACGT" * 7)]
res = th.tm(pairs)
out = {"thal": [[float(t).hex(), int(e1), int(e2)] for t, e1, e2 in res]}
seqs = make_genome(77, 4, 400000)
text = genome_text(seqs)
fm9 = os.path.join(%(tmp)r, "g.fm9")
if not os.path.exists(fm9): dicey_amd.build_index(text if isinstance(text, bytes) else text.encode(), fm9)
prim = []
while len(prim) < 300:
    c = rng.randrange(4); p = rng.randrange(0, 390000); L = rng.randint(16, 27)
    s = seqs[c][p:p + L]
    if "N" in s or len(s) < L: continue
    if rng.random() < 0.5: s = revcomp(s)
    prim.append(s)
with dicey_amd.FmIndex(fm9) as ix:
    for kw in (dict(), dict(hamming=True, kmer=13, cut_temp=35.0), dict(distance=0, cut_temp=30.0)):
        sites, mt, fl, nh = dicey_amd.search_sites(ix, th, prim, [len(s) + 1 for s in seqs], **kw)
        out["sites_%%s" %% sorted(kw.items())] = [[s["ref"], s["pos"], s["primer"], s["on_for"], float(s["temp"]).hex(), s["genome"]] for s in sites] + [nh, fl]
th.close()
json.dump(out, open(sys.argv[1], "w"))
'''


def run_child(tmp, name, env_extra):
    out = os.path.join(tmp, name + ".json")
    env = dict(os.environ, **env_extra)
    r = subprocess.run([sys.executable, "-c", CHILD % {"root": ROOT, "tmp": tmp}, out], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    return json.load(open(out))


def test_wave_kernel_equals_sequential_kernel(tmp_path):
    tmp = str(tmp_path)
    wave = run_child(tmp, "wave", {})
    from conftest import build_exp_lib
    seq = run_child(tmp, "seq", {"DICEY_DEBUG_THAL_REDO": "1", "DICEY_LIB": build_exp_lib()})  # (a test switch: the development build reads it)
    assert wave.keys() == seq.keys()
    for k in wave:
        assert wave[k] == seq[k], k
    assert len(wave["thal"]) > 6000 and all(len(v) > 50 for k, v in wave.items() if k.startswith("sites"))
