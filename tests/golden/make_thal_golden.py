"""Regenerates tests/golden/thal_vectors.json from the REFERENCE ITSELF: oracle/_ref/libthalref.so is the unmodified
/root/reference/src/thal.h compiled in place (oracle/Makefile).  Doubles are stored as hex so the comparison is exact.
tests/golden/primer3_params.json holds primer3's parameter tables as fixture data (p3config.py writes them out)."""
import ctypes as C, json, os, random, struct
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
R = C.CDLL(os.path.join(ROOT, "oracle", "_ref", "libthalref.so"))
R.ref_thal_init.argtypes = [C.c_char_p] + [C.c_double] * 5
R.ref_thal.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
import sys
sys.path.insert(0, HERE)
import p3config
cfg = p3config.config_dir()
assert R.ref_thal_init(cfg.encode(), 37.0, 50.0, 1.5, 50.0, 0.6) == 0
rng = random.Random(20260929)
def rc(s): return s.translate(str.maketrans("ACGTN", "TGCAN"))[::-1]
def mut(s, k):
    s = list(s)
    for _ in range(k):
        p = rng.randrange(len(s)); r = rng.random()
        if r < .4: s[p] = rng.choice("ACGT")
        elif r < .7 and len(s) > 5: del s[p]
        else: s.insert(p, rng.choice("ACGT"))
    return "".join(s)
pairs = [("GCCCCATAGGTTTTGAACTCA", rc("GCCCCATAGGTTTTGAACTCA"))]  # SURVEY §8(c) known answer: 58.126046031301769, 21/21
for _ in range(1500):
    L = rng.randint(10, 30); p = "".join(rng.choice("ACGT") for _ in range(L)); k = rng.random()
    if k < 0.2: t = rc(p)
    elif k < 0.7: t = rng.choice("ACGT") * rng.randint(0, 3) + mut(rc(p), rng.randint(0, 4)) + "".join(rng.choice("ACGT") for _ in range(rng.randint(0, 3)))
    elif k < 0.8: t = "".join(rng.choice("ACGT") for _ in range(rng.randint(8, 35)))
    elif k < 0.9: t = mut(rc(p), 1).replace("A", "N", 1)
    else:
        h = "".join(rng.choice("ACGT") for _ in range(rng.randint(3, 8))); p = h + rc(h); t = p
    pairs.append((p, t))
out = []
t = C.c_double(); a = C.c_int(); b = C.c_int()
for p, q in pairs:
    ok = R.ref_thal(p.encode(), q.encode(), C.byref(t), C.byref(a), C.byref(b))
    out.append([p, q, struct.pack(">d", t.value).hex(), a.value, b.value, ok])
json.dump({"params": {"temp_c": 37.0, "mv": 50.0, "dv": 1.5, "dna_conc": 50.0, "dntp": 0.6}, "vectors": out},
          open(os.path.join(HERE, "thal_vectors.json"), "w"))
print("written", len(out))

# second set: long oligos (padlock probes are 2 x armlen = 40 nt against their perfect complement; `search` windows reach
# ~35 nt), the 60-nt limit, and other salt / DNA concentrations (RC and the salt correction are host-side constants)
assert R.ref_thal_init(cfg.encode(), 37.0, 40.0, 2.5, 100.0, 0.8) == 0
rng = random.Random(20260930)
pairs = []
for _ in range(700):
    L = rng.choice([30, 36, 40, 40, 44, 48, 52, 60]); p = "".join(rng.choice("ACGT") for _ in range(L)); k = rng.random()
    if k < 0.45: t = rc(p)
    elif k < 0.8: t = mut(rc(p), rng.randint(1, 5))
    elif k < 0.9: t = "".join(rng.choice("ACGT") for _ in range(rng.randint(20, 60)))
    else: t = rc(p)[:rng.randint(15, L)]
    pairs.append((p, t))
pairs += [("AT" * 20, "AT" * 20), ("A" * 40, "T" * 40), ("GC" * 24, "GC" * 24), ("ACGT" * 16, "ACGT" * 16), ("A" * 61, "T" * 61),
          ("ACGT" * 15 + "A", "T" + "ACGT" * 15), ("N" * 30, "N" * 30), ("ACGTN" * 8, "NACGT" * 8)]
out = []
tm_, e1_, e2_ = C.c_double(), C.c_int(), C.c_int()
for p, q in pairs:
    ok = R.ref_thal(p.encode(), q.encode(), C.byref(tm_), C.byref(e1_), C.byref(e2_))
    out.append([p, q, struct.pack(">d", tm_.value).hex(), e1_.value if ok else -1, e2_.value if ok else -1, ok])  # thal() leaves align_end unset when it refuses
json.dump({"params": {"temp_c": 37.0, "mv": 40.0, "dv": 2.5, "dna_conc": 100.0, "dntp": 0.8}, "vectors": out},
          open(os.path.join(HERE, "thal_vectors_long.json"), "w"))
print("written", len(out))
