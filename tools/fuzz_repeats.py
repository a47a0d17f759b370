#!/usr/bin/env python3
"""Randomised differential run of dg_hunt against the checker on REPEAT-RICH genomes whose N runs are long (GPU box): what round 6
added — the locate job kernels on {position, context} records, prefix levels, the bucket sort, k_nres / k_nkeep for queries with
N — sees random copy numbers (tens to tens of thousands), random -m, lengths 15-29, distances 0-2, Hamming mode, N's anywhere
in the query (inside, at the ends, two of them), copies at sequence ends and next to N runs.
usage: fuzz_repeats.py [seed] [configurations]; prints the first mismatch of every failing configuration."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, dicey_amd
from conftest import genome_text, revcomp

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = random.Random(seed)
nseq = rng.choice([1, 2, 3, 5])
size = rng.choice([600_000, 1_500_000, 4_000_000])
seqs = [bytearray(rng.choice(b"ACGT") for _ in range(size // nseq)) for _ in range(nseq)]
fams = []
for _ in range(rng.randint(2, 5)):
    L = rng.choice([24, 40, 60, 120])
    unit = "".join(rng.choice("ACGT") for _ in range(L))
    copies = rng.choice([30, 300, 2000, 12000, 40000])
    div = rng.choice([0.0, 0.02, 0.1])
    for _ in range(copies):
        s = seqs[rng.randrange(nseq)]
        p = rng.randrange(0, len(s) - L)
        u = list(unit)
        if div:
            for k in range(L):
                if rng.random() < div: u[k] = rng.choice("ACGT")
        s[p:p + L] = "".join(u).encode()
    for s in seqs:   # a copy at either end of every sequence
        s[:L] = unit.encode(); s[-L:] = unit.encode()
    fams.append(unit)
for s in seqs:       # N runs of at least 12, some of them flush against a family copy
    for _ in range(rng.randint(2, 6)):
        p = rng.randrange(1000, len(s) - 1000)
        n = rng.randint(12, 400)
        s[p:p + n] = b"N" * n
    for _ in range(4):
        p = rng.randrange(1000, len(s) - 1000)
        n = rng.randint(12, 60)
        s[p:p + n] = b"N" * n
        u = rng.choice(fams)
        if rng.random() < 0.5: s[p + n:p + n + len(u)] = u.encode()
        else: s[p - len(u):p] = u.encode()
for s in seqs:       # (family copies planted over a run may have cut it short: no run below 12)
    i = 0
    while True:
        i = s.find(b"N", i)
        if i < 0: break
        j = i
        while j < len(s) and s[j] == 78: j += 1
        if j - i < 12: s[i:min(len(s), i + 12)] = b"N" * (min(len(s), i + 12) - i); j = i + 12
        i = j
seqs = [s.decode() for s in seqs]
text = genome_text(seqs)
fm9 = "/tmp/fuzz_rep_%d.fm9" % seed
dicey_amd.build_index(text, fm9, device=0)
orc = O.Index(fm9)
O.fast_neighbors(True)
ix = dicey_amd.FmIndex(fm9)
names = ["c%d" % i for i in range(nseq)]
seqlen = [len(s) + 1 for s in seqs]
bad = 0
for c in range(nconf):
    ham = rng.random() < 0.3
    d = rng.choice([0, 1, 1, 1, 2])
    maxlen = 22 if (d == 2 and not ham) else rng.choice([20, 24, 29])
    kw = dict(distance=d, hamming=ham, forward_only=rng.random() < 0.2, max_locations=rng.choice([1, 7, 100, 1000, 1000, 1000]))
    qs = []
    for _ in range(rng.randint(40, 160)):
        L = rng.randint(15, maxlen)
        r = rng.random()
        if r < 0.6:
            u = rng.choice(fams); o = rng.randrange(0, len(u) - L + 1) if len(u) >= L else 0; q = u[o:o + L]
        elif r < 0.85:
            s = rng.choice(seqs); p = rng.randrange(0, len(s) - L); q = s[p:p + L]
        else: q = "".join(rng.choice("ACGT") for _ in range(L))
        if rng.random() < 0.3: q = revcomp(q)
        q = list(q)
        for _ in range(rng.choice([0, 0, 1, 1, 2])):
            if len(q) < 12: break
            k = rng.randrange(len(q)); t = rng.random()
            if t < 0.4: q[k] = rng.choice("ACGT")
            elif t < 0.6: del q[k]
            elif t < 0.8: q.insert(k, rng.choice("ACGT"))
            else: q[rng.choice([0, 1, k, k, len(q) - 2, len(q) - 1])] = "N"
        qs.append("".join(q)[:maxlen])
    try:
        got = ix.hunt(qs, seqlen, **kw)
    except Exception as e:
        print("conf", c, kw, "library refused:", str(e)[:160]); bad += 1; continue
    _, hits = orc.hunt(seqlen, names, qs, want_hits=True, **kw)
    per = {}
    for h in hits: per.setdefault(h[0], []).append(h[1:])
    mism = [qi for qi, qr in enumerate(got.queries)
            if [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits] != per.get(qi, [])]
    nh = sum(len(q.hits) for q in got.queries)
    print("conf", c, kw, "queries", len(qs), "hits", nh, "MISMATCH %d first %r" % (len(mism), qs[mism[0]]) if mism else "ok")
    bad += bool(mism)
print("genome", size, "sequences", nseq, "families", [(len(u)) for u in fams])
print("failing configurations:", bad)
os.remove(fm9)
