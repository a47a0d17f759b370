#!/usr/bin/env python3
"""Randomised differential run of dg_hunt against the checker on texts FULL of N runs and sequence boundaries (GPU box): what round 6
added for queries with N (k_nres, k_nkeep, the pruning by the text's shortest N run) and for single-occurrence hits (aligned from
their own codes) sees queries that start / end / straddle N runs and sequence ends, 10-31 nt — shorter and longer than the table
order —, extra edits, lower case, distances 0-2, Hamming mode, forward only, small -m.  The shortest N run of the text is drawn
per seed (1: no pruning at all, 2, 3, 6).
usage: fuzz_n.py [seed] [configurations]   (DICEY_KMER_K / DICEY_KMER_K2 pick the layout)"""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, dicey_amd
from conftest import genome_text, revcomp

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 24
rng = random.Random(seed)
nmin = rng.choice([1, 2, 2, 3, 6])
nseq = rng.choice([3, 8, 30])
seqs = []
for c in range(nseq):
    s = bytearray(rng.choice(b"ACGT") for _ in range(rng.randrange(400, 60000 // nseq + 800)))
    for _ in range(max(2, len(s) // 300)):
        p = rng.randrange(30, len(s) - 60)
        n = rng.choice([nmin, nmin, nmin + 1, rng.randrange(nmin, 40)])
        s[p:p + n] = b"N" * n
    if rng.random() < 0.3:  # a copy of a stretch next to an N run elsewhere (repeats around N)
        a = rng.randrange(30, len(s) - 120); b = rng.randrange(30, len(s) - 120)
        s[b:b + 60] = s[a:a + 60]
    seqs.append(s.decode())
text = genome_text(seqs)
fm9 = "/tmp/fuzz_n_%d.fm9" % seed
O.build_fm9(text, fm9)
orc = O.Index(fm9)
O.fast_neighbors(True)
_lib = None
if "DICEY_LIB" in os.environ:
    from dicey_amd import _capi
    _lib = _capi.load(os.environ["DICEY_LIB"])
ix = dicey_amd.FmIndex(fm9, _lib=_lib, **({"compact": True, "pre5": False} if os.environ.get("FUZZ_ONE_SHOT") else {}))  # FUZZ_ONE_SHOT: the open flags of `dicey hunt`
names = ["c%d" % i for i in range(nseq)]
seqlen = [len(s) + 1 for s in seqs]
bad = 0
for c in range(nconf):
    ham = rng.random() < 0.4
    d = rng.choice([0, 1, 1, 1, 2])
    maxlen = rng.choice([12, 15, 20, 22]) if (d == 2 and not ham) else rng.choice([12, 15, 20, 25, 31])
    kw = dict(distance=d, hamming=ham, forward_only=rng.random() < 0.3, max_locations=rng.choice([1, 2, 5, 1000, 1000]))
    qs = []
    for _ in range(rng.randint(40, 160)):
        s = seqs[rng.randrange(nseq)]
        L = rng.randint(10, maxlen)
        r = rng.random()
        npos = [i for i in range(len(s)) if s[i] == "N" and (s[i - 1] != "N" or (i + 1 < len(s) and s[i + 1] != "N"))]
        if r < 0.55 and npos:     # a window that touches an N run's edge by 1-2 characters, at either end
            e = rng.choice(npos)
            if s[e - 1] != "N": p = e - L + rng.choice([1, 1, 2])        # ends with the run's first N('s)
            else: p = e - rng.choice([0, 0, 1])                           # starts with the run's last N('s)
        elif r < 0.7: p = rng.choice([0, 1, len(s) - L, len(s) - L - 1])  # sequence ends
        else: p = rng.randrange(0, max(1, len(s) - L))
        p = max(0, min(p, len(s) - L))
        q = list(s[p:p + L])
        if rng.random() < 0.15: q = list(revcomp("".join(q)))
        for _ in range(rng.choice([0, 0, 1, 2])):
            k = rng.randrange(len(q)); t = rng.random()
            if t < 0.4: q[k] = rng.choice("ACGTN")
            elif t < 0.7 and len(q) > 10: del q[k]
            else: q.insert(k, rng.choice("ACGT"))
        q = "".join(q)[:maxlen]
        if rng.random() < 0.1: q = q.lower()
        qs.append(q)
    if os.environ.get("FUZZ_ONLY") and c != int(os.environ["FUZZ_ONLY"]):
        continue
    try:
        got = ix.hunt(qs, seqlen, **kw)
    except Exception as e:
        print("conf", c, kw, "library refused:", str(e)[:120]); continue
    _, hits = orc.hunt(seqlen, names, qs, want_hits=True, **kw)
    per = {}
    for h in hits: per.setdefault(h[0], []).append(h[1:])
    mism = [qi for qi, qr in enumerate(got.queries)
            if [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits] != per.get(qi, [])]
    nh = sum(len(q.hits) for q in got.queries)
    print("conf", c, kw, "nmin", nmin, "queries", len(qs), "hits", nh, "MISMATCH %d first %r" % (len(mism), qs[mism[0]]) if mism else "ok")
    if mism:
        qi = mism[0]
        print("    got ", [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in got.queries[qi].hits][:4])
        print("    want", per.get(qi, [])[:4])
        alone = ix.hunt([qs[qi]], seqlen, **kw)
        print("    alone", [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in alone.queries[0].hits][:4], "lens in batch", sorted({len(q) for q in qs}))
    bad += bool(mism)
print("failing configurations:", bad)
