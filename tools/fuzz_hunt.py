#!/usr/bin/env python3
"""Randomised differential run of dg_hunt against the oracle (GPU box): random parameter combinations, query lengths, edits,
repeats and non-DNA letters on a small repeat-rich genome.  Prints the first mismatch of every failing configuration."""
import os, random, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O, dicey_amd
from conftest import make_genome, genome_text, revcomp

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 40
rng = random.Random(seed)
seqs = make_genome(500 + seed, 4, 20000, iupac=True)
# make it repeat-rich: copy segments around, with and without small edits
seqs = [list(s) for s in seqs]
for _ in range(60):
    a, b = rng.randrange(4), rng.randrange(4)
    p, q, L = rng.randrange(18000), rng.randrange(18000), rng.randint(15, 300)
    seg = seqs[a][p:p + L]
    if rng.random() < 0.5 and len(seg) > 5:
        k = rng.randrange(len(seg)); seg = seg[:k] + [rng.choice("ACGT")] + seg[k + 1:]
    seqs[b][q:q + len(seg)] = seg
seqs = ["".join(s) for s in seqs]
text = genome_text(seqs)
fm9 = "/tmp/fuzz_%d.fm9" % seed
O.build_fm9(text, fm9)
orc = O.Index(fm9)
if os.environ.get("FUZZ_FAST_NEIGHBORS"):  # the checker's hash-set neighbourhoods (tested equal to the literal ones): d=2 configurations in seconds
    O.fast_neighbors(True)
_lib = None
if "DICEY_LIB" in os.environ:  # e.g. tools/hostemu/libdiceygpu_hostemu.so for a GPU-less sweep
    from dicey_amd import _capi
    _lib = _capi.load(os.environ["DICEY_LIB"])
ix = dicey_amd.FmIndex(fm9, _lib=_lib, **({"compact": True, "pre5": False} if os.environ.get("FUZZ_ONE_SHOT") else {}))  # FUZZ_ONE_SHOT: the open flags of `dicey hunt`
names = ["c%d" % i for i in range(4)]
seqlen = [len(s) + 1 for s in seqs]
bad = 0
for c in range(nconf):
    ham = rng.random() < 0.4
    d = rng.choice([0, 1, 1, 1, 2])
    maxlen = (rng.choice([14, 20, 20, 22]) if os.environ.get("FUZZ_FAST_NEIGHBORS") else 20) if (d == 2 and not ham) else rng.choice([12, 20, 31, 40, 47])
    if d == 2 and ham: maxlen = min(maxlen, 47)
    kw = dict(distance=d, hamming=ham, forward_only=rng.random() < 0.3, max_locations=rng.choice([1, 2, 5, 1000, 1000]))
    qs = []
    for _ in range(rng.randint(30, 150)):
        L = rng.randint(10, maxlen); cidx = rng.randrange(4); p = rng.randrange(0, 20000 - L)
        r = rng.random()
        if r < 0.7: q = seqs[cidx][p:p + L]
        elif r < 0.8: q = "".join(rng.choice("ACGT") for _ in range(L))
        elif r < 0.9: q = rng.choice("ACGT") * L
        else: q = revcomp(seqs[cidx][p:p + L])
        q = list(q)
        for _ in range(rng.choice([0, 0, 1, 2])):
            if not q: break
            k = rng.randrange(len(q)); t = rng.random()
            if t < 0.4: q[k] = rng.choice("ACGTN")
            elif t < 0.7 and len(q) > 10: del q[k]
            else: q.insert(k, rng.choice("ACGT"))
        q = "".join(q)[:maxlen]
        if rng.random() < 0.1: q = q.lower()
        qs.append(q)
    if os.environ.get("FUZZ_ONLY") and c != int(os.environ["FUZZ_ONLY"]):  # (the generator has advanced as in a full run)
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
    print("conf", c, kw, "queries", len(qs), "hits", nh, "MISMATCH %d first %r" % (len(mism), qs[mism[0]]) if mism else "ok")
    bad += bool(mism)
    if mism and os.environ.get("FUZZ_VERBOSE"):
        for qi in mism[:3]:
            print("  query", qi, repr(qs[qi]), "flags", got.queries[qi].flags)
            print("    got ", [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in got.queries[qi].hits][:6])
            print("    want", per.get(qi, [])[:6])
            alone = ix.hunt([qs[qi]], seqlen, **kw)
            print("    alone", [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in alone.queries[0].hits][:6])
print("failing configurations:", bad)
