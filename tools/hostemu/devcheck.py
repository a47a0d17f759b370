"""DEVELOPMENT TOOL: run the kernel bodies under host emulation and compare with the oracle.
Not a test of the product (the product is the gfx950 build); used to debug logic without a GPU."""
import os, random, sys, time
os.environ.setdefault("DICEY_NO_BLOCK_LOCATE", "1")  # the workgroup-cooperative locate needs barriers, which the emulator lacks
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
import dicey_amd
from dicey_amd import _capi

EMU = _capi.load(os.path.join(ROOT, "tools/hostemu/libdiceygpu_hostemu.so"))

def genome(rng, nchr, L, nrate=0.002, repeat=True):
    seqs = []
    for c in range(nchr):
        s = []
        while len(s) < L:
            r = rng.random()
            if r < nrate: s.extend('N' * rng.randint(1, 200))
            elif repeat and r < nrate + 0.001 and len(s) > 200:
                a = rng.randrange(len(s) - 100); s.extend(s[a:a + rng.randint(20, 100)])
            elif repeat and r < nrate + 0.0015: s.extend(rng.choice('ACGT') * rng.randint(5, 40))
            else: s.append(rng.choice('ACGT'))
        seqs.append(''.join(s[:L]))
    return seqs

def mutate(rng, q):
    r = rng.random(); p = rng.randrange(len(q))
    if r < 0.33: return q[:p] + rng.choice('ACGT') + q[p+1:]
    if r < 0.66: return q[:p] + q[p+1:]
    return q[:p] + rng.choice('ACGT') + q[p:]

def queries(rng, text, n, lens=(20,)):
    t = text.decode(); qs = []
    while len(qs) < n:
        m = rng.choice(lens); r = rng.random()
        if r < 0.8:
            p = rng.randrange(len(t) - m); q = t[p:p+m]
            if '\n' in q: continue
            if rng.random() < 0.5: q = mutate(rng, q)
            if rng.random() < 0.3: q = O_rc(q)
        else: q = ''.join(rng.choice('ACGT') for _ in range(m))
        if rng.random() < 0.05: q = q.lower()
        qs.append(q)
    return qs

def O_rc(s):
    return s.upper().translate(str.maketrans('ACGTN', 'TGCAN'))[::-1]

def compare(ix_emu, ix_orc, seqlen, names, qs, **kw):
    t = time.time(); res = ix_emu.hunt(qs, seqlen, **kw); te = time.time() - t
    t = time.time(); js, hits = ix_orc.hunt(seqlen, names, qs, want_hits=True, **kw); to = time.time() - t
    per = {}
    for h in hits: per.setdefault(h[0], []).append(h[1:])
    bad = 0
    for qi, qr in enumerate(res.queries):
        a = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
        b = per.get(qi, [])
        if a != b:
            bad += 1
            if bad <= 3:
                print("MISMATCH q", qi, qs[qi], kw); print("  emu:", a[:6]); print("  orc:", b[:6])
    return bad, te, to, res

if __name__ == "__main__":
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 7
    rng = random.Random(seed)
    seqs = genome(rng, 3, 40000)
    text = ("\n".join(seqs) + "\n").encode()
    names = ["chr%d" % i for i in range(len(seqs))]; seqlen = [len(s) + 1 for s in seqs]
    fm9 = "/tmp/devcheck_%d.fm9" % seed
    O.build_fm9(text, fm9)
    orc = O.Index(fm9); emu = dicey_amd.FmIndex(fm9, _lib=EMU)
    tot = 0
    for kw, nq, lens in [(dict(distance=1), 300, (20,)), (dict(distance=0), 100, (18, 20)), (dict(distance=1, hamming=True), 200, (20, 15)),
                         (dict(distance=2, hamming=True), 100, (20,)), (dict(distance=1, forward_only=True), 100, (12, 25)),
                         (dict(distance=1, max_locations=3), 200, (10, 11, 12))]:
        qs = queries(rng, text, nq, lens)
        bad, te, to, res = compare(emu, orc, seqlen, names, qs, **kw)
        print(kw, "queries", nq, "mismatching", bad, "emu %.2fs orc %.2fs" % (te, to), res.counters)
        tot += bad
    print("TOTAL MISMATCHES", tot)
