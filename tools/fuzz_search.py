#!/usr/bin/env python3
"""Randomised differential run of `dicey search` (the repo's binary) against the oracle (restated silica.h + the reference's
own thal.h, oracle/_ref) on a small repeat-rich genome: random primer sets and option combinations; JSON must be identical."""
import gzip, os, random, subprocess, sys, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O
from conftest import make_genome, genome_text, revcomp

seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nconf = int(sys.argv[2]) if len(sys.argv) > 2 else 12
rng = random.Random(seed)
DICEY = os.environ.get("DICEY_BIN", os.path.join(ROOT, "dicey_amd", "dicey"))  # DICEY_BIN: e.g. a tools/hostemu-linked binary
d = tempfile.mkdtemp(prefix="fuzz_search_")
seqs = [list(s) for s in make_genome(900 + seed, 3, 30000)]
for _ in range(40):  # repeats, some with an edit
    a, b = rng.randrange(3), rng.randrange(3)
    p, q, L = rng.randrange(28000), rng.randrange(28000), rng.randint(20, 400)
    seg = seqs[a][p:p + L]
    if rng.random() < 0.5:
        k = rng.randrange(len(seg)); seg = seg[:k] + [rng.choice("ACGT")] + seg[k + 1:]
    seqs[b][q:q + len(seg)] = seg
seqs = ["".join(s) for s in seqs]
names = ["chrA", "chrB", "chrC"]
fa = os.path.join(d, "g.fa.gz")
with gzip.open(fa, "wt") as f:
    for n, s in zip(names, seqs):
        f.write(">%s\n" % n)
        for i in range(0, len(s), 70): f.write(s[i:i + 70] + "\n")
text = genome_text(seqs)
if "DICEY_BIN" in os.environ: O.build_fm9(text, os.path.join(d, "g.fa.fm9"))  # the emulator build has no index builder
else: assert subprocess.run([DICEY, "index", fa], capture_output=True).returncode == 0
orc = O.Index(os.path.join(d, "g.fa.fm9"))
seqlen = [len(s) + 1 for s in seqs]
bad = 0
for c in range(nconf):
    rec = []
    while len(rec) < rng.randint(6, 30):
        ci, p = rng.randrange(3), rng.randrange(0, 27000)
        L1, L2, dist = rng.randint(16, 30), rng.randint(16, 30), rng.randint(60, 2500)
        fw, rv = seqs[ci][p:p + L1], revcomp(seqs[ci][p + dist:p + dist + L2])
        if "N" in fw + rv or len(rv) < L2: continue
        for k_ in range(rng.choice([0, 0, 1, 2])):
            k = rng.randrange(2, L1 - 2); fw = fw[:k] + rng.choice("ACGT") + fw[k + 1:]
        if rng.random() < 0.1: fw = fw.lower()
        rec += [(">p%d_f" % len(rec), fw), (">p%d_r" % len(rec), rv)]
    if rng.random() < 0.3: rec.insert(rng.randrange(len(rec)), (">short", "ACGTACGTACG"))
    fasta = "".join("%s\n%s\n" % r for r in rec)
    pf = os.path.join(d, "p%d.fa" % c)
    open(pf, "w").write(fasta)
    args, kw = [], {}
    if rng.random() < 0.4: args += ["-n"]; kw["hamming"] = True
    if rng.random() < 0.5:
        dd = rng.choice([0, 1]); args += ["-d", str(dd)]; kw["distance"] = dd
    if rng.random() < 0.6:
        k = rng.randint(11, 16); args += ["-k", str(k)]; kw["kmer"] = k
    if rng.random() < 0.5:
        ct = rng.choice([35.0, 40.0, 50.0, 55.0]); args += ["-c", str(ct)]; kw["cutTemp"] = ct
    if rng.random() < 0.3:
        m = rng.choice([1, 3, 20]); args += ["-m", str(m)]; kw["max_locations"] = m
    if rng.random() < 0.4:
        l = rng.choice([300, 1000, 5000]); args += ["-l", str(l)]; kw["maxProdSize"] = l
    if rng.random() < 0.3:
        cp = rng.choice([1.0, 3.5]); args += ["--cutoffPenalty", str(cp)]; kw["cutofPen"] = cp
    want, wrc = orc.search(seqlen, names, text, fasta, genome=fa, **kw)
    r = subprocess.run([DICEY, "search", "-i", O.PRIMER3_CONFIG, "-g", fa, *args, pf], capture_output=True, text=True, timeout=600)
    ok = r.returncode == wrc and r.stdout == want
    print("conf", c, args, "primers", len(rec), "json bytes", len(want), "ok" if ok else "MISMATCH rc %d/%d %s" % (r.returncode, wrc, r.stderr[-200:]))
    if not ok:
        bad += 1
        for i, (x, y) in enumerate(zip(r.stdout, want)):
            if x != y:
                print("  first difference at byte", i, repr(r.stdout[max(0, i - 80):i + 80]), "\n  want", repr(want[max(0, i - 80):i + 80])); break
print("failing configurations:", bad)
