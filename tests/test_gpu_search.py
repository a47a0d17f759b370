"""`dicey search` (in-silico PCR) end to end: the repo's binary against the oracle — the restated silica.h driver running
on the REFERENCE's own thal() and JSON number formatting (oracle/_ref, compiled from /root/reference in place)."""
import gzip
import json
import os
import random
import subprocess

import pytest

import oracle_lib as O
from conftest import genome_text, make_genome, revcomp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DICEY = os.path.join(ROOT, "dicey_amd", "dicey")
needs_ref = pytest.mark.skipif(O.ref_libs() is None, reason="oracle/_ref (reference thal.h / json.hpp builds) not present")


def make_pcr(tmp_path_factory):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dicey_amd", "cli"), "-s"])
    d = tmp_path_factory.mktemp("pcr")
    seqs = make_genome(61, 3, 25000)
    names = ["chrA", "chrB", "chrC"]
    fa = d / "genome.fa.gz"
    with gzip.open(fa, "wt") as f:
        for n, s in zip(names, seqs):
            f.write(">%s\n" % n)
            for i in range(0, len(s), 70):
                f.write(s[i:i + 70] + "\n")
    assert subprocess.run([DICEY, "index", str(fa)], capture_output=True).returncode == 0
    rng = random.Random(4)
    rec = []
    while len(rec) < 24:  # primer pairs sampled from the genome, some with one mismatch
        c = rng.randrange(3)
        p = rng.randrange(0, 22000)
        L1, L2, dist = rng.randint(17, 25), rng.randint(17, 25), rng.randint(80, 1500)
        fw, rv = seqs[c][p:p + L1], revcomp(seqs[c][p + dist:p + dist + L2])
        if "N" in fw + rv or len(rv) < L2:
            continue
        if rng.random() < 0.3:
            k = rng.randrange(3, L1 - 3)
            fw = fw[:k] + rng.choice("ACGT") + fw[k + 1:]
        rec += [(">pair%d_f" % len(rec), fw), (">pair%d_r" % len(rec), rv)]
    rec.insert(5, (">tooshort", "ACGTACGTAC"))          # <= k: skipped, and its bases leak into the next record (silica.h:363-389)
    rec.insert(9, (">lower", rec[2][1].lower()))
    rec.append((">withN", seqs[1][3000:3010] + "R" + seqs[1][3011:3024]))
    rec.append((">edge", seqs[0][:21]))
    fasta = "".join("%s\n%s\n" % r for r in rec)
    pf = d / "primers.fa"
    pf.write_text(fasta)
    return {"dir": d, "fa": str(fa), "fm9": str(d / "genome.fa.fm9"), "seqs": seqs, "names": names, "text": genome_text(seqs),
            "seqlen": [len(s) + 1 for s in seqs], "primers": str(pf), "fasta": fasta}


@pytest.fixture(scope="module")
def pcr(tmp_path_factory):
    return make_pcr(tmp_path_factory)


@needs_ref
@pytest.mark.parametrize("extra,kw", [
    ([], {}),
    (["-d", "0", "-c", "40"], dict(distance=0, cutTemp=40.0)),
    (["-n", "-k", "13", "-l", "900", "--cutoffPenalty", "3.5"], dict(hamming=True, kmer=13, maxProdSize=900, cutofPen=3.5)),
    (["-m", "3"], dict(max_locations=3)),
])
def test_search_json_identical_to_oracle(pcr, extra, kw):
    ix = O.Index(pcr["fm9"])
    want, wrc = ix.search(pcr["seqlen"], pcr["names"], pcr["text"], pcr["fasta"], genome=pcr["fa"], **kw)
    r = subprocess.run([DICEY, "search", "-i", O.PRIMER3_CONFIG, "-g", pcr["fa"], *extra, pcr["primers"]], capture_output=True, text=True)
    assert r.returncode == wrc, r.stderr
    assert r.stdout == want
    j = json.loads(r.stdout)
    if not kw:
        assert len(j["data"]["primers"]) >= 15 and len(j["data"]["amplicons"]) >= 3  # the designed pairs are found


@needs_ref
def test_search_sites_library_level(pcr):
    import dicey_amd
    prim = ["".join(ch for ch in s.upper()) for h, s in [ln.split("\n")[:2] for ln in pcr["fasta"].split(">")[1:]] if len(s) > 15][:20]
    prim = [p if set(p) <= set("ACGT") else "".join(c if c in "ACGT" else "N" for c in p) for p in prim]
    with dicey_amd.FmIndex(pcr["fm9"]) as ix:
        th = dicey_amd.Thal(O.PRIMER3_CONFIG)
        sites, mt, fl, nh = dicey_amd.search_sites(ix, th, prim, pcr["seqlen"])
        assert nh >= len(sites) > 0 and all(s["temp"] > 45.0 for s in sites)
        # Tm against the perfect complement equals the reference thal() on (primer, revcomp)
        T, _ = O.ref_libs()
        import ctypes as C
        T.ref_thal.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        t, a, b = C.c_double(), C.c_int(), C.c_int()
        for p, m in zip(prim, mt):
            T.ref_thal(p.encode(), revcomp(p).encode(), C.byref(t), C.byref(a), C.byref(b))
            assert t.value == m
        th.close()


@needs_ref
def test_randomised_search_configurations_against_oracle():
    """tools/fuzz_search.py: random primer sets and option combinations, JSON identical to the oracle's."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_search.py"), "11", "10"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "failing configurations: 0" in r.stdout, r.stdout[-1500:]
