"""GPU parity for inputs where neighbors()' size cap matters (reference src/neighbors.h:50, hunter.h:342-345): the
library answers them like the reference instead of refusing.  Hits in push order AND the per-query message vector are
compared with the oracle.  (Development: DICEY_LIB=<emulator build> runs the same checks without a GPU; the marked
tests themselves always load the gfx950 library.)"""
import json
import os
import random

import pytest

try:
    import torch
    torch.cuda.is_available()
except Exception:  # pragma: no cover
    torch = None

import oracle_lib as O
from conftest import revcomp

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ix(small_genome):
    import dicey_amd
    from conftest import exp_lib
    # (the development build: DICEY_CAP_HOST — read per batch — is a test switch the product library ignores, experiments.hpp)
    h = dicey_amd.FmIndex(small_genome["fm9"], device=0, _lib=exp_lib())
    yield h
    h.close()


@pytest.fixture(params=["device", "host"], autouse=True)
def cap_path(request, monkeypatch):
    """r04: capped neighbourhoods of A/C/G/T sequences up to 29 nt (edit distance <= 2) are enumerated on the device (k_cap_enum);
    DICEY_CAP_HOST keeps every one of them on the host enumerator (nbhd_host.hpp), which also serves the rest (N, longer, Hamming).
    Every test of this module runs both ways."""
    if request.param == "host":
        monkeypatch.setenv("DICEY_CAP_HOST", "1")
    return request.param


def compare(ix, g, qs, **kw):
    """hits (push order) and messages of every query against the oracle's restated hunter.h loop"""
    got = ix.hunt(qs, g["seqlen"], **kw)
    orc = O.Index(g["fm9"])
    lines, per = orc.hunt_parallel(g["seqlen"], g["names"], qs, workers=min(32, os.cpu_count() or 1), **kw)  # (r04: one thread, 115 s a test)
    assert len(lines) == len(qs) and all(ln is not None for ln in lines)
    ml, mn = kw.get("max_locations", 1000), kw.get("max_neighborhood", 10000)
    for qi, qr in enumerate(got.queries):
        a = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
        assert a == per.get(qi, []), (qi, qs[qi], kw, len(a), len(per.get(qi, [])))
        want_msgs = [e["title"] for e in json.loads(lines[qi])["errors"]]
        assert qr.messages(ml, mn) == want_msgs, (qi, qs[qi], kw)
    return got


def sample(g, rng, L, edits=0):
    s = g["seqs"][rng.randrange(len(g["seqs"]))]
    while True:
        p = rng.randrange(len(s) - L)
        q = s[p:p + L]
        if all(c in "ACGT" for c in q):
            break
    q = list(q)
    for _ in range(edits):
        k = rng.randrange(len(q))
        r = rng.random()
        if r < 0.4:
            q[k] = rng.choice("ACGT")
        elif r < 0.7:
            del q[k]
        else:
            q.insert(k, rng.choice("ACGT"))
    return "".join(q)


def test_edit2_on_21_to_30_mers(ix, small_genome):
    """the normal primer lengths at -d 2: under the cap for most 21-24-mers, over it for most >= 25-mers"""
    from dicey_amd import DG_Q_NBHD_EXCEEDED
    rng = random.Random(7)
    qs = [sample(small_genome, rng, L, edits=rng.choice([0, 1, 2])) for L in (21, 22, 23, 24, 25, 25, 26, 27, 28, 30)]
    qs.append(revcomp(sample(small_genome, rng, 25, 1)))
    qs.append(sample(small_genome, rng, 20, 2))  # provably silent: stays inside the search kernel
    got = compare(ix, small_genome, qs, distance=2)
    fired = [bool(q.flags & DG_Q_NBHD_EXCEEDED) for q in got.queries]
    assert any(fired) and not all(fired)


def test_edit2_with_n_and_lowercase(ix, small_genome):
    rng = random.Random(8)
    base = sample(small_genome, rng, 20)
    qs = [base[:5] + "N" + base[6:12] + "N" + base[13:],  # two N at d=2 on a 20-mer: bound above the cap
          base[:3] + "nn" + base[5:11] + "R" + base[12:], base.lower(), "N" * 20, base[:10] + "N" * 10]
    compare(ix, small_genome, qs, distance=2)
    compare(ix, small_genome, qs, distance=2, forward_only=True, max_locations=4)


def test_hamming3_and_small_caps(ix, small_genome):
    rng = random.Random(9)
    qs = [sample(small_genome, rng, 20, rng.choice([0, 1, 2, 3])) for _ in range(6)] + ["A" * 20, "ACGT" * 5]
    compare(ix, small_genome, qs, distance=3, hamming=True)  # 20-mers at Hamming 3: 31 k strings, cap fires
    qs = [sample(small_genome, rng, L, 1) for L in (12, 16, 20, 24, 31, 40)] * 2
    for cap in (0, 1, 2, 17, 100, 145, 146):  # -x around the size of an edit-1 neighbourhood (7m+5 strings generated)
        compare(ix, small_genome, qs, distance=1, max_neighborhood=cap)
        compare(ix, small_genome, qs, distance=1, hamming=True, max_neighborhood=cap)
    compare(ix, small_genome, qs, distance=2, max_neighborhood=500, max_locations=3)


def test_max_locations_zero(ix, small_genome):
    """-m 0 (hunter.h:349,434): no hits, and the 'More than 0 matches' warning for every searched query"""
    rng = random.Random(10)
    qs = [sample(small_genome, rng, 20, 1) for _ in range(5)] + ["ACGTAC", "N" * 12]
    got = compare(ix, small_genome, qs, distance=1, max_locations=0)
    assert all(not q.hits for q in got.queries)


def test_mixed_batch_keeps_every_query(ix, small_genome):
    """one capped query must not change the answers of its neighbours in the batch"""
    rng = random.Random(11)
    qs = []
    for _ in range(40):
        qs.append(sample(small_genome, rng, rng.choice([10, 14, 20]), rng.choice([0, 1])))
    qs[7] = sample(small_genome, rng, 27, 1)
    qs[23] = sample(small_genome, rng, 25, 2)
    qs[24] = "acgtn" * 4
    compare(ix, small_genome, qs, distance=2)


def test_neighborhood_count_capped(ix, small_genome):
    """padlock's neighbourhood totals (padlock.h:396-405) for arms whose neighbourhood reaches the cap"""
    import ctypes as C
    L = ix._L
    rng = random.Random(12)
    arms = [sample(small_genome, rng, n) for n in (20, 22, 25, 25, 26)]
    buf = "".join(arms).encode()
    off = (C.c_uint64 * (len(arms) + 1))()
    t = 0
    for i, a in enumerate(arms):
        off[i] = t
        t += len(a)
    off[len(arms)] = t
    fw = (C.c_uint64 * len(arms))()
    rv = (C.c_uint64 * len(arms))()
    from dicey_amd import _capi
    _capi.check(L, L.dg_neighborhood_count(ix.handle, 2, 0, 10000, buf, off, len(arms), fw, rv))
    orc = O.Index(small_genome["fm9"])
    for i, a in enumerate(arms):
        for strand, got in ((a, fw[i]), (revcomp(a), rv[i])):
            want = sum(orc.count(s.encode()) for s in O.neighbors(strand, 2, True, 10000))
            assert got == want, (a, got, want)


def test_edit2_long_primers_on_the_kernel_path_with_a_wide_cap(ix, small_genome):
    """-x 200000: the cap cannot fire for 25…30-mers at d=2, so they run on k_search2p itself (two passes over the pairs of edit
    positions above 22 nt, strings up to 32 characters in one 64-bit register) instead of the host enumeration; the checker's
    hash-set neighbours take the same cap."""
    rng = random.Random(2930)
    qs = [sample(small_genome, rng, L, edits=rng.randrange(3)) for L in (23, 25, 26, 28, 29, 30, 30, 27)]
    qs += ["A" * 30, "ACGT" * 7 + "AC", "AAAAACCCCCGGGGGTTTTTAAAAACCCCC"]  # runs: the duplicate-string rules at work
    O.fast_neighbors(True)
    try:
        compare(ix, small_genome, qs, distance=2, max_neighborhood=200000)
        compare(ix, small_genome, qs[:6], distance=2, max_neighborhood=200000, max_locations=4, forward_only=True)
    finally:
        O.fast_neighbors(False)


def test_many_capped_primers_on_the_device_path(ix, small_genome):
    """200 primers of 21-29 nt at distance 2 (most of them reach the cap), low-complexity ones included, -x 10000 / 3000 / 64:
    hits, order and messages against the checker's hash-set neighbourhoods (tested equal to the literal ones)"""
    rng = random.Random(404)
    qs = [sample(small_genome, rng, rng.choice([21, 23, 25, 25, 27, 29]), edits=rng.choice([0, 1, 2])) for _ in range(60)]
    qs += ["A" * 25, "AC" * 13, "ACG" * 9, "AAAAACCCCCGGGGGTTTTTAAAAA", "T" * 21 + "G"]
    O.fast_neighbors(True)
    try:
        compare(ix, small_genome, qs, distance=2)
        compare(ix, small_genome, qs[:20], distance=2, max_neighborhood=3000, forward_only=True)
        compare(ix, small_genome, qs[20:40], distance=2, max_neighborhood=64, max_locations=5)
    finally:
        O.fast_neighbors(False)
