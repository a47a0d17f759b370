"""k_locate_topk (r03): the `max_locations` smallest positions of a repeat-rich interval through the suffix array's block
minima instead of passes over the interval (hunter.h:355-357: locate, std::sort, first min(occs, max_locations) entries).
Every case is compared with the oracle hit for hit; the genomes are built so that the walk starts at level 0, 1, 2 and 3 of the
hierarchy, with positions spread, clustered (tandem array) and mixed."""
import random

import pytest

try:
    import torch
    torch.cuda.is_available()
except Exception:  # pragma: no cover
    torch = None

import oracle_lib as O
from conftest import genome_text, open_index

pytestmark = pytest.mark.gpu


def _compare(ix, orc, g, qs, **kw):
    got = ix.hunt(qs, g["seqlen"], **kw)
    _, hits = orc.hunt(g["seqlen"], g["names"], qs, want_hits=True, **kw)
    per = {}
    for h in hits:
        per.setdefault(h[0], []).append(h[1:])
    for qi, qr in enumerate(got.queries):
        a = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
        assert a == per.get(qi, []), (qi, qs[qi], kw, len(a), len(per.get(qi, [])))
    return got


def _planted(seed, copies, background, tandem=0, unit_len=20, divergent=0.0):
    """two sequences: `copies` copies of a random unit at random places of a random background (+ a tandem array of `tandem` units)"""
    rng = random.Random(seed)
    unit = "".join(rng.choice("ACGT") for _ in range(unit_len))
    n = background
    bg = bytearray(rng.choice(b"ACGT") for _ in range(n))
    taken = set()
    for _ in range(copies):
        p = rng.randrange(0, n - unit_len - 1)
        u = unit
        if divergent and rng.random() < divergent:
            k = rng.randrange(unit_len)
            u = u[:k] + rng.choice("ACGT") + u[k + 1:]
        bg[p:p + unit_len] = u.encode()
        taken.add(p)
    s = bg.decode()
    if tandem:
        cut = n // 3
        s = s[:cut] + unit * tandem + s[cut:]
    half = len(s) // 2
    return unit, [s[:half], s[half:]]


def _index(tmp_path, seqs, name):
    import dicey_amd
    text = genome_text(seqs)
    path = str(tmp_path / name)
    dicey_amd.build_index(text, path, device=0)  # the GPU builder (byte-identical to the oracle's, tests/test_gpu_parity.py)
    return path, {"seqlen": [len(s) + 1 for s in seqs], "names": ["r%d" % (i + 1) for i in range(len(seqs))]}


@pytest.mark.parametrize("copies,background,tandem", [
    (3000, 400_000, 0),        # level 0: the interval fits the LDS buffer
    (40_000, 3_000_000, 0),    # starts at level 1
    (150_000, 8_000_000, 0),   # starts at level 2, spread positions
    (20_000, 2_000_000, 30_000),  # spread copies + a tandem array: most of the smallest positions share their top bytes
    (-3000, 400_000, 0),       # DICEY_NO_SA_MINIMA: no block minima, repeat-rich strings by radix select over the interval (k_locate_big)
])
def test_topk_locate_equals_oracle(tmp_path, monkeypatch, copies, background, tandem):
    _topk_case(tmp_path, monkeypatch, copies, background, tandem)


@pytest.mark.parametrize("copies,background,tandem", [(150_000, 8_000_000, 0), (20_000, 2_000_000, 30_000)])
def test_topk_locate_without_prefix_levels(tmp_path, monkeypatch, copies, background, tandem):
    """r06: strings with far more occurrences than they report take a run of a prefix level's records (FmView::plv) instead of the
    walk down the block minima; DICEY_NO_PLV opens the index without the levels — the walk must give the same hits"""
    monkeypatch.setenv("DICEY_NO_PLV", "1")
    _topk_case(tmp_path, monkeypatch, copies, background, tandem)


def _topk_case(tmp_path, monkeypatch, copies, background, tandem):
    import dicey_amd
    if copies < 0:
        copies = -copies
        monkeypatch.setenv("DICEY_NO_SA_MINIMA", "1")
    unit, seqs = _planted(copies + tandem, copies, background, tandem, divergent=0.2)
    path, g = _index(tmp_path, seqs, "rep.fm9")
    orc = O.Index(path)
    sub = unit[:19] + ("A" if unit[19] != "A" else "C")
    qs = [unit, sub, unit[1:] + "G", unit[:10] + unit[11:] + "T", seqs[0][5000:5020], unit[2:] + "AC"]
    with open_index(path) as ix:
        for kw in (dict(distance=0, max_locations=1000), dict(distance=1, max_locations=1000), dict(distance=1, max_locations=3),
                   dict(distance=0, max_locations=1024), dict(distance=0, max_locations=1025), dict(distance=1, max_locations=700),
                   dict(distance=1, hamming=True, max_locations=40), dict(distance=0, max_locations=1, forward_only=True),
                   dict(distance=1, max_locations=2500)):
            got = _compare(ix, orc, g, qs, **kw)
            assert len(got.queries[0].hits) == min(kw["max_locations"], len(got.queries[0].hits))
        assert len(ix.hunt([unit], g["seqlen"], distance=0, max_locations=1000).queries[0].hits) == 1000


def test_topk_locate_million_copy_family(tmp_path):
    """an Alu-scale family: 1.1 M copies of a 20-mer over 40 Mb (level 3 of the hierarchy), the first 1000 positions"""
    import dicey_amd
    rng = random.Random(77)
    unit = "".join(rng.choice("ACGT") for _ in range(20))
    parts = []
    for i in range(1_100_000):
        parts.append(unit)
        parts.append("".join(rng.choice("ACGT") for _ in range(rng.randint(8, 24))))
    s = "".join(parts)
    third = len(s) // 3
    seqs = [s[:third], s[third:2 * third], s[2 * third:]]
    path, g = _index(tmp_path, seqs, "alu.fm9")
    text = genome_text(seqs)
    # expected positions straight from the text (the oracle's locate of 1.1 M occurrences takes minutes)
    want, p = [], text.find(unit.encode())
    while p >= 0 and len(want) < 1000:
        want.append(p)
        p = text.find(unit.encode(), p + 1)
    with open_index(path) as ix:
        for m in (1000, 7):
            R = ix.hunt([unit], g["seqlen"], distance=0, max_locations=m, forward_only=True)
            hits = R.queries[0].hits
            assert len(hits) == m
            cum = [0]
            for x in g["seqlen"]:
                cum.append(cum[-1] + x)
            got = [cum[h.chr] + h.start - 1 for h in hits]
            assert got == want[:m]
            assert all(h.score == 0 and h.refalign == unit and h.queryalign == unit for h in hits)



def test_job_kernels_come_back_after_a_batch_without_repeat_rich_strings(tmp_path):
    """The locate job kernels are left out when the previous batch of the handle queued no job; a batch that then needs them is
    repeated with them (and until then the verify kernel must not touch hit slots nobody wrote)."""
    import dicey_amd
    unit, seqs = _planted(5, 3000, 400_000, divergent=0.1)
    path, g = _index(tmp_path, seqs, "hint.fm9")
    orc = O.Index(path)
    rng = random.Random(12)
    plain = ["".join(rng.choice("ACGT") for _ in range(20)) for _ in range(64)]  # random 20-mers: at most a stray hit, no job
    rich = [unit, unit[1:] + "A", seqs[1][777:797]]
    with open_index(path) as ix:
        for qs in (plain, rich, plain, plain, rich, rich):
            _compare(ix, orc, g, qs, distance=1, max_locations=1000)


def _family_genome(seed, unit_len=60, copies=420):
    """three sequences with a repeat family whose copies also sit where a hit's context is special: at the very start of the text,
    at the end of a sequence (the next character is the separator), at the start of one, and flanked by N runs"""
    rng = random.Random(seed)
    unit = "".join(rng.choice("ACGT") for _ in range(unit_len))
    seqs = []
    for c in range(3):
        bg = bytearray(rng.choice(b"ACGT") for _ in range(150_000))
        for _ in range(copies // 3):
            p = rng.randrange(100, len(bg) - unit_len - 100)
            bg[p:p + unit_len] = unit.encode()
        for _ in range(6):  # N runs that touch a copy on either side, and one character away from it
            p = rng.randrange(1000, len(bg) - 1000)
            gap = rng.choice((0, 0, 1, 2))
            bg[p - 40 - gap:p - gap] = b"N" * 40
            bg[p:p + unit_len] = unit.encode()
            bg[p + unit_len + gap:p + unit_len + gap + 30] = b"N" * 30
        seqs.append(bg.decode())
    seqs[0] = unit + seqs[0][unit_len:]                          # text position 0
    seqs[0] = seqs[0][:-unit_len] + unit                         # followed by the separator
    seqs[1] = unit + seqs[1][unit_len:-unit_len - 1] + unit + "G"  # preceded by the separator; one character before the next
    seqs[2] = seqs[2][:-unit_len] + unit                         # the end of the text
    return unit, seqs


@pytest.mark.parametrize("no_sax", [False, True])
def test_hits_of_repeat_rich_strings_carry_their_context(tmp_path, monkeypatch, no_sax):
    """r06: the locate job kernels read {position, context word} records (FmView::sax) and k_verify_memo takes the <= d characters
    either side of a hit from its seed instead of the text (hunter.h:363-378).  Copies at the start / end of the text and of
    sequences, next to N runs, strings shorter and longer than the context window serves (16..27 characters), distances 0-2;
    the same without the records (DICEY_NO_SAX: every hit reads the text)."""
    import dicey_amd
    if no_sax:
        monkeypatch.setenv("DICEY_NO_SAX", "1")
    unit, seqs = _family_genome(31)
    path, g = _index(tmp_path, seqs, "fam.fm9")
    orc = O.Index(path)
    rng = random.Random(8)
    qs = []
    for m in (15, 17, 18, 20, 22, 25, 27, 29):
        for off in (0, 1, 7, len(unit) - m - 1, len(unit) - m):
            q = unit[off:off + m]
            qs.append(q)
            k = rng.randrange(m)
            qs.append(q[:k] + rng.choice("ACGT") + q[k + 1:])   # a substitution
            qs.append(q[:k] + q[k + 1:])                        # a deletion
            qs.append(q[:k] + rng.choice("ACGT") + q[k:])       # an insertion
    with open_index(path) as ix:
        for kw in (dict(distance=1, max_locations=1000), dict(distance=0, max_locations=1000), dict(distance=1, max_locations=100),
                   dict(distance=1, hamming=True, max_locations=1000)):
            _compare(ix, orc, g, qs, **kw)
        short = [q for q in qs if len(q) <= 22][::3]   # (the checker needs a third of a second for each of these)
        O.fast_neighbors(True)
        try:
            _compare(ix, orc, g, short, distance=2, max_locations=300)
        finally:
            O.fast_neighbors(False)


def test_small_buffer_topk_kernel_serves_a_batch_with_thousands_of_repeat_rich_strings(tmp_path):
    """k_locate_topk<576> (list JL_MID) only runs when the handle's previous batch queued >= 2 048 workgroup jobs: 2 400 queries that
    each hit a 420-copy family through several strings, twice (the first batch sets the hint)."""
    import dicey_amd
    unit, seqs = _family_genome(32)
    path, g = _index(tmp_path, seqs, "fam2.fm9")
    orc = O.Index(path)
    rng = random.Random(9)
    qs = []
    while len(qs) < 2400:
        off = rng.randrange(0, len(unit) - 20 + 1)
        q = unit[off:off + 20]
        k = rng.randrange(20)
        qs.append(q[:k] + rng.choice("ACGT") + q[k + 1:])
    with open_index(path) as ix:
        ix.hunt(qs, g["seqlen"], distance=1, max_locations=1000)
        got = _compare(ix, orc, g, qs, distance=1, max_locations=1000)
        assert sum(len(q.hits) for q in got.queries) > 400 * 2000
        _compare(ix, orc, g, qs[:300], distance=1, max_locations=150)


@pytest.mark.parametrize("K,K2", [(16, 18), (17, 18)])
def test_single_occurrence_hits_are_aligned_from_their_own_codes(tmp_path, monkeypatch, K, K2):
    """r06: a kept string with ONE occurrence found through a filtered table interval leaves k_search1s with its characters (Sel::key)
    and the character in front of the occurrence; k_locate takes the position and the character behind it from the suffix's record
    (FmView::sax) and k_verify_memo aligns the hit without a text line (band_window_from_key).  Unique windows of many short
    sequences — at their first and last positions (the neighbour is the separator: no context), one and two characters inside,
    next to N runs and to lower-case / IUPAC characters — as exact queries and with one edit, table orders 16 and 17 (the default
    layouts of a 3 Gb genome, forced on this small one); DICEY_NO_DIRECT_CTX (read per batch by the development build): the same hits
    with the text read as before — and fewer record reads, i.e. the path under test really ran."""
    monkeypatch.setenv("DICEY_KMER_K", str(K))
    monkeypatch.setenv("DICEY_KMER_K2", str(K2))
    monkeypatch.setenv("DICEY_NO_DIRECT_CTX", "1")   # (also makes open_index load the development build)
    rng = random.Random(40 + K)
    seqs = []
    for c in range(40):
        s = bytearray(rng.choice(b"ACGT") for _ in range(rng.randrange(300, 900)))
        for _ in range(3):
            p = rng.randrange(40, len(s) - 60)
            s[p:p + rng.randrange(1, 12)] = b"N" * rng.randrange(1, 12)
        for _ in range(3):
            p = rng.randrange(40, len(s) - 60)
            s[p] = rng.choice(b"RYKMacgt")
        seqs.append(s.decode())
    path, g = _index(tmp_path, seqs, "uniq%d.fm9" % K)
    orc = O.Index(path)
    qs = []
    for s in seqs:
        n = len(s)
        starts = [0, 1, 2, 3, n - 20, n - 21, n - 22, n - 23] + [rng.randrange(0, n - 25) for _ in range(10)]
        npos = [i for i, ch in enumerate(s) if ch == "N"]
        for p in npos[:2] + npos[-2:]:   # windows that end / start right at an N run, and one character away
            starts += [p - 20, p - 21, p + 1, p + 2]
        for st in starts:
            for m in (18, 20, 21, 23):
                if st < 0 or st + m > n:
                    continue
                q = s[st:st + m].upper()
                if any(ch not in "ACGT" for ch in q):
                    continue
                k = rng.randrange(m)
                qs.append(rng.choice([q, q[:k] + rng.choice("ACGT") + q[k + 1:], q[:k] + q[k + 1:], q[:k] + rng.choice("ACGT") + q[k:]]))
    qs = qs[:6000]
    with open_index(path) as ix:
        plain = _compare(ix, orc, g, qs, distance=1)
        monkeypatch.delenv("DICEY_NO_DIRECT_CTX")
        got = _compare(ix, orc, g, qs, distance=1)
        nh = sum(len(q.hits) for q in got.queries)
        assert nh > len(qs)
        # a hit of the new path counts its 8-byte record as two suffix-array reads: most hits of this batch take it
        assert got.counters["sa_reads"] - plain.counters["sa_reads"] > nh // 2, (got.counters, plain.counters, nh)
        _compare(ix, orc, g, qs[::5], distance=1, max_locations=1)
        _compare(ix, orc, g, qs[::7], distance=0)
        _compare(ix, orc, g, qs[::9], distance=1, forward_only=True)


@pytest.mark.parametrize("K", [16, 0])
def test_short_queries_that_end_in_n_keep_their_hits_next_to_n_runs(tmp_path, monkeypatch, K):
    """r06 regression (found by tools/fuzz_hunt.py in the full-size layout): a batch whose queries are all shorter than the table
    order has no flat kernel; a strand with one N at an end is marked for k_nkeep all the same, and k_nkeep used to be launched only
    beside a flat kernel — the strings that keep the N (the unedited query in Hamming mode: exact hits where the text continues with
    an N run) were never searched.  Queries of 10-15 nt ending / starting at the N runs of a small text, table order 16 and the
    small text's own order, edit and Hamming mode, alone in the batch and next to 20-mers (which bring the flat kernel along).
    Second regression of the same round (tools/fuzz_n.py): queries with one character FEWER than the text next to the run — the string
    that occurs is an insertion that keeps the N, and k_nkeep tried only eight of the nine operations per position."""
    if K:
        monkeypatch.setenv("DICEY_KMER_K", str(K))
        monkeypatch.setenv("DICEY_KMER_K2", "18")
    rng = random.Random(77)
    seqs = []
    for c in range(6):
        s = bytearray(rng.choice(b"ACGT") for _ in range(4000))
        for _ in range(12):
            p = rng.randrange(100, len(s) - 100)
            n = rng.randrange(2, 25)
            s[p:p + n] = b"N" * n
        seqs.append(s.decode())
    path, g = _index(tmp_path, seqs, "nruns%d.fm9" % K)
    orc = O.Index(path)
    short, long_ = [], []
    for s in seqs:
        for p in [i for i in range(20, len(s) - 20) if s[i] == "N" and s[i - 1] != "N"][:8]:   # first N of a run
            for m in (10, 12, 15):
                w = s[p - m + 1:p + 1]
                if "N" in w[:-1]:
                    continue
                short += [w, w[:3] + rng.choice("ACGT") + w[4:]]
                # the text holds one character MORE than the query (the neighbourhood string is an insertion that keeps the N): every
                # inserted base must be tried — k_nkeep enumerated eight of the nine operations per position and never inserted a T
                w1 = s[p - m:p + 1]
                if "N" not in w1[:-1]:
                    k = rng.randrange(2, m - 2)
                    short.append(w1[:k] + w1[k + 1:])
            long_.append(s[p - 19:p + 1])
        for p in [i for i in range(20, len(s) - 20) if s[i] == "N" and s[i + 1] != "N"][:8]:   # last N of a run
            for m in (10, 13):
                w = s[p:p + m]
                if "N" in w[1:]:
                    continue
                short += [w, w[:-4] + rng.choice("ACGT") + w[-3:]]
                w1 = s[p:p + m + 1]
                if "N" not in w1[1:]:
                    k = rng.randrange(3, m - 1)
                    short.append(w1[:k] + w1[k + 1:])
    with open_index(path) as ix:
        for kw in (dict(distance=1, hamming=True), dict(distance=1), dict(distance=1, hamming=True, forward_only=True, max_locations=2)):
            got = _compare(ix, orc, g, short, **kw)
            assert sum(len(q.hits) for q in got.queries) >= len(short) // 3, kw
            _compare(ix, orc, g, short[::3] + long_, **kw)
