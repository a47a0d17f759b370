"""The layout a GRCh38-size index gets by default — K-mer table of order 17 (34-bit codes, 137 GB), long presence filter of order 18
(36-bit codes, four permuted copies) — forced on the small test genomes with DICEY_KMER_K / DICEY_KMER_K2 (index.hip), so that
the code paths only a 3 Gb genome reaches otherwise (kf_word / head_window_occurs on codes above 32 bits, queries shorter than
the table order next to longer ones) run under `-m gpu` against the oracle: the whole comparison matrix of the other GPU modules
(edit / Hamming, distance 0-2, forward only, -m 3, N and lower case, 10-31-mers, capped 25-mers, > 255 nt, `search` binding
sites, padlock counts) with every index opened in this layout.  A second, smaller pass runs the 16 / 19 layout (r02's other
candidate).  Then two size checks: 1 000 distance-2 queries on a 100 Mb genome against the oracle, and one 1.25 M-query
distance-2 batch (configs[3]'s per-GPU share) through size-independent properties."""
import os
import random
from concurrent.futures import ThreadPoolExecutor

import pytest

try:
    import torch
    torch.cuda.is_available()
except Exception:  # pragma: no cover
    torch = None

import oracle_lib as O
from conftest import open_index, genome_text, make_genome, make_queries, revcomp

pytestmark = pytest.mark.gpu


class layout:
    """every index opened inside the block gets table order K and long-filter order K2"""

    def __init__(self, K, K2):
        self.env = {"DICEY_KMER_K": str(K), "DICEY_KMER_K2": str(K2)}

    def __enter__(self):
        self.old = {k: os.environ.get(k) for k in self.env}
        os.environ.update(self.env)

    def __exit__(self, *a):
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _open(small_genome):
    import dicey_amd
    return dicey_amd.FmIndex(small_genome["fm9"], device=0)


def test_layout_is_what_was_asked_for(small_genome):
    """the forced orders really are in place: 8 * 4^17 bytes of table + 4 * 4^18 / 8 bytes of long filter"""
    with layout(17, 18), _open(small_genome) as ix:
        assert ix.stats()["hbm_bytes"] > (8 << 34) + 4 * ((1 << 36) >> 3)


@pytest.mark.parametrize("K,K2", [(17, 18), (16, 19)])
def test_hunt_matrix_in_the_full_size_layout(small_genome, K, K2):
    import test_gpu_parity as P
    full = K == 17
    with layout(K, K2), _open(small_genome) as ix:
        for kw, nq, lens in [(dict(distance=1), 1500 if full else 400, (20,)), (dict(distance=0), 300, (18, 20)),
                             (dict(distance=1, hamming=True), 500, (20, 15)), (dict(distance=2, hamming=True), 200, (20,)),
                             (dict(distance=1, forward_only=True), 300, (12, 25, 31)), (dict(distance=1, max_locations=3), 400, (10, 11, 12)),
                             (dict(distance=1), 300, tuple(range(10, 32)))]:
            P.test_hunt_hits_equal_oracle_push_order(ix, small_genome, kw, nq, lens)
        P.test_hunt_edge_cases(ix, small_genome)
        P.test_long_queries_are_answered_next_to_short_ones(ix, small_genome)
        P.test_former_envelope_is_answered(ix, small_genome)
        P.test_flat_distance_two_kernel(ix, small_genome, None)
        if full:
            P.test_hunt_edit_distance_two(ix, small_genome)
            P.test_device_pointer_view_for_the_rccl_gather(ix, small_genome)
            P.test_device_entry_point_rechecks_a_cached_length_bound(ix, small_genome)


def test_capped_neighbourhoods_in_the_full_size_layout(small_genome):
    import test_gpu_capped as T
    with layout(17, 18), _open(small_genome) as ix:
        T.test_edit2_on_21_to_30_mers(ix, small_genome)
        T.test_edit2_with_n_and_lowercase(ix, small_genome)
        T.test_max_locations_zero(ix, small_genome)
        T.test_mixed_batch_keeps_every_query(ix, small_genome)
        T.test_neighborhood_count_capped(ix, small_genome)
        T.test_edit2_long_primers_on_the_kernel_path_with_a_wide_cap(ix, small_genome)


def test_search_sites_and_padlock_counts_in_the_full_size_layout(tmp_path_factory):
    """dg_search_sites (k_search1 on the 15-mer neighbourhoods + k_site_wave) and dg_neighborhood_count open their own indexes:
    the environment puts them into the 17 / 18 layout"""
    import test_gpu_search as S
    import test_gpu_padlock as PL
    with layout(17, 18):
        if O.ref_libs() is not None:
            pcr = S.make_pcr(tmp_path_factory)
            S.test_search_sites_library_level(pcr)
            S.test_search_json_identical_to_oracle(pcr, [], {})
        PL.test_neighborhood_count_matches_oracle(PL.make_scenario(tmp_path_factory))


def _oracle_hits_parallel(path, g, qs, workers=32, **kw):
    """the oracle's hunt loop over query chunks on host threads (ctypes releases the GIL; the handle is read-only)"""
    orc = O.Index(path)
    chunk = max(1, (len(qs) + workers - 1) // workers)
    parts = [(i, qs[i:i + chunk]) for i in range(0, len(qs), chunk)]

    def run(part):
        base, sub = part
        _, hits = orc.hunt(g["seqlen"], g["names"], sub, want_hits=True, **kw)
        per = {}
        for h in hits:
            per.setdefault(base + h[0], []).append(h[1:])
        return per
    out = {}
    with ThreadPoolExecutor(max_workers=workers) as ex:
        for per in ex.map(run, parts):
            out.update(per)
    return out


@pytest.fixture(scope="module")
def genome_100mb(tmp_path_factory):
    """104 Mb in 6 sequences, i.i.d. with N runs; built on the device, indexed by the GPU builder"""
    import dicey_amd
    import numpy as np
    rng = np.random.default_rng(2024)
    lens = [30_000_000, 24_000_000, 18_000_000, 14_000_000, 10_000_000, 8_000_000]
    seqs = []
    for L in lens:
        a = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, L)]
        for _ in range(20):
            p = int(rng.integers(0, L - 5000))
            a[p:p + int(rng.integers(10, 4000))] = ord("N")
        seqs.append(a.tobytes().decode())
    text = genome_text(seqs)
    path = str(tmp_path_factory.mktemp("g100") / "g100.fm9")
    dicey_amd.build_index(text, path, device=0)
    return {"seqs": seqs, "fm9": path, "seqlen": [len(s) + 1 for s in seqs], "names": ["c%d" % i for i in range(len(seqs))]}


def _planted_queries(g, rng, n, m=20, edits=(0, 1, 2)):
    qs, origin = [], []
    while len(qs) < n:
        c = rng.randrange(len(g["seqs"]))
        p = rng.randrange(len(g["seqs"][c]) - m - 2)
        w = g["seqs"][c][p:p + m + 2]
        if "N" in w:
            continue
        q = list(w[:m])
        ne = rng.choice(edits)
        for _ in range(ne):  # substitutions only: the origin stays a hit at distance <= ne
            k = rng.randrange(m)
            q[k] = rng.choice([x for x in "ACGT" if x != q[k]])
        q = "".join(q)
        strand = "+"
        if rng.random() < 0.3:
            q, strand = revcomp(q), "-"
        qs.append(q)
        origin.append((c, p + 1, strand, ne))
    return qs, origin


def test_distance_two_on_100mb_against_the_oracle(genome_100mb):
    """1 000 20-mers at edit distance 2 on a 104 Mb genome (table order 14, long filter 16): hits in push order, alignments
    included, equal to the oracle's (hash-set neighbourhoods, tested equal to the literal ones in tests/test_oracle.py)"""
    import dicey_amd
    g = genome_100mb
    rng = random.Random(99)
    qs, _ = _planted_queries(g, rng, 900)
    qs += ["".join(rng.choice("ACGT") for _ in range(20)) for _ in range(100)]
    O.fast_neighbors(True)
    try:
        want = _oracle_hits_parallel(g["fm9"], g, qs, distance=2)
    finally:
        O.fast_neighbors(False)
    with dicey_amd.FmIndex(g["fm9"]) as ix:
        got = ix.hunt(qs, g["seqlen"], distance=2)
    nh = 0
    for qi, qr in enumerate(got.queries):
        a = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
        assert a == want.get(qi, []), (qi, qs[qi])
        nh += len(a)
    assert nh >= 900


def test_one_and_a_quarter_million_queries_at_distance_two(genome_100mb):
    """configs[3] gives every GPU 10 M / 8 = 1.25 M queries: one batch of that size at edit distance 2 (leaf and hit buffers 12x the
    bench's).  Properties: every planted query is found at its origin with a score no worse than its number of substitutions,
    every hit's alignment rows are consistent with its score, and a 2 000-query slice of the same batch run on its own gives
    the same hits."""
    import dicey_amd
    g = genome_100mb
    rng = random.Random(1250)
    qs, origin = _planted_queries(g, rng, 1_250_000)
    with dicey_amd.FmIndex(g["fm9"]) as ix:
        got = ix.hunt(qs, g["seqlen"], distance=2)
        assert len(got.queries) == len(qs)
        missing = 0
        for q, (c, p, strand, ne), r in zip(qs, origin, got.queries):
            ok = False
            for h in r.hits:
                if h.chr == c and h.strand == strand and abs(h.start - p) <= 2 and -h.score <= ne:
                    ok = True
                    break
            missing += not ok
        assert missing == 0
        for r in got.queries[::97]:
            for h in r.hits:
                assert len(h.refalign) == len(h.queryalign)
                cost = sum(1 for x, y in zip(h.refalign, h.queryalign) if x != y)
                assert cost == -h.score and 0 <= cost <= 2
        lo = 600_000
        sub = ix.hunt(qs[lo:lo + 2000], g["seqlen"], distance=2)
        key = lambda R: [[(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in q.hits] for q in R]
        assert key(sub.queries) == key(got.queries[lo:lo + 2000])


@pytest.mark.parametrize("prune", [True, False])
def test_queries_with_n_where_the_text_has_no_short_n_run(genome_100mb, prune, monkeypatch):
    """r05: the text's shortest run of N is >= 10 here, so k_search knows that no string which keeps a query's N can occur and walks
    only the strings whose N's are substituted or deleted (FmView::nrun_min).  Hits, order and alignments against the oracle for
    queries with one, two and three N's (inside, near the ends — where the pruning must stay off —, adjacent, lower case / IUPAC
    letters), edit and Hamming mode, distance 1 and 2; DICEY_NO_NRUN_PRUNE runs the unpruned walk on the same queries."""
    import dicey_amd
    if not prune:
        monkeypatch.setenv("DICEY_NO_NRUN_PRUNE", "1")
    g = genome_100mb
    rng = random.Random(555)
    base, _ = _planted_queries(g, rng, 240, edits=(0, 1))
    qs = []
    for i, q in enumerate(base):
        q = list(q)
        k = i % 6
        if k == 0:
            q[rng.randrange(3, 17)] = "N"
        elif k == 1:
            a = rng.randrange(3, 16)
            q[a] = "N"
            q[a + 1] = "n"
        elif k == 2:
            q[rng.randrange(0, 2)] = "N"             # within d of the left end
        elif k == 3:
            q[19 - rng.randrange(0, 2)] = "R"        # within d of the right end, an IUPAC letter
        elif k == 4:
            for p in rng.sample(range(2, 18), 3):
                q[p] = "N"
        else:
            q[rng.randrange(4, 16)] = "y"
        qs.append("".join(q))
    with open_index(g["fm9"]) as ix:
        for kw, sub in ((dict(distance=1), qs), (dict(distance=1, hamming=True), qs[:120]), (dict(distance=2, hamming=True), qs[:60])):
            want = _oracle_hits_parallel(g["fm9"], g, sub, **kw)
            got = ix.hunt(sub, g["seqlen"], **kw)
            nh = 0
            for qi, qr in enumerate(got.queries):
                a = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
                assert a == want.get(qi, []), (qi, sub[qi], kw)
                nh += len(a)
            assert nh > 20, (kw, nh)
        O.fast_neighbors(True)
        try:
            sub = qs[:36]
            want = _oracle_hits_parallel(g["fm9"], g, sub, distance=2)
            got = ix.hunt(sub, g["seqlen"], distance=2)
            for qi, qr in enumerate(got.queries):
                a = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
                assert a == want.get(qi, []), (qi, sub[qi])
        finally:
            O.fast_neighbors(False)
