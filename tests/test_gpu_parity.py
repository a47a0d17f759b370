"""GPU parity tests (run with -m gpu on an MI355X): every call goes through the C ABI of libdiceygpu.so and is
compared bit for bit with the oracle on the same seeded inputs."""
import os
import random
import zlib

import pytest

try:  # torch bundles its own HIP runtime: it only finds the GPU if it initialises before libdiceygpu's (system) runtime does
    import torch
    torch.cuda.is_available()
except Exception:  # pragma: no cover
    torch = None

import oracle_lib as O
from conftest import open_index, genome_text, make_genome, make_queries

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gpu_small(small_genome):
    import dicey_amd
    ix = dicey_amd.FmIndex(small_genome["fm9"], device=0)
    yield ix
    ix.close()


def test_native_library_is_the_loaded_one():
    from dicey_amd import _capi
    L = _capi.load()
    assert L.dg_device_count() >= 1
    maps = open("/proc/self/maps").read()
    assert "libdiceygpu.so" in maps and "hostemu" not in maps


def test_index_stats(gpu_small, small_genome):
    st = gpu_small.stats()
    assert st["n"] == len(small_genome["text"]) + 1
    assert st["hbm_bytes"] > st["n"] * 5


def test_count_locate_extract_match_bruteforce(gpu_small, small_genome):
    text = small_genome["text"]
    rng = random.Random(17)
    pats = []
    for _ in range(3000):
        if rng.random() < 0.7:
            p = rng.randrange(len(text) - 14)
            pats.append(text[p:p + rng.randint(1, 14)])
        else:
            pats.append("".join(rng.choice("ACGTNRY") for _ in range(rng.randint(1, 8))).encode())
    pats += [b"\n", b"N" * 30, b"A", text[:40], text[-40:]]
    want = [O.bf_locate(text, p) for p in pats]
    assert gpu_small.count(pats) == [len(w) for w in want]
    assert gpu_small.locate(pats) == want
    full = text + b"\0"
    rs = []
    for _ in range(500):
        b = rng.randrange(len(text))
        rs.append((b, min(len(text), b + rng.randint(0, 60))))
    rs += [(0, 0), (len(text), len(text)), (0, len(text))]
    assert gpu_small.extract(rs) == [full[a:b + 1] for a, b in rs]


def _compare(ix, orc, g, qs, **kw):
    got = ix.hunt(qs, g["seqlen"], **kw)
    _, hits = orc.hunt(g["seqlen"], g["names"], qs, want_hits=True, **kw)
    per = {}
    for h in hits:
        per.setdefault(h[0], []).append(h[1:])
    for qi, qr in enumerate(got.queries):
        a = [(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in qr.hits]
        assert a == per.get(qi, []), (qi, qs[qi], kw)
    return got


@pytest.mark.parametrize("kw,nq,lens", [
    (dict(distance=1), 1500, (20,)),
    (dict(distance=0), 300, (18, 20)),
    (dict(distance=1, hamming=True), 500, (20, 15)),
    (dict(distance=2, hamming=True), 200, (20,)),
    (dict(distance=1, forward_only=True), 300, (12, 25, 31)),
    (dict(distance=1, max_locations=3), 400, (10, 11, 12)),
])
def test_hunt_hits_equal_oracle_push_order(gpu_small, small_genome, kw, nq, lens):
    orc = O.Index(small_genome["fm9"])
    qs = make_queries(zlib.crc32(str(sorted(kw.items())).encode()) % 1000, small_genome["text"], nq, lens)
    _compare(gpu_small, orc, small_genome, qs, **kw)


def test_hunt_edge_cases(gpu_small, small_genome):
    g = small_genome
    orc = O.Index(g["fm9"])
    s = g["seqs"]
    qs = [s[0][:20], s[0][-20:], s[1][:19], s[2][-21:], s[1][1:21], s[2][:10], s[0][-10:],  # chromosome starts / ends
          "acgtnacgtacgtacgtacg", "ACGU" * 5, "A" * 20, "N" * 20, "ACGTAC", "", "ACGTACGTA",  # non-DNA, too short, empty
          s[0][5000:5060], s[1][300:400]]  # long queries
    got = _compare(gpu_small, orc, g, qs, distance=1)
    from dicey_amd import DG_Q_TOO_SHORT
    assert [bool(q.flags & DG_Q_TOO_SHORT) for q in got.queries[11:14]] == [True, True, True]
    assert got.queries[7].sequence == "ACGTNACGTACGTACGTACG" and got.queries[7].nondna == 1
    assert got.queries[8].nondna == 5
    _compare(gpu_small, orc, g, qs, distance=1, hamming=True)


def test_unsupported_envelope_fails_loudly(gpu_small, small_genome):
    import dicey_amd
    with pytest.raises(dicey_amd.DgError):
        gpu_small.hunt(["ACGT" * 8000], small_genome["seqlen"], distance=1)  # 32 000 nt: alignment lengths are 16-bit
    with pytest.raises(dicey_amd.DgError):
        gpu_small.hunt(["ACGTACGTACGT"], small_genome["seqlen"], distance=30)


def test_long_queries_are_answered_next_to_short_ones(gpu_small, small_genome):
    """queries above 255 nt (round 1 failed their whole batch): banded verify with the trace in HBM; the short queries of
    the same batch keep their answers"""
    g = small_genome
    orc = O.Index(g["fm9"])
    s = g["seqs"]
    long1 = s[0][1000:1256]                                   # 256 nt, exact
    long2 = s[1][200:500] + "A" + s[1][500:1100]              # 901 nt with an insertion
    long3 = s[2][3000:3300][:150] + s[2][3000:3300][151:]     # 299 nt with a deletion
    long4 = s[0][7000:7600]
    long4 = long4[:300] + ("C" if long4[300] != "C" else "G") + long4[301:]  # 600 nt with a substitution
    qs = [s[0][40:60], long1, s[1][900:925], long2, long3, "ACGTACGTACGTACGTACGT", long4, s[2][10:40]]
    _compare(gpu_small, orc, g, qs, distance=1)
    _compare(gpu_small, orc, g, qs, distance=1, hamming=True)
    _compare(gpu_small, orc, g, qs, distance=0)
    _compare(gpu_small, orc, g, [long1, s[0][40:60]], distance=2, hamming=True)  # the cap fires for the long one: host enumeration


def test_former_envelope_is_answered(gpu_small, small_genome):
    """inputs round 1 refused because the maxNeighborhood cap could fire: now answered like the reference (the capped
    enumeration is reproduced on the host, tests/test_gpu_capped.py has the broad coverage)"""
    orc = O.Index(small_genome["fm9"])
    # an N can become any of four bases: 38 N's at Hamming distance 2 are 11 401 strings, the cap (10 000) fires
    _compare(gpu_small, orc, small_genome, ["ACGTACGTAC" * 3 + "ACGTACGT", "N" * 38, "ACGTNNACGT" * 3 + "ACGTACGT"], distance=2, hamming=True)
    _compare(gpu_small, orc, small_genome, ["ACGTNACGTNACGTACGTAC", "ACGTNACGTAACGTACGTAC", "ACGT" * 5 + "A"], distance=2)


def test_larger_genome_roundtrip_properties():
    """Size-independent properties on a multi-megabase genome: every genome-sampled query is found at its origin with
    distance 0, and locate() positions really spell the pattern."""
    import dicey_amd
    import tempfile, os
    seqs = make_genome(5, 4, 400000, repeats=False)
    text = genome_text(seqs)
    with tempfile.TemporaryDirectory() as td:
        path = os.path.join(td, "big.fm9")
        O.build_fm9(text, path)
        with dicey_amd.FmIndex(path) as ix:
            rng = random.Random(1)
            seqlen = [len(s) + 1 for s in seqs]
            qs, origin = [], []
            while len(qs) < 5000:
                c = rng.randrange(4)
                p = rng.randrange(len(seqs[c]) - 20)
                q = seqs[c][p:p + 20]
                if "N" in q:
                    continue
                qs.append(q)
                origin.append((c, p + 1))
            got = ix.hunt(qs, seqlen, distance=1)
            for q, (c, p), r in zip(qs, origin, got.queries):
                assert any(h.score == 0 and h.chr == c and h.start == p and h.strand == "+" and h.refalign == q for h in r.hits)
            locs = ix.locate([q.encode() for q in qs[:500]])
            for q, ps in zip(qs, locs):
                assert ps and all(text[x:x + 20] == q.encode() for x in ps)


@pytest.mark.parametrize("seed,nchr,length,iupac", [(31, 1, 1000, False), (32, 3, 30000, True), (33, 5, 200000, False)])
def test_gpu_index_builder_writes_the_same_file_as_the_oracle(tmp_path, seed, nchr, length, iupac):
    """dg_index_build (GPU suffix array + wavelet tree + sdsl serialisation) vs the oracle's CPU construction:
    the two .fm9 files must be byte-identical."""
    import dicey_amd
    seqs = make_genome(seed, nchr, length, iupac=iupac)
    text = genome_text(seqs)
    a, b = str(tmp_path / "gpu.fm9"), str(tmp_path / "cpu.fm9")
    dicey_amd.build_index(text, a)
    O.build_fm9(text, b)
    ga, gb = open(a, "rb").read(), open(b, "rb").read()
    assert len(ga) == len(gb)
    assert ga == gb


def test_gpu_builder_handles_long_repeats(tmp_path):
    """Prefix doubling must converge on texts with long exact repeats and long N runs."""
    import dicey_amd
    rng = random.Random(4)
    unit = "".join(rng.choice("ACGT") for _ in range(700))
    seqs = [unit * 6 + "N" * 5000 + unit[:333] + "ACGT" * 500, "N" * 3000 + unit * 2]
    text = genome_text(seqs)
    a, b = str(tmp_path / "gpu.fm9"), str(tmp_path / "cpu.fm9")
    dicey_amd.build_index(text, a)
    O.build_fm9(text, b)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_capacity_retry_path(small_genome, monkeypatch):
    """Leaf regions and hit buffers start from guesses the kernels check themselves; a batch that overflows them is
    repeated with larger buffers and must give the same answer."""
    import dicey_amd
    monkeypatch.setenv("DICEY_DEBUG_CAPS", "2")
    orc = O.Index(small_genome["fm9"])
    with open_index(small_genome["fm9"]) as ix:
        qs = make_queries(77, small_genome["text"], 400)
        _compare(ix, orc, small_genome, qs, distance=1)
        # distance 2 under the same tiny capacities (leaf regions overflow several times before they fit)
        O.fast_neighbors(True)
        try:
            _compare(ix, orc, small_genome, [q[:m] for q, m in zip(qs[:40], [12, 14, 20, 18] * 10) if len(q) >= m], distance=2)
        finally:
            O.fast_neighbors(False)


def test_hunt_edit_distance_two(gpu_small, small_genome):
    """BASELINE configs[3] shape (20-mers, edit distance 2) on a handful of queries — the oracle's neighbors() needs
    seconds per 20-mer at d=2, exactly like the reference."""
    orc = O.Index(small_genome["fm9"])
    for n, m in ((10, 12), (6, 15), (4, 20)):
        qs = [q[:m] for q in make_queries(200 + m, small_genome["text"], 4 * n, (m,)) if len(q) >= m][:n]
        _compare(gpu_small, orc, small_genome, qs, distance=2)


def test_device_pointer_view_for_the_rccl_gather(gpu_small, small_genome):
    """bench.py hands the hit records to torch.distributed as views of libdiceygpu's device buffers."""
    import ctypes as C
    import torch
    from dicey_amd import _capi
    from dicey_amd.shard import device_bytes
    L = _capi.load()
    qs = make_queries(5, small_genome["text"], 50)
    qb = b"".join(q.encode() for q in qs)
    off = [0]
    for q in qs:
        off.append(off[-1] + len(q))
    d_q = torch.frombuffer(bytearray(qb), dtype=torch.uint8).cuda()
    d_off = torch.tensor(off, dtype=torch.int64).cuda()
    sl = (C.c_uint32 * 3)(*small_genome["seqlen"])
    p = _capi.HuntParams(1, 0, 0, 1000, 10000)
    rp = C.POINTER(_capi.HuntResult)()
    _capi.check(L, L.dg_hunt_device(gpu_small.handle, C.byref(p), sl, 3, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_off.data_ptr()),
                                    len(qs), len(qb), 1, C.byref(rp)))
    R = rp.contents
    hb = device_bytes(R.d_hits, R.nhits * C.sizeof(_capi.Hit), torch.device("cuda", 0))
    host = bytes(C.string_at(C.addressof(R.hits.contents), R.nhits * C.sizeof(_capi.Hit)))
    assert R.nhits > 0 and bytes(hb.cpu().numpy().tobytes()) == host
    # the compact alignment description travels the same way: one 32-bit word per hit at distance 1
    assert R.ops_per_hit == 1 and R.d_ops
    ob = device_bytes(R.d_ops, R.nhits * 4, torch.device("cuda", 0))
    assert bytes(ob.cpu().numpy().tobytes()) == bytes(C.string_at(C.addressof(R.ops.contents), R.nhits * 4))
    assert not R.refalign  # rows exist only after dg_hunt_rows
    _capi.check(L, L.dg_hunt_rows(rp))
    assert R.refalign and R.aln_stride >= 20
    L.dg_hunt_result_free(rp)


def test_device_submit_keeps_two_batches_in_flight(gpu_small, small_genome):
    """dg_hunt_device_submit / dg_hunt_wait: six different batches resident in HBM through the handle's three lanes (submit k,
    collect k - 2), fetched hits identical to dg_hunt_device's of the same batch; a fourth submit before a wait is refused, and so
    is dg_hunt_device while a ticket is open."""
    import ctypes as C
    import torch
    from dicey_amd import _capi
    L = _capi.load()
    sl = (C.c_uint32 * 3)(*small_genome["seqlen"])
    p = _capi.HuntParams(1, 0, 0, 1000, 10000, 0, _capi.DG_HUNT_COMPACT)
    batches = []
    for b in range(6):
        qs = make_queries(70 + b, small_genome["text"], 120 + 17 * b, (14, 20, 24))
        qb = b"".join(q.encode() for q in qs)
        off = [0]
        for q in qs:
            off.append(off[-1] + len(q))
        batches.append((torch.frombuffer(bytearray(qb), dtype=torch.uint8).cuda(), torch.tensor(off, dtype=torch.int64).cuda(), len(qs), len(qb)))

    timeline = []

    def payload(rp):
        R = rp.contents
        timeline.append((R.t_base_gen, R.t_search_begin_ms, R.t_search_end_ms))
        words = 2 + R.ops_per_hit
        out = (R.nhits, bytes(C.string_at(R.chits, R.nhits * 4 * words)), bytes(C.string_at(R.qinfo, R.nq * 4)),
               bytes(C.string_at(C.cast(R.hit_off, C.c_void_p), (R.nq + 1) * 8)))
        L.dg_hunt_result_free(rp)
        return out

    want = []
    for d_q, d_off, n, nb in batches:
        rp = C.POINTER(_capi.HuntResult)()
        _capi.check(L, L.dg_hunt_device(gpu_small.handle, C.byref(p), sl, 3, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_off.data_ptr()), n, nb, 1, C.byref(rp)))
        want.append(payload(rp))
    assert sum(w[0] for w in want) > 0
    got, open_ = [], []
    for k, (d_q, d_off, n, nb) in enumerate(batches):
        tk = C.c_void_p()
        _capi.check(L, L.dg_hunt_device_submit(gpu_small.handle, C.byref(p), sl, 3, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_off.data_ptr()), n, nb, 1, C.byref(tk)))
        open_.append(tk)
        if k == 2:  # all three lanes taken
            t3 = C.c_void_p()
            assert L.dg_hunt_device_submit(gpu_small.handle, C.byref(p), sl, 3, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_off.data_ptr()), n, nb, 1,
                                           C.byref(t3)) != 0 and not t3.value
            rp = C.POINTER(_capi.HuntResult)()
            assert L.dg_hunt_device(gpu_small.handle, C.byref(p), sl, 3, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_off.data_ptr()), n, nb, 1, C.byref(rp)) != 0
        if len(open_) > 2:
            rp = C.POINTER(_capi.HuntResult)()
            _capi.check(L, L.dg_hunt_wait(open_.pop(0), C.byref(rp)))
            got.append(payload(rp))
    while open_:
        rp = C.POINTER(_capi.HuntResult)()
        _capi.check(L, L.dg_hunt_wait(open_.pop(0), C.byref(rp)))
        got.append(payload(rp))
    assert got == want
    # the lanes' common timeline: once the second lane exists every batch reports where its search kernel ran
    on_line = [t for t in timeline[len(batches) + 1:] if t[0]]
    assert len(on_line) >= len(batches) - 3 and all(0 <= b0 <= e0 for _, b0, e0 in on_line)
    assert max(e0 for _, _, e0 in on_line) > min(b0 for _, b0, _ in on_line)


def test_repeat_rich_strings_use_the_workgroup_locate(tmp_path):
    """A 20-mer present thousands of times: locate must return the max_locations SMALLEST positions in ascending order
    (hunter.h:355-357), through the radix-select kernel and through the per-lane fallback (take > 16384)."""
    import dicey_amd
    rng = random.Random(8)
    unit = "".join(rng.choice("ACGT") for _ in range(20))
    parts = []
    for i in range(6000):
        parts.append(unit)
        parts.append("".join(rng.choice("ACGT") for _ in range(rng.randint(3, 9))))
    seqs = ["".join(parts[:8000]), "".join(parts[8000:])]
    text = genome_text(seqs)
    path = str(tmp_path / "rep.fm9")
    O.build_fm9(text, path)
    g = {"seqlen": [len(s) + 1 for s in seqs], "names": ["r1", "r2"]}
    orc = O.Index(path)
    with dicey_amd.FmIndex(path) as ix:
        for kw in (dict(distance=0, max_locations=1000), dict(distance=1, max_locations=700), dict(distance=0, max_locations=100000),
                   dict(distance=1, hamming=True, max_locations=5000)):
            _compare(ix, orc, g, [unit, unit[:19] + ("A" if unit[19] != "A" else "C"), unit[1:] + "G"], **kw)


def test_flat_distance_two_kernel(gpu_small, small_genome, monkeypatch):
    """k_search2 (one workgroup per strand, lane per pair of operations, long presence filter): lengths around the table
    order and the filter order, up to the 30 nt it takes, mixed with queries it leaves to k_search (N, > 30 nt); the checker
    enumerates with its hash-set neighbours (tested equal to the literal ones in tests/test_oracle.py)."""
    orc = O.Index(small_genome["fm9"])
    O.fast_neighbors(True)
    try:
        qs = make_queries(411, small_genome["text"], 72, (10, 11, 12, 13, 14, 18, 19, 20, 20))
        qs += make_queries(412, small_genome["text"], 4, (21, 22, 24, 30))  # the checker needs seconds for each of these
        qs += ["ACGTNACGTAACGTACGTAC", small_genome["seqs"][0][700:733], "A" * 20, "AC" * 10, small_genome["seqs"][1][40:60].lower()]
        _compare(gpu_small, orc, small_genome, qs, distance=2)
        _compare(gpu_small, orc, small_genome, qs[:40], distance=2, max_locations=5, forward_only=True)
    finally:
        O.fast_neighbors(False)


# library switches that survive r04's pruning (DESIGN.md §6 lists them): each one is a code path of the shipped library and runs
# through the same matrix as the default
SWITCHES = {"nofuse": {"DICEY_NO_FUSED_SELECT": "1"}, "noband": {"DICEY_NO_BAND_VERIFY": "1"}, "classic": {"DICEY_CLASSIC_RESULTS": "1"},
            "ch4": {"DICEY_VERIFY_CH": "4"}, "ch8": {"DICEY_VERIFY_CH": "8"}, "caps": {"DICEY_DEBUG_CAPS": "3"},
            "lcap2": {"DICEY_FUSED_LCAP": "2"}, "noprep": {"DICEY_NO_PREP_FUSION": "1"}, "caphost": {"DICEY_CAP_HOST": "1"}, "nominima": {"DICEY_NO_SA_MINIMA": "1"}, "nopre5": {"DICEY_NO_PRE5": "1"},
            "nopre5d2": {"DICEY_NO_PRE5_D2": "1"}, "nofuse2": {"DICEY_NO_FUSED_SELECT2": "1"}, "noflatham2": {"DICEY_NO_FLAT_HAMMING2": "1"},
            "nonwin": {"DICEY_NO_N_WINDOW": "1"},
            "nolong2": {"DICEY_NO_LONG2": "1"}, "nodirectctx": {"DICEY_NO_DIRECT_CTX": "1"}}


@pytest.mark.parametrize("mode", ["no_table", "K8", "K11", "K13", "K9_nolong", "K9_long10", "K10_long14"] + ["K9_long10+" + k for k in SWITCHES])
def test_every_search_mode_gives_the_same_hits(small_genome, monkeypatch, mode):
    """Interval mode (no table), window mode with a table shorter than every query, and tables long enough that some
    queries fall back to interval mode (10/11-mers against K=11/13) must all reproduce the oracle — and so must every library
    switch on top of the usual layout."""
    import dicey_amd
    if "+" in mode:
        mode, sw = mode.split("+")
        for k, v in SWITCHES[sw].items():
            monkeypatch.setenv(k, v)
    if mode != "no_table":
        monkeypatch.setenv("DICEY_KMER_K", mode[1:].split("_")[0])
    if mode.endswith("_nolong"):  # what a device short of HBM gets: the table and its own filter only
        monkeypatch.setenv("DICEY_KMER_K2", "0")
    elif "_long" in mode:         # long filter right above the table order / well above it
        monkeypatch.setenv("DICEY_KMER_K2", mode.split("_long")[1])
    orc = O.Index(small_genome["fm9"])
    with open_index(small_genome["fm9"], kmer_table=(mode != "no_table")) as ix:
        qs = make_queries(31, small_genome["text"], 300, (10, 11, 14, 20, 33))
        _compare(ix, orc, small_genome, qs, distance=1)
        _compare(ix, orc, small_genome, qs[:120], distance=1, hamming=True)
        _compare(ix, orc, small_genome, [q[:12] for q in qs[:12]], distance=2)
        # Hamming distance 2 (r05: on k_search2p with "no edit" in place of the deletions; queries with N / above 30 nt stay on k_search;
        # lcap2 sends every group's leaves to the generic select kernels, nofuse2 all of them, noflatham2 keeps the r04 route)
        _compare(ix, orc, small_genome, qs[:150] + ["A" * 20, "ACGT" * 5, "AC" * 9], distance=2, hamming=True)
        _compare(ix, orc, small_genome, qs[150:220], distance=2, hamming=True, max_locations=2, forward_only=True)
        # queries with N: left of every table window (window mode + root split, r05), inside the window zone, at either end, several
        nq_ = []
        for i, q in enumerate(q for q in qs[100:160] if len(q) >= 14):
            p = [0, 1, 2, 3, len(q) // 2, len(q) - 3, len(q) - 1][i % 7]
            nq_.append(q[:p] + "N" + q[p + 1:])
        nq_ += [qs[0][:2] + "NN" + qs[0][4:], "N" + qs[1][1:-1] + "N"]
        _compare(ix, orc, small_genome, nq_, distance=1)
        _compare(ix, orc, small_genome, nq_[:20], distance=1, hamming=True)
        _compare(ix, orc, small_genome, nq_[:10], distance=2, hamming=True)
        if "_" in mode:  # the flat distance-2 kernel around the long filter's order (strings of 12..18 characters)
            O.fast_neighbors(True)
            try:
                _compare(ix, orc, small_genome, [q[:m] for q, m in zip(qs[200:232], [14, 15, 16, 12] * 8) if len(q) >= m], distance=2)
                # r05: queries whose two-deletion strings are shorter than the long filter's order (11-mers at K2 = 10 ... 15-mers at 14)
                # next to longer ones: k_search2p's LONG2 body leaves them out, the batch is repeated with the r04 body, and the
                # handle stays on it for the next batches (then tries LONG2 again)
                k2 = int(mode.split("_long")[1]) if "_long" in mode else 0
                if k2:
                    short = [q[:k2 + 1] for q in qs[232:240] if len(q) >= k2 + 1] + [q[:k2 + 4] for q in qs[240:248] if len(q) >= k2 + 4]
                    for _ in range(2):
                        _compare(ix, orc, small_genome, short, distance=2)
                    _compare(ix, orc, small_genome, [q[:k2 + 3] for q in qs[248:256] if len(q) >= k2 + 3], distance=2)
            finally:
                O.fast_neighbors(False)


def test_shared_handles_run_concurrently_and_agree(small_genome):
    """dg_index_share: two host threads, each with its own handle (stream + workspaces) on one resident index."""
    import threading
    import dicey_amd
    sc = small_genome
    qs = make_queries(9, sc["text"], 400, (20,))
    with dicey_amd.FmIndex(sc["fm9"]) as ix:
        want = ix.hunt(qs, sc["seqlen"])
        other = ix.share()
        got = [None, None]

        def run(k, h):
            for _ in range(5):
                got[k] = h.hunt(qs, sc["seqlen"])
        ths = [threading.Thread(target=run, args=(0, ix)), threading.Thread(target=run, args=(1, other))]
        for t in ths:
            t.start()
        for t in ths:
            t.join()
        key = lambda R: [[(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in q.hits] for q in R.queries]
        assert key(got[0]) == key(want) and key(got[1]) == key(want)
        other.close()


def test_randomised_configurations_against_oracle():
    """tools/fuzz_hunt.py: random parameter combinations, lengths, edits, repeats and non-DNA letters (found the N-aware cap bound)."""
    import subprocess
    import sys
    from conftest import ROOT
    # FUZZ_FAST_NEIGHBORS: the checker enumerates distance-2 neighbourhoods with its hash-set form (tested equal to the literal
    # restatement in tests/test_oracle.py); the literal form cost this test 95 of the suite's 770 s
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_hunt.py"), "7", "12"], capture_output=True, text=True, timeout=900,
                       env=dict(os.environ, FUZZ_FAST_NEIGHBORS="1"))
    assert r.returncode == 0, r.stderr[-1500:]
    assert "failing configurations: 0" in r.stdout, r.stdout[-1500:]


def test_device_entry_point_rechecks_a_cached_length_bound(gpu_small, small_genome):
    """dg_hunt_device skips reading the offsets back when buffer, count and byte total repeat; offsets rewritten in place
    with a longer query are caught on the device and the batch is redone."""
    import ctypes as C
    from dicey_amd import _capi
    L = _capi.load()
    s = small_genome["seqs"]
    qa = [s[0][100:120], s[1][200:220], s[2][300:320]]          # 20 + 20 + 20
    qb = [s[0][100:112], s[1][200:236], s[2][300:312]]          # 12 + 36 + 12: same count, same byte total
    sl = (C.c_uint32 * 3)(*small_genome["seqlen"])
    p = _capi.HuntParams(1, 0, 0, 1000, 10000)
    d_q = torch.zeros(60, dtype=torch.uint8, device="cuda")
    d_off = torch.zeros(4, dtype=torch.int64, device="cuda")

    def run(qs):
        d_q.copy_(torch.frombuffer(bytearray("".join(qs).encode()), dtype=torch.uint8))
        off = [0]
        for q in qs:
            off.append(off[-1] + len(q))
        d_off.copy_(torch.tensor(off, dtype=torch.int64))
        torch.cuda.synchronize()
        rp = C.POINTER(_capi.HuntResult)()
        _capi.check(L, L.dg_hunt_device(gpu_small.handle, C.byref(p), sl, 3, C.c_void_p(d_q.data_ptr()), C.c_void_p(d_off.data_ptr()),
                                        3, 60, 1, C.byref(rp)))
        R = rp.contents
        out = [(R.hits[i].query, R.hits[i].chr, R.hits[i].start, R.hits[i].score) for i in range(R.nhits)]
        L.dg_hunt_result_free(rp)
        return out

    first = run(qa)
    assert run(qa) == first and len(first) >= 3          # second call: cached bound, same answer
    want = [(i, h.chr, h.start, h.score) for i, q in enumerate(gpu_small.hunt(qb, small_genome["seqlen"]).queries) for h in q.hits]
    assert run(qb) == want and any(h[0] == 1 for h in want)   # the 36-mer exceeds the cached 20: detected, redone


def test_submit_wait_on_two_handles_equals_the_blocking_call(gpu_small, small_genome):
    """dg_hunt_submit / dg_hunt_wait: three batches in flight on ONE handle (ABI 5: the library's internal lanes) and, as in
    ABI 4, on a second handle of the same resident index; a fourth submit on a handle with three in flight is refused, a blocking
    call on a handle with a batch in flight as well; results equal dg_hunt's in both result forms."""
    import dicey_amd
    g = small_genome
    qa = make_queries(31, g["text"], 400)
    qb = make_queries(32, g["text"], 300)
    qc = make_queries(33, g["text"], 200)
    key = lambda R: [[(h.score, h.chr, h.start, h.strand, h.refalign, h.queryalign) for h in q.hits] + [q.flags, q.nondna, q.sequence, q.distance]
                     for q in R.queries]
    want_a, want_b = key(gpu_small.hunt(qa, g["seqlen"], distance=1)), key(gpu_small.hunt(qb, g["seqlen"], distance=2))
    want_c = key(gpu_small.hunt(qc, g["seqlen"], distance=1, hamming=True))
    assert want_a == key(gpu_small.hunt(qa, g["seqlen"], distance=1, compact=False))
    other = gpu_small.share()
    try:
        for rnd in range(3):
            cpt = rnd != 1
            ta = gpu_small.hunt_submit(qa, g["seqlen"], distance=1, compact=cpt)
            tb = gpu_small.hunt_submit(qb, g["seqlen"], distance=2, compact=cpt)      # second lane of the same handle
            te = gpu_small.hunt_submit(qc, g["seqlen"], distance=1, hamming=True, compact=cpt)  # third lane (r04)
            tc = other.hunt_submit(qc, g["seqlen"], distance=1, hamming=True, compact=cpt, max_query_len=64)
            with pytest.raises(dicey_amd.DgError):
                gpu_small.hunt_submit(qb, g["seqlen"], distance=1)  # three batches per handle
            with pytest.raises(dicey_amd.DgError):
                gpu_small.hunt(qb, g["seqlen"], distance=1)         # the blocking call needs an idle handle
            assert key(gpu_small.hunt_wait(ta)) == want_a
            td = gpu_small.hunt_submit(qc, g["seqlen"], distance=1, hamming=True, compact=cpt)  # lane of `ta` is free again
            assert key(gpu_small.hunt_wait(tb)) == want_b
            assert key(gpu_small.hunt_wait(te)) == want_c
            assert key(other.hunt_wait(tc)) == want_c
            assert key(gpu_small.hunt_wait(td)) == want_c
        with pytest.raises(dicey_amd.DgError):  # a bound that does not hold fails the batch loudly
            gpu_small.hunt(qa, g["seqlen"], distance=1, max_query_len=12)
    finally:
        other.close()


@pytest.mark.parametrize("lcap", [None, "3", "0"])
def test_fused_select_and_its_hand_over_to_the_generic_kernels(small_genome, monkeypatch, lcap):
    """Distance 1: k_search1s settles the select stage inside the search kernel and the generic kernels (k_search for queries
    with N / above 31 nt, scan, pack, alive, rank) are left out once a batch had no work for them.  A later batch that needs them
    is repeated with them; a workgroup whose strings overflow its LDS list (forced here with DICEY_FUSED_LCAP) hands its groups
    over the same way.  Every batch must equal the checker, whatever the handle saw before."""
    import dicey_amd
    if lcap is not None:
        monkeypatch.setenv("DICEY_FUSED_LCAP", lcap)
    g = small_genome
    orc = O.Index(g["fm9"])
    rng = random.Random(404)
    pure = make_queries(41, g["text"], 600, (20,))
    pure = [q for q in pure if set(q) <= set("ACGT")]
    low = ["A" * 20, "AC" * 10, "ACG" * 7, g["seqs"][0][100:120], "T" * 19 + "G"] * 3
    mixed = pure[:150] + ["ACGTNACGTACGTACGTACG", g["seqs"][1][50:90], "acgtacgtacgtacgtacgtnn", g["seqs"][2][7:19]] + pure[150:200]
    rng.shuffle(mixed)
    monkeypatch.setenv("DICEY_KMER_K", "9")
    with open_index(g["fm9"]) as ix:
        for qs, kw in [(pure, dict(distance=1)), (pure[:300], dict(distance=1)), (mixed, dict(distance=1)), (pure[300:], dict(distance=1)),
                       (pure[:200], dict(distance=1, hamming=True)), (low, dict(distance=1)), (mixed, dict(distance=1, hamming=True)),
                       (pure[:100], dict(distance=1, max_locations=2)), (pure[:64], dict(distance=1, forward_only=True))]:
            _compare(ix, orc, g, qs, **kw)
        # edit distance 2 on the same handle (r04: the select stage inside k_search2p; DICEY_FUSED_LCAP lowers its list's capacity too,
        # so the lcap forms hand every group over to the generic select kernels)
        O.fast_neighbors(True)
        try:
            short = [q[:m] for q, m in zip(pure[:24], [14, 16, 18, 20] * 6)]
            _compare(ix, orc, g, short + ["ACGTNACGTACGTACG"], distance=2)
        finally:
            O.fast_neighbors(False)
        # count mode (padlock.h:396-421) goes through the same kept strings
        rc = lambda q: q[::-1].translate(str.maketrans("ACGT", "TGCA"))
        for q, (gf, gr) in zip(pure[:40], ix.neighborhood_count([q.encode() for q in pure[:40]], distance=1)):
            assert gf == sum(orc.count(s.encode()) for s in O.neighbors(q, 1, True)), q
            assert gr == sum(orc.count(s.encode()) for s in O.neighbors(rc(q), 1, True)), q


def test_decreasing_offsets_fail_loudly_even_with_a_length_bound(gpu_small, small_genome):
    """with dg_hunt_params::max_query_len the host used to skip its pass over the offsets and k_prepare ran its byte loop
    over qoff[q + 1] - qoff[q] wrapped to ~4e9.  Host buffers (dg_hunt), device buffers (dg_hunt_device, with and without the bound,
    cap-prone distance included): DG_EINVAL each time, and the resident index still answers the next batch correctly."""
    import ctypes as C
    import numpy as np
    import dicey_amd
    from dicey_amd import _capi
    L = _capi.load()
    g = small_genome
    qs = make_queries(77, g["text"], 64, (20,))
    want = [[(h.score, h.chr, h.start, h.strand) for h in q.hits] for q in gpu_small.hunt(qs, g["seqlen"], distance=1).queries]
    qbytes = "".join(qs).encode()
    off = np.zeros(len(qs) + 1, dtype=np.uint64)
    off[1:] = np.cumsum([len(q) for q in qs])
    sl = (C.c_uint32 * len(g["seqlen"]))(*g["seqlen"])
    bad_sets = []
    o1 = off.copy(); o1[10], o1[11] = o1[11], o1[10]                    # one decreasing pair in the middle
    bad_sets.append(o1)
    o2 = off.copy(); o2[5] = np.uint64(1) << np.uint64(40)              # leaves the buffer (and decreases right behind)
    bad_sets.append(o2)
    o3 = off.copy(); o3[-1] = o3[-1] - np.uint64(3)                     # does not end at total_qbytes (device path only: the host path takes qoff[nq] as the total)
    for bound, dist in ((22, 1), (0, 1), (22, 2), (64, 2)):
        for oi, o in enumerate(bad_sets + [o3]):
            p = _capi.HuntParams(dist, 0, 0, 1000, 10000, bound, _capi.DG_HUNT_COMPACT)
            oa = (C.c_uint64 * len(o))(*[int(x) for x in o])
            rp = C.POINTER(_capi.HuntResult)()
            if oi < 2:
                rc = L.dg_hunt(gpu_small.handle, C.byref(p), sl, len(g["seqlen"]), qbytes, oa, len(qs), C.byref(rp))
                assert rc != 0 and not rp, (bound, dist, oi, rc)
            if torch is not None:
                dq = torch.frombuffer(bytearray(qbytes), dtype=torch.uint8).cuda()
                do = torch.from_numpy(o.view(np.int64).copy()).cuda()
                torch.cuda.synchronize()
                rp = C.POINTER(_capi.HuntResult)()
                rc = L.dg_hunt_device(gpu_small.handle, C.byref(p), sl, len(g["seqlen"]), C.c_void_p(dq.data_ptr()), C.c_void_p(do.data_ptr()),
                                      len(qs), len(qbytes), 0, C.byref(rp))
                assert rc != 0 and not rp, (bound, dist, oi, rc)
    assert [[(h.score, h.chr, h.start, h.strand) for h in q.hits] for q in gpu_small.hunt(qs, g["seqlen"], distance=1, max_query_len=22).queries] == want
    with pytest.raises(Exception):   # a bound that does not hold is refused on the host, before anything is uploaded
        gpu_small.hunt(qs, g["seqlen"], distance=1, max_query_len=19)


def test_one_shot_open_flags_answer_the_same(small_genome):
    """DG_OPEN_COMPACT (no suffix-array records with context, no prefix levels) and DG_OPEN_NO_PRE5 (narrow table intervals extended
    through the Occ blocks instead of filtered by the preceding-characters array) — what `dicey hunt` passes for one input — against
    the oracle: distance 1 and 2, edit and Hamming, and the footprint really is smaller."""
    import dicey_amd
    with dicey_amd.FmIndex(small_genome["fm9"], device=0) as full, dicey_amd.FmIndex(small_genome["fm9"], device=0, compact=True, pre5=False) as lean:
        assert lean.stats()["hbm_bytes"] < full.stats()["hbm_bytes"] - 2 * len(small_genome["text"])
        for kw, nq, lens in [(dict(distance=1), 600, (20, 18, 25)), (dict(distance=2), 24, (20,)), (dict(distance=1, hamming=True), 200, (20,)),
                             (dict(distance=1, max_locations=3), 200, (12, 16, 17))]:
            test_hunt_hits_equal_oracle_push_order(lean, small_genome, kw, nq, lens)
