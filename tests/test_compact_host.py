"""ABI 5 compact results on the host side (include/dicey_gpu.h: dg_chit_unpack, dg_normalize_query, dg_hunt_expand): a hand-built
compact result — text positions, packed words, operation words, per-query words — must expand to the dg_hit records, flags and
normalised sequences it encodes.  Host code only: runs without a GPU."""
import ctypes as C

import pytest

from dicey_amd import _capi

MISMATCH, REF_GAP, QUERY_GAP, NONE = 0, 1, 2, 0xFFFFFFFF


@pytest.fixture(scope="module")
def lib():
    return _capi.load()


def meta(score, strand, delta, aln_len):
    return ((-score) & 15) | ((1 if strand == "-" else 0) << 4) | (((delta + 32) & 127) << 5) | (aln_len << 16)


def test_normalize_query(lib):
    for raw, want, bad in [(b"acgtACGT", b"ACGTACGT", 0), (b"ACNNRYacgu", b"ACNNNNACGN", 5), (b"", b"", 0), (b"nnnn", b"NNNN", 4)]:
        out = C.create_string_buffer(len(raw) + 1)
        n = C.c_uint32(99)
        assert lib.dg_normalize_query(raw, len(raw), out, C.byref(n)) == 0
        assert out.raw[:len(raw)] == want and n.value == bad


def test_expand_rebuilds_hits_flags_and_sequences(lib):
    seq_start = [0, 1001, 3002]  # three sequences of 1000, 2000, ... characters + separators
    queries = [b"ACGTACGTACGTACGTACGT", b"acgtnacgtacgtacgtacg", b"ACGTAC", b"TTTTTTTTTTTTTTTTTTTT"]
    # (query, text position, score, strand, delta, aln_len, ops)
    hits = [(0, 10, 0, "+", -1, 20, [NONE, NONE]),
            (0, 1001 + 5, -1, "-", 0, 20, [7 | (MISMATCH << 16) | (ord("G") << 24), NONE]),
            (1, 3002 + 77, -2, "+", 2, 21, [3 | (QUERY_GAP << 16) | (ord("A") << 24), 9 | (MISMATCH << 16) | (ord("T") << 24)]),
            (3, 1000, 0, "-", 0, 20, [NONE, NONE])]  # position 1000 is the separator of sequence 0: still sequence 0 (hunter.h:358-362)
    oph, W = 2, 4
    nq, nh = len(queries), len(hits)
    chits = (C.c_uint32 * (nh * W))()
    hit_off = (C.c_uint64 * (nq + 1))()
    for h, (q, pos, sc, st, dl, al, ops) in enumerate(hits):
        chits[h * W:h * W + W] = [pos, meta(sc, st, dl, al)] + ops
        hit_off[q + 1] += 1
    for i in range(nq):
        hit_off[i + 1] += hit_off[i]
    qinfo = (C.c_uint32 * nq)(0 | (1 << 8), 0 | (2 << 8) | (1 << 16), _capi.DG_Q_TOO_SHORT | (1 << 8), _capi.DG_Q_MAX_MATCHES | (1 << 8))
    ss = (C.c_uint64 * 3)(*seq_start)
    R = _capi.HuntResult()
    R.nq, R.nhits, R.ops_per_hit, R.compact, R.nseq = nq, nh, oph, 1, 3
    R.hit_off = C.cast(hit_off, C.POINTER(C.c_uint64))
    R.chits = C.cast(chits, C.POINTER(C.c_uint32))
    R.qinfo = C.cast(qinfo, C.POINTER(C.c_uint32))
    R.seq_start = C.cast(ss, C.POINTER(C.c_uint64))
    buf = b"".join(queries)
    off = (C.c_uint64 * (nq + 1))()
    for i, q in enumerate(queries):
        off[i + 1] = off[i] + len(q)
    # one hit through dg_chit_unpack
    H = _capi.Hit()
    ops = C.POINTER(C.c_uint32)()
    assert lib.dg_chit_unpack(C.byref(R), 1, 0, C.byref(H), C.byref(ops)) == 0
    assert (H.score, H.chr, H.start, H.query, H.aln_len, chr(H.strand)) == (-1, 1, 5 + 0 + 1, 0, 20, "-") and ops[0] == hits[1][6][0]
    assert lib.dg_chit_unpack(C.byref(R), nh, 0, C.byref(H), None) != 0  # out of range
    # wrong bytes for this batch are refused (the replaced-character counts do not match)
    assert lib.dg_hunt_expand(C.byref(R), b"".join(q.upper().replace(b"N", b"A") for q in queries), off) != 0
    assert lib.dg_hunt_expand(C.byref(R), buf, off) == 0
    got = [(R.hits[h].query, R.hits[h].chr, R.hits[h].start, R.hits[h].score, chr(R.hits[h].strand), R.hits[h].aln_len) for h in range(nh)]
    assert got == [(0, 0, 10, 0, "+", 20), (0, 1, 6, -1, "-", 20), (1, 2, 80, -2, "+", 21), (3, 0, 1001, 0, "-", 20)]
    assert [R.ops[i] for i in range(nh * oph)] == [x for h in hits for x in h[6]]
    assert [R.qflags[i] for i in range(nq)] == [0, 0, _capi.DG_Q_TOO_SHORT, _capi.DG_Q_MAX_MATCHES]
    assert [R.qdistance[i] for i in range(nq)] == [1, 2, 1, 1] and [R.qnondna[i] for i in range(nq)] == [0, 1, 0, 0]
    assert C.string_at(R.qseq, off[nq]) == b"ACGTACGTACGTACGTACGT" + b"ACGTNACGTACGTACGTACG" + b"ACGTAC" + b"T" * 20
    assert lib.dg_hunt_expand(C.byref(R), buf, off) == 0  # idempotent
    import ctypes.util
    C.CDLL(ctypes.util.find_library("c")).free(C.c_void_p(R.expanded_))
