"""A13 hardening: the .fm9 reader against files it did not write.  No genuine `dicey index` file exists offline
(the layout in dicey_amd/csrc/sdsl_file.hpp is restated from sdsl-lite), so what can be tested is the DEFENCE:
every perturbation of a section — wrong width, missing select support, another child order of the Huffman tree, truncation, trailing
bytes, a rank support of another shape ... — must fail with DG_EFORMAT and a message that NAMES the section, and none may pass.
dg_fm9_check runs on the host (no GPU): reference src/index.h:121-122 (store_to_checked_file), src/hunter.h:253-256 (load)."""
import json
import os
import struct

import pytest

import oracle_lib as O
from conftest import make_genome, genome_text

DG_EFORMAT = -3


@pytest.fixture(scope="module")
def fm9(tmp_path_factory):
    seqs = make_genome(55, 2, 6000, iupac=True)
    path = str(tmp_path_factory.mktemp("acc") / "g.fm9")
    O.build_fm9(genome_text(seqs), path)
    return path


def _check(path, deep=True):
    import dicey_amd
    return dicey_amd.check_fm9(path, deep=deep)


def _sections(rep):
    return {s["name"]: (s["offset"], s["bytes"]) for s in rep["sections"]}


def test_a_file_of_the_writers_passes_and_every_byte_is_accounted_for(fm9):
    rep = _check(fm9)
    assert rep["ok"] is True and rep["rc"] == 0 and rep["layout"] == "store_to_checked_file"
    secs = rep["sections"]
    assert [s["name"] for s in secs] == ["wt header (size, sigma)", "wt bit_vector", "wt rank_support_v", "wt select_support_mcl<1>",
                                         "wt select_support_mcl<0>", "wt byte_tree nodes", "wt byte_tree c_to_leaf / path", "sa_samples",
                                         "isa_samples", "alphabet char2comp", "alphabet comp2char", "alphabet C", "alphabet sigma"]
    at = 8
    for s in secs:   # contiguous from behind the class hash to the end of the file
        assert s["offset"] == at
        at += s["bytes"]
    assert at == os.path.getsize(fm9) == rep["file_bytes"]
    # the plain store_to_file layout (no hash) is the same object
    plain = fm9 + ".plain"
    open(plain, "wb").write(open(fm9, "rb").read()[8:])
    rp = _check(plain)
    assert rp["ok"] is True and rp["layout"] == "store_to_file" and rp["n"] == rep["n"]


def _perturbations(data, secs, n, sigma):
    """(label, bytes of the perturbed file, words the message must contain)"""
    d = bytearray(data)
    out = []

    def mod(label, must, fn):
        b = bytearray(d)
        r = fn(b)
        out.append((label, bytes(r if r is not None else b), must))

    o_bv, l_bv = secs["wt bit_vector"]
    o_rk, l_rk = secs["wt rank_support_v"]
    o_s1, l_s1 = secs["wt select_support_mcl<1>"]
    o_s0, l_s0 = secs["wt select_support_mcl<0>"]
    o_nd, l_nd = secs["wt byte_tree nodes"]
    o_cp, l_cp = secs["wt byte_tree c_to_leaf / path"]
    o_sa, l_sa = secs["sa_samples"]
    o_isa, l_isa = secs["isa_samples"]
    o_c2c, _ = secs["alphabet comp2char"]
    o_C, l_C = secs["alphabet C"]
    o_sig, _ = secs["alphabet sigma"]
    mod("truncated inside the bit vector", ["wt bit_vector", "beyond the end"], lambda b: b[:o_bv + l_bv // 2])
    mod("truncated inside the samples", ["sa_samples", "beyond the end"], lambda b: b[:o_sa + l_sa // 2])
    mod("trailing bytes", ["trailing bytes"], lambda b: b + b"\0" * 16)
    mod("one byte missing at the end", ["alphabet sigma"], lambda b: b[:-1])

    def width(b, off, w):
        b[off + 8] = w
    mod("sa_samples announced with another width", ["sa_samples"], lambda b: width(b, o_sa, 64))
    mod("sa_samples narrower than the values need", ["sa_samples", "width"], lambda b: width(b, o_sa, 8))
    mod("isa_samples with another width than sa_samples", ["isa_samples"], lambda b: width(b, o_isa, b[o_sa + 8] + 1))

    def swapped_samples(b):   # written in the other order: isa first
        sa, isa = bytes(b[o_sa:o_sa + l_sa]), bytes(b[o_isa:o_isa + l_isa])
        b[o_sa:o_isa + l_isa] = isa + sa
    mod("isa_samples in front of sa_samples", ["sa_samples"], swapped_samples)
    mod("select supports absent from the file", ["select_support_mcl<1>|byte_tree nodes"], lambda b: b[:o_s1] + b[o_nd:])
    mod("only one select support present", ["select_support_mcl<0>|byte_tree nodes"], lambda b: b[:o_s0] + b[o_nd:])

    def node_field(b, v, which):   # node v: bv_pos u64, bv_pos_rank u64, parent u16, child0 u16, child1 u16
        return o_nd + 8 + 22 * v + {"pos": 0, "rank": 8, "parent": 16, "c0": 18, "c1": 20}[which]

    def swap_children(b):
        a, c = node_field(b, 0, "c0"), node_field(b, 0, "c1")
        b[a:a + 2], b[c:c + 2] = b[c:c + 2], b[a:a + 2]
    mod("the root's children in the other order", ["c_to_leaf / path", "child order"], swap_children)

    def bad_rank_offset(b):
        for v in range(2 * sigma - 1):
            if struct.unpack_from("<H", b, node_field(b, v, "c0"))[0] != 0xFFFF and v:
                off = node_field(b, v, "rank")
                struct.pack_into("<Q", b, off, struct.unpack_from("<Q", b, off)[0] + 1)
                return
    mod("an inner node's bv_pos_rank off by one", ["byte_tree nodes", "bv_pos_rank"], bad_rank_offset)

    def node_count(b):
        struct.pack_into("<Q", b, o_nd, 2 * sigma)
    mod("another number of tree nodes", ["byte_tree nodes", "nodes"], node_count)

    def leaf_symbol(b):
        for v in range(2 * sigma - 1):
            if struct.unpack_from("<H", b, node_field(b, v, "c0"))[0] == 0xFFFF:
                off = node_field(b, v, "rank")
                struct.pack_into("<Q", b, off, struct.unpack_from("<Q", b, off)[0] ^ 1)
                return
    mod("a leaf that carries another symbol", ["c_to_leaf / path", "carries symbol"], leaf_symbol)

    def rank_super(b):   # absolute count of the second superblock off by one
        off = o_rk + 8 + 16
        struct.pack_into("<Q", b, off, struct.unpack_from("<Q", b, off)[0] + 1)
    mod("a rank superblock count off by one", ["rank_support_v", "superblock 1"], rank_super)

    def rank_fields_reversed(b):   # the seven 9-bit fields of superblock 0 in the opposite order
        off = o_rk + 8 + 8
        w = struct.unpack_from("<Q", b, off)[0]
        f = [(w >> (63 - 9 * j)) & 0x1FF for j in range(1, 8)]
        w2 = 0
        for j, v in zip(range(1, 8), reversed(f)):
            w2 |= v << (63 - 9 * j)
        struct.pack_into("<Q", b, off, w2)
    mod("the 9-bit fields of a rank word in the other order", ["rank_support_v", "9-bit field"], rank_fields_reversed)
    mod("half of the rank words missing", ["rank_support_v|bit_vector|select"], lambda b: b[:o_rk] + struct.pack("<Q", (l_rk - 8) * 4) + b[o_rk + 8:o_rk + 8 + (l_rk - 8) // 2] + b[o_rk + l_rk:])

    def flip_bv(b):
        b[o_bv + 8 + 3] ^= 0x10
    mod("one bit of the bit vector flipped", ["rank_support_v"], flip_bv)

    def sigma_field(b):
        struct.pack_into("<H", b, o_sig, sigma + 1)
    mod("alphabet sigma differs from the tree's", ["alphabet sigma"], sigma_field)

    def c_total(b):
        off = o_C + 8 + 8 * sigma
        struct.pack_into("<Q", b, off, n + 1)
    mod("C[sigma] is not the text size", ["alphabet C"], c_total)

    def c_shift(b):   # one symbol more of comp 1, one less of comp 2: totals still add up, the leaves disagree
        off = o_C + 8 + 8 * 2
        struct.pack_into("<Q", b, off, struct.unpack_from("<Q", b, off)[0] + 1)
    mod("C[] disagrees with the symbol totals of the tree", ["alphabet C", "leaf"], c_shift)

    def comp_order(b):
        b[o_c2c + 8 + 1], b[o_c2c + 8 + 2] = b[o_c2c + 8 + 2], b[o_c2c + 8 + 1]
    mod("comp2char not in byte order", ["alphabet comp2char"], comp_order)

    def sa_first(b):   # entry 0 of sa_samples must be n - 1: overwrite its low byte
        b[o_sa + 9] ^= 0x01
    mod("sa_samples[0] is not the sentinel's position", ["sa_samples", "entry 0"], sa_first)

    def isa_entry(b):
        wd = b[o_isa + 8]
        v = int.from_bytes(b[o_isa + 9:o_isa + 9 + 8], "little")
        v = (v & ~((1 << wd) - 1)) | ((v + 32) & ((1 << wd) - 1))   # ISA[0] moved by 32 ranks: still a multiple-of-32 rank if it was one
        b[o_isa + 9:o_isa + 9 + 8] = v.to_bytes(8, "little")
    out.append(("isa_samples inconsistent with sa_samples (if entry 0 lands on a sample)", None, ["isa_samples"]))  # placeholder, see below
    out.pop()

    def leaf_of_internal(b):
        for ch in range(256):
            lf = struct.unpack_from("<H", b, o_cp + 2 * ch)[0]
            if lf != 0xFFFF:
                struct.pack_into("<H", b, o_cp + 2 * ch, 0)   # the root
                return
    mod("c_to_leaf names an inner node", ["c_to_leaf / path", "not a leaf"], leaf_of_internal)
    return out


def test_every_perturbed_file_is_refused_by_section_name(fm9, tmp_path):
    rep = _check(fm9)
    data = open(fm9, "rb").read()
    cases = _perturbations(data, _sections(rep), rep["n"], rep["sigma"])
    assert len(cases) >= 20
    seen = set()
    for i, (label, blob, must) in enumerate(cases):
        assert blob != data, label
        p = str(tmp_path / ("p%02d.fm9" % i))
        open(p, "wb").write(blob)
        r = _check(p)
        assert r["ok"] is False and r["rc"] == DG_EFORMAT, (label, r)
        err = r["error"]
        assert "section '" in err, (label, err)
        for m in must:
            assert any(alt in err for alt in m.split("|")), (label, m, err)
        seen.add(label)
    assert len(seen) == len(cases)


def test_isa_samples_are_held_against_sa_samples(fm9, tmp_path):
    """ISA[64 k] = r with r a multiple of 32 must meet sa_samples[r / 32] = 64 k: perturb such an entry"""
    rep = _check(fm9)
    secs = _sections(rep)
    data = bytearray(open(fm9, "rb").read())
    o_isa, l_isa = secs["isa_samples"]
    o_sa, _ = secs["sa_samples"]
    wd = data[o_isa + 8]
    words = int.from_bytes(data[o_isa + 9:o_isa + l_isa], "little")
    nisa = (int.from_bytes(data[o_isa:o_isa + 8], "little")) // wd
    sa_words = int.from_bytes(data[o_sa + 9:secs["isa_samples"][0]], "little")
    hit = None
    for k in range(nisa):
        r = (words >> (k * wd)) & ((1 << wd) - 1)
        if r % 32 == 0:
            assert (sa_words >> ((r // 32) * wd)) & ((1 << wd) - 1) == 64 * k
            hit = (k, r)
            break
    assert hit is not None
    k, r = hit
    r2 = r + 32 if r + 32 < rep["n"] else r - 32
    words = (words & ~(((1 << wd) - 1) << (k * wd))) | (r2 << (k * wd))
    data[o_isa + 9:o_isa + l_isa] = words.to_bytes(l_isa - 9, "little")
    p = str(tmp_path / "isa.fm9")
    open(p, "wb").write(bytes(data))
    rr = _check(p)
    assert rr["ok"] is False and "isa_samples" in rr["error"] and "sa_samples[" in rr["error"]
    assert _check(p, deep=False)["ok"] is True   # (the shallow check reads no sample values: that is what `deep` is for)


def test_the_product_writer_and_the_checker_agree(tmp_path):
    """dicey_amd/csrc/sdsl_writer.hpp (the GPU builder's writer) is exercised on the GPU box; here: the oracle writer's files of
    several shapes — one symbol short of a full byte alphabet, a single sequence, N-rich — all pass the deep check"""
    for seed, nchr, ln, kw in ((1, 1, 3000, {}), (2, 3, 1500, {"nrate": 0.05}), (3, 2, 700, {"iupac": True, "repeats": False})):
        p = str(tmp_path / ("w%d.fm9" % seed))
        O.build_fm9(genome_text(make_genome(seed, nchr, ln, **kw)), p)
        r = _check(p)
        assert r["ok"] is True, r
