"""The algebra behind k_search2p<., LONG2 = true> (dicey_amd/csrc/hunt_search.hpp, r05), checked on the CPU with a Python model.

The kernel never builds the 64 twice-edited strings of a pair of positions.  It computes, per lane, nine base places in the long
filter (first operation deletion / substitution / insertion x the same for the second) and the images of the edited characters' two
code bits, and ORs them per probe.  That is only right if
  (1) the place of a window code's bit in a copy of the filter (word offset, bit number) is a permutation of the code's bits, so that
      place(a | b) == place(a) | place(b) for codes without common bits (filt_pos), and
  (2) the window code of apply_edit(apply_edit(q, p1, op1), p2, op2) — the last K2 characters of the twice-edited string, the
      enumeration of the reference's neighbors.h:47-83 at distance 2 — equals base(kind1, kind2) | character 1 << slot 1 |
      character 2 << slot 2, masked to the window.
This file restates both sides in Python (the formulas, operand for operand, of `filt_pos`, `apply_edit`, `edit_string` and the
lambdas `kind` / `batch`) and compares them for every pair of positions, every operation pair and every copy of the filter on random
and low-complexity queries — edit and Hamming mode.  The GPU parity tests hold the kernel itself against the oracle; this one holds
the derivation, and needs no GPU.
"""
import random

import pytest

K2 = 18
MASK2 = (1 << (2 * K2)) - 1
FIELDS = (0, 9, 18, 27)  # first code bit of the in-line field of the four copies


def pack(q):
    v = 0
    for ch in q:
        v = (v << 2) | "ACGT".index(ch)
    return v


def apply_edit(pk, ln, pos, op, ham=False):
    """hunt_search.hpp apply_edit / apply_edit_h: op 0 deletion (Hamming: none), 1-3 the other three bases, 4-7 insert A, C, G, T"""
    R = ln - pos
    low = pk & ((1 << (2 * R)) - 1)
    old = (pk >> (2 * R)) & 3
    if op == 0:
        if ham:
            return pk, ln
        return low | ((pk >> (2 * R + 2)) << (2 * R)), ln - 1
    if op < 4:
        c = (old + op) & 3
        return pk ^ ((old ^ c) << (2 * R)), ln
    c = op - 4
    return low | (c << (2 * R)) | ((pk >> (2 * R)) << (2 * R + 2)), ln + 1


def edit_string(pk, ln, pos, op, ham=False):
    """hunt_search.hpp edit_string: the three kinds in one expression"""
    R2 = 2 * (ln - pos)
    dele, ins = (op == 0 and not ham), op >= 4
    old = (pk >> R2) & 3
    c = op - 4 if ins else (old + op) & 3
    left = pk >> (R2 if ins else R2 + 2)
    oln = ln - 1 if dele else ln + 1 if ins else ln
    return (pk & ((1 << R2) - 1)) | (0 if dele else c << R2) | (left << (R2 if dele else R2 + 2)), oln


def filt_pos(w, s):
    """(word offset inside the copy, bit number) of window code w in the copy whose in-line field starts at code bit s"""
    inl = (w >> s) & 511
    off = ((w >> (s + 9)) << (s + 4)) | ((w & ((1 << s) - 1)) << 4) | (inl >> 5)
    return off, inl & 31


def direct_place(s2, s):
    return filt_pos(s2 & MASK2, s)


def slot_img(bpos, s):
    lo = filt_pos((1 << bpos) & MASK2, s)
    both = filt_pos((3 << bpos) & MASK2, s)
    return (lo[0], both[0] ^ lo[0]), (lo[1], both[1] ^ lo[1])


def img_of(c, i2):
    return (i2[0] if c & 1 else 0) | (i2[1] if c & 2 else 0)


def kind(qpk, m, p1, p2, k1, s, ham):
    """the lambda `kind`: base places and images for the second operation after a first operation of shape k1"""
    b1 = 2 * (m - p1)
    low1 = qpk & ((1 << b1) - 1)
    if k1 == 0:
        s1 = qpk if ham else low1 | ((qpk >> (b1 + 2)) << b1)
        l1 = m if ham else m - 1
    elif k1 == 1:
        s1, l1 = qpk & ~(3 << b1), m
    else:
        s1, l1 = low1 | ((qpk >> b1) << (b1 + 2)), m + 1
    b2 = 2 * (l1 - p2)
    low2 = s1 & ((1 << b2) - 1)
    q2a = (qpk >> (2 * (m - p2))) & 3
    pd = filt_pos((s1 if ham else low2 | ((s1 >> (b2 + 2)) << b2)) & MASK2, s)
    ps = filt_pos(s1 & ~(3 << b2) & MASK2, s)
    pi = filt_pos((low2 | ((s1 >> b2) << (b2 + 2))) & MASK2, s)
    io, ib = slot_img(b2, s)
    so = [0] + [img_of((q2a + op2) & 3, io) for op2 in (1, 2, 3)]
    bits0 = pd[1]
    for op2 in (1, 2, 3):
        bits0 |= (ps[1] | img_of((q2a + op2) & 3, ib)) << (8 * op2)
    bits1 = (pi[1] * 0x01010101) | (ib[0] << 8) | (ib[1] << 16) | ((ib[0] | ib[1]) << 24)
    return {"boff": (pd[0], ps[0], pi[0]), "so": so, "io": io, "bits": (bits0, bits1)}


def or_place(qpk, m, p1, op1, p2, op2, s, ham):
    """the lambda `batch`: a probe's place from the parts"""
    k1 = 0 if op1 == 0 else 2 if op1 >= 4 else 1
    qa = (qpk >> (2 * (m - p1))) & 3
    e1o, e1b = slot_img(2 * (m - p1), s)
    c1 = (qa + op1) & 3 if k1 == 1 else op1 - 4 if k1 == 2 else 0
    i1o, i1b = img_of(c1, e1o), img_of(c1, e1b)
    kk = kind(qpk, m, p1, p2, k1, s, ham)
    k2 = 0 if op2 == 0 else 1 if op2 < 4 else 2
    i2o = kk["so"][op2] if k2 == 1 else img_of(op2 - 4, kk["io"]) if k2 == 2 else 0
    off = kk["boff"][k2] | i1o | i2o
    bit = ((kk["bits"][op2 >> 2] | (i1b * 0x01010101)) >> (8 * (op2 & 3))) & 31
    return off, bit


def valid_pair(m, p1, op1, p2, op2, ham):
    """where the kernel's v1 / v2 rules can be true at all (the duplicate rules only remove more): the first operation leaves p2
    characters to its left, nothing is inserted after the last character, Hamming mode has no indels"""
    ins1 = op1 >= 4
    if ham:
        if op1 >= 4 or op2 >= 4:
            return False
        v1 = (1 <= op1 <= 3) or (op1 == 0 and p1 == 1)
        v2 = (p2 < p1 and op1 != 0) if 1 <= op2 <= 3 else (op2 == 0 and p2 == 1)
        return v1 and v2
    return (p1 > p2 or ins1) and not (p1 == m and ins1)


def queries(rng):
    qs = ["".join(rng.choice("ACGT") for _ in range(rng.choice((20, 20, 21, 24, 27, 30)))) for _ in range(6)]
    qs += ["A" * 20, "ACACACACACACACACACACAC", "GGGGGGGGGGTTTTTTTTTTCC", "ACGT" * 5]
    return qs


def test_place_is_a_permutation_of_the_code_bits():
    rng = random.Random(5)
    for s in FIELDS:
        seen = set()
        for b in range(2 * K2):
            off, bit = filt_pos(1 << b, s)
            assert bin(off).count("1") + bin(bit).count("1") == 1  # one code bit -> one bit of the place
            seen.add((off, bit))
        assert len(seen) == 2 * K2  # 36 code bits -> 36 distinct single-bit places
        for _ in range(2000):
            a = rng.getrandbits(2 * K2)
            b = rng.getrandbits(2 * K2) & ~a
            pa, pb, pab = filt_pos(a, s), filt_pos(b, s), filt_pos(a | b, s)
            assert pab == (pa[0] | pb[0], pa[1] | pb[1])
        assert max(filt_pos(MASK2, s)[0], 0) < 1 << 31  # word offsets inside a copy fit 32 bits at K2 = 18


def test_edit_string_is_apply_edit_without_the_operation_word():
    rng = random.Random(6)
    for q in queries(rng):
        pk, m = pack(q), len(q)
        for ham in (False, True):
            for pos in range(1, m + 1):
                for op in range(4 if ham else 8):
                    assert edit_string(pk, m, pos, op, ham) == apply_edit(pk, m, pos, op, ham), (q, pos, op, ham)


@pytest.mark.parametrize("ham", [False, True])
def test_or_of_parts_is_the_place_of_the_built_string(ham):
    rng = random.Random(7 + ham)
    checked = 0
    for q in queries(rng):
        pk, m = pack(q), len(q)
        if m < K2 + (0 if ham else 2):  # LONG2 takes queries whose shortest string still asks the long filter (Batch::fast2_minlen)
            continue
        for p2 in range(1, m):
            for p1 in range(p2, m + 1):
                for op1 in range(8):
                    for op2 in range(8):
                        if not valid_pair(m, p1, op1, p2, op2, ham):
                            continue
                        s1, l1 = apply_edit(pk, m, p1, op1, ham)
                        s2, l2 = apply_edit(s1, l1, p2, op2, ham)
                        assert l2 >= K2
                        for s in FIELDS:
                            assert or_place(pk, m, p1, op1, p2, op2, s, ham) == direct_place(s2, s), (q, p1, op1, p2, op2, s, ham)
                            checked += 1
    assert checked > (50000 if ham else 500000)
