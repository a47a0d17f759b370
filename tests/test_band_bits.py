"""The verify stage's bit-plane banded alignment (dicey_amd/csrc/band_bits.hpp, r04) compiled for the host and held against the
checker's needle() (needle.h:59-138, hunter.h:383-401) on random hits: text around a neighbourhood string of the query, context
clipped at the text ends and cut at sequence separators exactly like hunter.h:363-378.  No GPU needed: the same header is what
k_verify_memo runs on the device."""
import ctypes as C
import os
import random
import subprocess

import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))
MISMATCH, REF_GAP, QUERY_GAP, NONE = 0, 1, 2, 0xFFFFFFFF


@pytest.fixture(scope="module")
def bb():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host")])
    L = C.CDLL(os.path.join(HERE, "host", "libbandbits.so"))
    L.bb_align.argtypes = [C.c_char_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint32, C.c_int, C.c_int,
                           C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]
    L.bb_align.restype = C.c_int
    L.bb_align_kw.argtypes = L.bb_align.argtypes + [C.c_int, C.c_uint64, C.c_uint32, C.c_uint32]
    L.bb_align_kw.restype = C.c_int
    return L


def masks_of(q):
    return [sum(1 << i for i, ch in enumerate(q) if ch == x) for x in "ACGT"]


def ops_of_rows(ra, qa):
    ops = []
    for col, (x, y) in enumerate(zip(ra, qa)):
        if x == y and x != "-":
            continue
        kind = QUERY_GAP if y == "-" else (REF_GAP if x == "-" else MISMATCH)
        ops.append(col | (kind << 16) | ((0 if kind == REF_GAP else ord(x)) << 24))
    return ops


def run(bb, text, loc, mlen, q, d, indel, wide):
    buf = text.encode("latin-1") + b"\0" * 16  # text[n-1] = 0 like the index's copy, and room for the aligned word loads
    info, pre_eff = C.c_uint32(), C.c_uint32()
    ops = (C.c_uint32 * 2)()
    m = (C.c_uint32 * 4)(*masks_of(q))
    fault = bb.bb_align(buf, len(text) + 1, loc, mlen, len(q), d, int(indel), int(wide), m, C.byref(info), ops, C.byref(pre_eff))
    return fault, info.value, [x for x in ops if x != NONE], pre_eff.value


def expected(text, loc, mlen, q, d):
    """hunter.h:363-401 on the text: window, '\\n' cut, needle, column stripping"""
    n_text = len(text) + 1
    pre, post = min(d, loc), min(d, n_text - loc - mlen)
    pre_eff = 0
    for i in range(1, pre + 1):
        if text[loc - i] == "\n":
            break
        pre_eff = i
    post_eff = 0
    for i in range(post):
        if loc + mlen + i >= len(text) or text[loc + mlen + i] == "\n":
            break
        post_eff = i + 1
    window = text[loc - pre_eff: loc + mlen + post_eff]
    score, ra, qa, lead = O.needle_hunt(window, q)
    return score, ra, qa, lead, pre_eff


def random_case(rng, d):
    n = rng.randrange(10, 33)
    q = "".join(rng.choice("ACGTN" if rng.random() < 0.1 else "ACGT") for _ in range(n))
    if rng.random() < 0.3:  # low complexity: ties between gap placements
        unit = "".join(rng.choice("ACGT") for _ in range(rng.randrange(1, 4)))
        q = (unit * 40)[:n]
    s = list(q)
    for _ in range(rng.randrange(0, d + 1)):
        k = rng.randrange(len(s))
        r = rng.random()
        if r < 0.4:
            s[k] = rng.choice("ACGT")
        elif r < 0.7:
            del s[k]
        elif k > 0 or True:
            s.insert(k, rng.choice("ACGT"))
    s = "".join(s)
    alpha = "ACGTN" if rng.random() < 0.8 else "ACGTNRY\n"
    if rng.random() < 0.3:  # context that continues the string's own characters: alignments can slide
        left = (q * 3)[-rng.randrange(0, 6):] if rng.random() < 0.5 else s[:1] * rng.randrange(0, 5)
        right = s[-1:] * rng.randrange(0, 5)
    else:
        left = "".join(rng.choice(alpha) for _ in range(rng.randrange(0, 12)))
        right = "".join(rng.choice(alpha) for _ in range(rng.randrange(0, 12)))
    return q, left + s + right + "\n", len(left), len(s)  # the indexed text ends with a separator (SEQk '\n'), then the sentinel


@pytest.mark.parametrize("d,wide", [(0, 0), (1, 0), (1, 1), (2, 1)])
def test_bit_plane_alignment_equals_needle(bb, d, wide):
    rng = random.Random(1000 + 10 * d + wide)
    checked = 0
    for _ in range(6000):
        q, text, loc, mlen = random_case(rng, d)
        score, ra, qa, lead, pre_eff = expected(text, loc, mlen, q, d)
        if -score > d:
            continue  # not a hit of the <= d neighbourhood (an edit next to the context made it worse): the kernel never sees it
        fault, info, ops, pe = run(bb, text, loc, mlen, q, d, True, wide)
        want_info = (score & 255) | (lead << 8) | (len(ra) << 16)
        assert fault == 0 and pe == pre_eff and info == want_info and ops == ops_of_rows(ra, qa), \
            (q, text, loc, mlen, d, (score, ra, qa, lead), (fault, hex(info), [hex(x) for x in ops], pe))
        checked += 1
    assert checked > 4000


def test_hamming_mode_counts_mismatches(bb):
    rng = random.Random(5)
    for _ in range(2000):
        n = rng.randrange(10, 33)
        q = "".join(rng.choice("ACGT") for _ in range(n))
        s = list(q)
        for k in rng.sample(range(n), rng.randrange(0, 3)):
            s[k] = rng.choice("ACGTN")
        s = "".join(s)
        left = "".join(rng.choice("ACGT\n") for _ in range(rng.randrange(0, 5)))
        text = left + s + "".join(rng.choice("ACGT") for _ in range(rng.randrange(0, 5))) + "\n"
        fault, info, ops, _ = run(bb, text, len(left), n, q, 0, False, rng.randrange(2))  # the kernel passes d = 0 in Hamming mode: no context
        mm = [(i, s[i]) for i in range(n) if s[i] != q[i]]
        assert fault == 0 and info == (((-len(mm)) & 255) | (n << 16))
        assert ops == [i | (MISMATCH << 16) | (ord(c) << 24) for i, c in mm][:2]


@pytest.mark.parametrize("d,wide", [(0, 0), (1, 0), (1, 1)])
def test_window_from_the_hits_own_codes_equals_the_window_from_the_text(bb, d, wide):
    """r06: a hit whose string and neighbouring characters are all A/C/G/T can bring them along as 2-bit codes (Sel::key, the seed's
    context bits); band_window_from_key spreads them into the bytes band_align_bits would have read from the text.  Same result as
    the text path — the text handed to the key path is garbage, so a stray read of it would show."""
    rng = random.Random(77 + d + wide)
    code = {"A": 0, "C": 1, "G": 2, "T": 3}
    checked = 0
    for _ in range(4000):
        n = rng.randrange(12, 23)
        q = "".join(rng.choice("ACGT") for _ in range(n))
        s = list(q)
        for _e in range(rng.randrange(0, d + 1)):
            k = rng.randrange(len(s))
            r = rng.random()
            if r < 0.4:
                s[k] = rng.choice("ACGT")
            elif r < 0.7 and len(s) > 11:
                del s[k]
            else:
                s.insert(k, rng.choice("ACGT"))
        s = "".join(s)
        if len(s) + 2 > 24:
            continue
        left = (s[:1] * 3 if rng.random() < 0.3 else "".join(rng.choice("ACGT") for _ in range(3)))
        right = (s[-1:] * 3 if rng.random() < 0.3 else "".join(rng.choice("ACGT") for _ in range(3)))
        text = left + s + right + "\n"
        loc, mlen = len(left), len(s)
        score, ra, qa, lead, pre_eff = expected(text, loc, mlen, q, d)
        if -score > d:
            continue
        want = run(bb, text, loc, mlen, q, d, True, wide)
        key = 0
        for ch in s:
            key = (key << 2) | code[ch]
        buf = b"\n" * (len(text) + 17)
        info, pe = C.c_uint32(), C.c_uint32()
        ops = (C.c_uint32 * 2)()
        m = (C.c_uint32 * 4)(*masks_of(q))
        fault = bb.bb_align_kw(buf, len(text) + 1, loc, mlen, len(q), d, 1, int(wide), m, C.byref(info), ops, C.byref(pe), 1, key,
                               code[text[loc - 1]], code[text[loc + mlen]])
        got = (fault, info.value, [x for x in ops if x != NONE], pe.value)
        assert got == want and fault == 0, (q, text, loc, mlen, d, want, got)
        checked += 1
    assert checked > 2500
