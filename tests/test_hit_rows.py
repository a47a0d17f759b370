"""dg_hit_rows (include/dicey_gpu.h "compact alignment"): the two rows of a hit rebuilt from dg_hit + its operation words.  Host
code only — runs without a GPU.  Expected rows come from the checker's needle() (hunter.h:383-401) on random windows."""
import ctypes as C
import random

import pytest

import oracle_lib as O
from dicey_amd import _capi

MISMATCH, REF_GAP, QUERY_GAP, NONE = 0, 1, 2, 0xFFFFFFFF


def ops_of_rows(ra, qa):
    """the description a device kernel would emit for these kept rows"""
    ops = []
    for col, (x, y) in enumerate(zip(ra, qa)):
        if x == y and x != "-":
            continue
        kind = QUERY_GAP if y == "-" else (REF_GAP if x == "-" else MISMATCH)
        ops.append(col | (kind << 16) | ((0 if kind == REF_GAP else ord(x)) << 24))
    return ops


def revcomp(s):
    return s[::-1].translate(str.maketrans("ACGTN", "TGCAN"))


def rows_from_library(L, score, strand, aln_len, ops, per_hit, query_fw):
    hit = _capi.Hit(score, 0, 1, 0, aln_len, ord(strand), 0)
    arr = (C.c_uint32 * max(1, per_hit))(*(ops + [NONE] * (per_hit - len(ops))))
    ra = C.create_string_buffer(aln_len + 1)
    qa = C.create_string_buffer(aln_len + 1)
    rc = L.dg_hit_rows(C.byref(hit), arr if per_hit else None, per_hit, query_fw.encode(), len(query_fw), ra, qa)
    return rc, ra.raw[:aln_len].decode("latin-1"), qa.raw[:aln_len].decode("latin-1")


@pytest.fixture(scope="module")
def lib():
    return _capi.load()


def test_rows_equal_the_checkers_alignment_on_random_windows(lib):
    rng = random.Random(77)
    n_checked = 0
    for it in range(3000):
        n = rng.randrange(10, 33)
        d = rng.randrange(0, 3)
        q = "".join(rng.choice("ACGTN" if rng.random() < 0.1 else "ACGT") for _ in range(n))
        s = list(q)
        for _ in range(rng.randrange(0, d + 1)):  # a neighbourhood string: up to d edits of the query
            k = rng.randrange(len(s))
            r = rng.random()
            if r < 0.4:
                s[k] = rng.choice("ACGT")
            elif r < 0.7 and len(s) > 8:
                del s[k]
            else:
                s.insert(k, rng.choice("ACGT"))
        window = "".join(rng.choice("ACGTNR") for _ in range(rng.randrange(0, d + 1))) + "".join(s) + \
            "".join(rng.choice("ACGTNY") for _ in range(rng.randrange(0, d + 1)))
        score, ra, qa, lead = O.needle_hunt(window, q)  # kept rows after lead / trail stripping
        if -score > 2:
            continue
        ops = ops_of_rows(ra, qa)
        assert len(ops) == -score, (window, q, ra, qa)
        strand = rng.choice("+-")
        fw = q if strand == "+" else revcomp(q)  # the library receives the forward strand and forms the other one itself
        for per_hit in {max(len(ops), d), 2, 4}:
            if per_hit < len(ops):
                continue
            rc, ra2, qa2 = rows_from_library(lib, score, strand, len(ra), ops, per_hit, fw)
            assert rc == 0 and (ra2, qa2) == (ra, qa), (window, q, strand, ops, ra, qa, ra2, qa2)
        n_checked += 1
    assert n_checked > 2000


def test_corrupted_descriptions_are_refused(lib):
    q = "ACGTACGTACGTACGTACGT"
    assert rows_from_library(lib, 0, "+", 20, [], 1, q)[0] == 0
    assert rows_from_library(lib, 0, "+", 21, [], 1, q)[0] != 0          # a column too many for the query
    assert rows_from_library(lib, -1, "+", 20, [5 | (QUERY_GAP << 16) | (65 << 24)], 1, q)[0] != 0   # 19 query characters used
    assert rows_from_library(lib, -1, "+", 20, [25 | (MISMATCH << 16) | (65 << 24)], 1, q)[0] != 0   # operation behind the row
    rc, ra, qa = rows_from_library(lib, -1, "-", 21, [3 | (QUERY_GAP << 16) | (ord("R") << 24)], 2, q)
    assert rc == 0 and qa == revcomp(q)[:3] + "-" + revcomp(q)[3:] and ra == revcomp(q)[:3] + "R" + revcomp(q)[3:]
