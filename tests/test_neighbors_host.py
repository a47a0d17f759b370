"""dg_neighbors (host side of libdiceygpu.so, no device needed): neighbors() of the reference WITH its size cap
(src/neighbors.h:29-92), against the oracle's literal restatement and the survey's known answers."""
import ctypes as C
import random

import pytest

import oracle_lib as O


@pytest.fixture(scope="module")
def L():
    from dicey_amd import _capi
    return _capi.load()


def dg_neighbors(L, q, d, indel, cap):
    out, cnt, fired = C.c_void_p(), C.c_uint64(), C.c_int()
    rc = L.dg_neighbors(q.encode(), len(q), d, 0 if indel else 1, cap, C.byref(out), C.byref(cnt), C.byref(fired))
    assert rc == 0, L.dg_last_error()
    s = C.string_at(out.value).decode().split("\n")[:-1]
    L.dg_buffer_free(out)
    assert len(s) == cnt.value
    return s, bool(fired.value)


def test_known_answers_from_survey(L):
    q = "TCTCTGCACACACGTTGTAC"  # SURVEY.md §8(c), measured on the unmodified reference headers
    assert [len(dg_neighbors(L, q, d, False, 10000)[0]) for d in (0, 1, 2)] == [1, 61, 1771]
    assert [len(dg_neighbors(L, q, d, True, 10000)[0]) for d in (0, 1, 2)] == [1, 116, 6019]


def test_equals_oracle_with_and_without_cap(L):
    rng = random.Random(1)
    cases = []
    for m in (10, 12, 15, 20):
        for d in (0, 1, 2):
            for indel in (True, False):
                for cap in (10000, 50, 7, 1, 0):
                    cases.append(("".join(rng.choice("ACGT") for _ in range(m)), d, indel, cap))
    cases += [("A" * 12, 2, True, 10000), ("ACACACACACAC", 2, True, 100), ("ACGTNNACGTAC", 2, True, 10000),
              ("ACGTNNACGTAC", 2, False, 300), ("ACGTACGTACGTA", 3, False, 10000), ("ACGTACGTACGT", 3, True, 2000),
              ("N" * 14, 2, True, 10000), ("ACGTACGTACGTACGTACGTACG", 1, True, 100)]
    for q, d, indel, cap in cases:
        got, fired = dg_neighbors(L, q, d, indel, cap)
        want = O.neighbors(q, d, indel, cap)
        assert got == list(want), (q, d, indel, cap, len(got), len(want))
        assert fired == (len(want) >= cap)


def test_cap_fires_on_25mers_at_edit_distance_2(L):
    """25-mers at d=2 run into the default cap; 21-24-mers mostly stay under it"""
    rng = random.Random(3)
    seen = set()
    for m in (22, 25, 25, 27):
        q = "".join(rng.choice("ACGT") for _ in range(m))
        got, fired = dg_neighbors(L, q, 2, True, 10000)
        want = O.neighbors(q, 2, True, 10000)
        assert got == list(want)
        assert fired == (len(want) >= 10000)
        seen.add(fired)
    assert seen == {True, False}
