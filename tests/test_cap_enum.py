"""neighbors() with its size cap as arithmetic over all leaves (dicey_amd/csrc/cap_enum.hpp, r04): leaf ranks in the reference's
depth-first order as closed forms, a string's birth (first leaf that spells it) and death (first leaf that spells a proper
substring), +1 / -1 events over ranks, the first rank where the working set reaches maxsize.  Held here — on the host, with the
header's own functions — against the literal restatement of neighbors.h:29-92 (oracle), cap silent and cap firing."""
import ctypes as C
import os
import random
import subprocess

import pytest

import oracle_lib as O

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def ce():
    subprocess.check_call(["make", "-s", "-C", os.path.join(HERE, "host")])
    L = C.CDLL(os.path.join(HERE, "host", "libcapenum.so"))
    L.ce_enumerate.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64),
                               C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
    L.ce_free.argtypes = [C.c_void_p]
    return L


def run(ce, q, d, maxsize):
    out, n, fired, nl = C.c_void_p(), C.c_uint64(), C.c_int(), C.c_uint64()
    rc = ce.ce_enumerate(q.encode(), len(q), d, maxsize, C.byref(out), C.byref(n), C.byref(fired), C.byref(nl))
    assert rc == 0, rc
    s = C.string_at(out).decode().split("\n")[:-1]
    ce.ce_free(out)
    assert len(s) == n.value
    return s, bool(fired.value)


def queries(rng, lens, n):
    out = []
    for _ in range(n):
        m = rng.choice(lens)
        r = rng.random()
        if r < 0.5:
            q = "".join(rng.choice("ACGT") for _ in range(m))
        elif r < 0.8:  # low complexity: many duplicate leaves, long chains of substrings
            unit = "".join(rng.choice("ACGT") for _ in range(rng.randrange(1, 4)))
            q = (unit * 40)[:m]
        else:
            q = "".join(rng.choice("AC") for _ in range(m))
        out.append(q)
    return out


def test_distance_one_every_cap(ce):
    rng = random.Random(11)
    for q in queries(rng, [10, 11, 13, 20, 27, 30], 60):
        full = O.neighbors(q, 1, True, 1 << 30)
        for cap in [1, 2, 3, 17, len(full) - 1, len(full), len(full) + 1, 10000]:
            if cap < 1:
                continue
            want = O.neighbors(q, 1, True, cap)
            got, fired = run(ce, q, 1, cap)
            assert got == sorted(want) and fired == (len(want) >= cap), (q, cap, len(got), len(want))


def test_distance_two_silent_and_firing(ce):
    rng = random.Random(12)
    for q in queries(rng, [10, 12, 14], 12):
        full = sorted(O.neighbors(q, 2, True, 1 << 30))
        got, fired = run(ce, q, 2, 1 << 30)
        assert got == full and not fired, (q, len(got), len(full))
        for cap in sorted({1, 2, 50, 117, 500, len(full) // 2, len(full) - 1, len(full)}):
            if cap < 1:
                continue
            want = O.neighbors(q, 2, True, cap)
            got, fired = run(ce, q, 2, cap)
            assert got == sorted(want) and fired == (len(want) >= cap), (q, cap, len(got), len(want))


def test_distance_two_primer_lengths_fast_checker(ce):
    """20-29-mers at distance 2 against the checker's hash-set form (tested equal to the literal one in test_oracle.py); the
    cap fires from 21 nt on at the default 10000"""
    rng = random.Random(13)
    for q in queries(rng, [20, 21, 23, 25, 27, 29], 12):
        for cap in (10000, 2000):
            want = O.neighbors_fast(q, 2, True, cap)
            got, fired = run(ce, q, 2, cap)
            assert got == sorted(want) and fired == (len(want) >= cap), (q, cap, len(got), len(want))
