"""primer3 thal() (`dicey search`, src/silica.h:437,511).  tests/golden/thal_vectors.json comes from the REFERENCE
ITSELF (oracle/_ref = unmodified src/thal.h compiled in place); the HIP kernel must reproduce every double bit for bit."""
import ctypes as C
import json
import os
import struct

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libthalref.so")


def _vectors(name="thal_vectors.json"):
    return json.load(open(os.path.join(GOLD, name)))


SETS = ["thal_vectors.json", "thal_vectors_long.json"]  # 10-35 nt at primer3 defaults; 30-61 nt at other salt / DNA settings


def test_golden_holds_the_survey_known_answer():
    v = _vectors()["vectors"][0]
    assert v[0] == "GCCCCATAGGTTTTGAACTCA"
    t = struct.unpack(">d", bytes.fromhex(v[2]))[0]
    assert repr(t) == "58.12604603130177" and v[3:5] == [21, 21]  # 58.126046031301769, 21/21


@pytest.mark.skipif(not os.path.exists(REF_SO), reason="oracle/_ref is built only where /root/reference exists")
@pytest.mark.parametrize("name", SETS)
def test_reference_build_reproduces_the_golden_vectors(name):
    R = C.CDLL(REF_SO)
    R.ref_thal_init.argtypes = [C.c_char_p] + [C.c_double] * 5
    R.ref_thal.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int), C.POINTER(C.c_int)]
    g = _vectors(name)
    p = g["params"]
    assert R.ref_thal_init(O.PRIMER3_CONFIG.encode(), p["temp_c"], p["mv"], p["dv"], p["dna_conc"], p["dntp"]) == 0
    t, a, b = C.c_double(), C.c_int(), C.c_int()
    for o1, o2, hx, e1, e2, ok in g["vectors"][:300] + g["vectors"][-20:]:
        assert R.ref_thal(o1.encode(), o2.encode(), C.byref(t), C.byref(a), C.byref(b)) == ok
        assert struct.pack(">d", t.value).hex() == hx and (not ok or (a.value, b.value) == (e1, e2))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SETS)
def test_thal_kernel_is_bit_identical_to_the_reference(name):
    import dicey_amd
    g = _vectors(name)
    p = g["params"]
    th = dicey_amd.Thal(O.PRIMER3_CONFIG, mv=p["mv"], dv=p["dv"], dntp=p["dntp"], dna_conc=p["dna_conc"])
    got = th.tm([(v[0], v[1]) for v in g["vectors"]])
    bad = [(v, r) for v, r in zip(g["vectors"], got)
           if struct.pack(">d", r[0]).hex() != v[2] or [r[1], r[2]] != v[3:5]]
    assert not bad, bad[:3]
    th.close()
