"""Number formatting of `dicey search` output: dicey_amd/cli/dtoa.hpp (own Grisu2) must print every double exactly like
nlohmann::json 3.5.0 does in the reference (src/silica.h:143,149,160-170).  The checker is the reference's own vendored
header compiled in place (oracle/_ref/libjsonref.so)."""
import ctypes as C
import os
import random
import struct
import subprocess

import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mine(tmp_path_factory):
    d = tmp_path_factory.mktemp("dtoa")
    src = d / "w.cpp"
    src.write_text('#include "%s/dicey_amd/cli/dtoa.hpp"\nextern "C" int my_dump(double x, char* o, int cap) { std::string s = '
                   'dtoa::dump_double(x); if ((int)s.size() + 1 > cap) return -1; memcpy(o, s.c_str(), s.size() + 1); return (int)s.size(); }\n' % ROOT)
    so = d / "libdtoa.so"
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(so), str(src)])
    L = C.CDLL(str(so))
    L.my_dump.argtypes = [C.c_double, C.c_char_p, C.c_int]
    return L


def test_known_formats(mine):
    b = C.create_string_buffer(64)
    for x, want in [(58.12604603130177, "58.12604603130177"), (0.0, "0.0"), (-0.0, "-0.0"), (100.0, "100.0"), (1e15, "1e+15"),
                    (0.0001, "0.0001"), (1e-5, "1e-05"), (0.30000000000000004, "0.30000000000000004"), (5e-324, "5e-324"),
                    (1.2345678901234568e17, "1.2345678901234568e+17"), (45.000000000000014, "45.000000000000014")]:
        mine.my_dump(x, b, 64)
        assert b.value.decode() == want  # values as printed by the reference header (recorded from oracle/_ref)


@pytest.mark.skipif(O.ref_libs() is None, reason="oracle/_ref not built (needs /root/reference)")
def test_random_doubles_match_the_reference_header(mine):
    _, J = O.ref_libs()
    J.ref_json_dump_double.argtypes = [C.c_double, C.c_char_p, C.c_int]
    a, b = C.create_string_buffer(64), C.create_string_buffer(64)
    rng = random.Random(11)
    for i in range(60000):
        if i % 3 == 0:
            x = struct.unpack("d", struct.pack("Q", rng.getrandbits(64)))[0]
            if x != x or x in (float("inf"), float("-inf")):
                continue
        elif i % 3 == 1:
            x = rng.uniform(20, 95)
        else:
            x = round(rng.uniform(0, 30), rng.randint(0, 8))
        J.ref_json_dump_double(x, a, 64)
        mine.my_dump(x, b, 64)
        assert a.value == b.value, repr(x)
