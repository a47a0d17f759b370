"""ctypes binding of the ORACLE (oracle/liboracle.so) — the checker, never the product.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os
import subprocess

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
_SO = os.path.join(_ODIR, "liboracle.so")


class HuntParams(C.Structure):
    _fields_ = [("distance", C.c_uint32), ("hamming", C.c_int32), ("forward_only", C.c_int32),
                ("max_locations", C.c_uint64), ("max_neighborhood", C.c_uint32)]


def build():
    srcs = [os.path.join(_ODIR, f) for f in ("oracle_capi.cpp", "fm9.hpp", "hunt_ref.hpp", "Makefile")]
    if (not os.path.exists(_SO)) or any(os.path.getmtime(s) > os.path.getmtime(_SO) for s in srcs):
        subprocess.check_call(["make", "-C", _ODIR, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_last_error.restype = C.c_char_p
        L.orc_free.argtypes = [C.c_void_p]
        L.orc_build_fm9.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p]
        L.orc_open.argtypes = [C.c_char_p]
        L.orc_open.restype = C.c_void_p
        L.orc_close.argtypes = [C.c_void_p]
        L.orc_size.argtypes = [C.c_void_p]
        L.orc_size.restype = C.c_uint64
        L.orc_count.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
        L.orc_count.restype = C.c_uint64
        L.orc_locate.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64]
        L.orc_locate.restype = C.c_uint64
        L.orc_extract.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_char_p]
        L.orc_sa.argtypes = [C.c_void_p, C.c_uint64]
        L.orc_sa.restype = C.c_uint64
        L.orc_code_len.argtypes = [C.c_void_p, C.c_int]
        L.orc_code_len.restype = C.c_uint32
        L.orc_bf_locate.argtypes = [C.c_char_p, C.c_uint64, C.c_char_p, C.c_uint64, C.POINTER(C.c_uint64), C.c_uint64]
        L.orc_bf_locate.restype = C.c_uint64
        L.orc_neighbors.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.POINTER(C.c_uint64)]
        L.orc_neighbors.restype = C.c_void_p
        L.orc_needle.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_uint32)]
        L.orc_needle.restype = C.c_int
        L.orc_hunt.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.c_uint32, C.POINTER(HuntParams),
                               C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p), C.c_uint64,
                               C.POINTER(C.c_uint64), C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
        L.orc_hunt.restype = C.c_void_p
        L.orc_hunt_timed.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.c_uint32, C.POINTER(HuntParams), C.POINTER(C.c_char_p),
                                     C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
        L.orc_hunt_timed.restype = C.c_double
        _lib = L
    return _lib


def _take(ptr, n=None):
    s = C.string_at(ptr) if n is None else C.string_at(ptr, n)
    lib().orc_free(ptr)
    return s


def build_fm9(text: bytes, path: str):
    if lib().orc_build_fm9(text, len(text), path.encode()) != 0:
        raise RuntimeError(lib().orc_last_error().decode())


def neighbors(query: str, dist: int, indel: bool, maxsize: int = 10000):
    cnt = C.c_uint64()
    p = lib().orc_neighbors(query.encode(), dist, int(indel), maxsize, C.byref(cnt))
    s = _take(p).decode()
    out = s.split("\n")[:-1] if s else []
    assert len(out) == cnt.value
    return out


def needle(a1: str, a2: str):
    r0, r1, tg = C.c_void_p(), C.c_void_p(), C.c_uint32()
    sc = lib().orc_needle(a1.encode(), a2.encode(), C.byref(r0), C.byref(r1), C.byref(tg))
    return sc, _take(r0).decode(), _take(r1).decode(), tg.value


def bf_locate(text: bytes, pat: bytes):
    n = lib().orc_bf_locate(text, len(text), pat, len(pat), None, 0)
    buf = (C.c_uint64 * max(1, n))()
    lib().orc_bf_locate(text, len(text), pat, len(pat), buf, n)
    return list(buf[:n])


class Index:
    def __init__(self, path):
        self.h = lib().orc_open(path.encode())
        if not self.h:
            raise RuntimeError(lib().orc_last_error().decode())

    def close(self):
        if self.h:
            lib().orc_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    @property
    def size(self):
        return lib().orc_size(self.h)

    def count(self, pat: bytes):
        return lib().orc_count(self.h, pat, len(pat))

    def locate(self, pat: bytes):
        n = self.count(pat)
        buf = (C.c_uint64 * max(1, n))()
        k = lib().orc_locate(self.h, pat, len(pat), buf, n)
        assert k == n
        return list(buf[:n])

    def extract(self, b, e):
        buf = C.create_string_buffer(e - b + 1)
        lib().orc_extract(self.h, b, e, buf)
        return buf.raw

    def sa(self, i):
        return lib().orc_sa(self.h, i)

    def code_len(self, ch):
        return lib().orc_code_len(self.h, ord(ch) if isinstance(ch, str) else ch)

    def hunt(self, seqlen, seqname, seqs, qnames=None, distance=1, hamming=False, forward_only=False,
             max_locations=1000, max_neighborhood=10000, genome="", outfile="", want_hits=False):
        """Returns (json_text, pushed_hits or None); pushed_hits = list of tuples in reference push order."""
        nseq, nq = len(seqlen), len(seqs)
        sl = (C.c_uint32 * nseq)(*seqlen)
        sn = (C.c_char_p * nseq)(*[s.encode() for s in seqname])
        p = HuntParams(distance, int(hamming), int(forward_only), max_locations, max_neighborhood)
        qs = (C.c_char_p * nq)(*[s.encode() for s in seqs])
        qn = (C.c_char_p * nq)(*[(s or "").encode() for s in (qnames or [""] * nq)])
        jl, hl, hb = C.c_uint64(), C.c_uint64(), C.c_void_p()
        jp = lib().orc_hunt(self.h, sl, sn, nseq, C.byref(p), genome.encode(), outfile.encode(), qn, qs, nq,
                            C.byref(jl), C.byref(hb) if want_hits else None, C.byref(hl))
        js = _take(jp, jl.value).decode()
        hits = None
        if want_hits:
            blob = _take(hb, hl.value).decode()
            hits = []
            for ln in blob.split("\n")[:-1]:
                f = ln.split("\t")
                hits.append((int(f[0]), int(f[1]), int(f[2]), int(f[3]), f[4], f[5], f[6]))
        return js, hits

    def hunt_timed(self, seqlen, seqs, threads=1, distance=1, hamming=False, forward_only=False,
                   max_locations=1000, max_neighborhood=10000):
        nseq, nq = len(seqlen), len(seqs)
        sl = (C.c_uint32 * nseq)(*seqlen)
        p = HuntParams(distance, int(hamming), int(forward_only), max_locations, max_neighborhood)
        qs = (C.c_char_p * nq)(*[s.encode() for s in seqs])
        ctr = (C.c_uint64 * 5)()
        th = C.c_uint64()
        dt = lib().orc_hunt_timed(self.h, sl, nseq, C.byref(p), qs, nq, threads, ctr, C.byref(th))
        return dt, dict(zip(["patterns", "bs_steps", "located", "extracted", "needles"], list(ctr))), th.value
