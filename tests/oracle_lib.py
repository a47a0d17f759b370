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


class SearchParams(C.Structure):
    _fields_ = [("hamming", C.c_int32), ("pruneprimer", C.c_int32), ("cutTemp", C.c_double), ("maxProdSize", C.c_uint32),
                ("cutofPen", C.c_double), ("penDiff", C.c_double), ("penMis", C.c_double), ("penLen", C.c_double),
                ("kmer", C.c_uint32), ("distance", C.c_uint32), ("maxNeighborhood", C.c_uint32), ("maxPruneCount", C.c_uint32),
                ("max_locations", C.c_uint64)]


REF_THAL = os.path.join(_ODIR, "_ref", "libthalref.so")
REF_JSON = os.path.join(_ODIR, "_ref", "libjsonref.so")
sys_path_added = os.path.join(_ROOT, "tests", "golden")
import sys as _sys
if sys_path_added not in _sys.path:
    _sys.path.insert(0, sys_path_added)
import p3config as _p3
PRIMER3_CONFIG = _p3.config_dir()
_ref = {}


def ref_libs(temp_c=37.0, mv=50.0, dv=1.5, dna_conc=50.0, dntp=0.6):
    """oracle/_ref: the reference's own thal.h and nlohmann json.hpp, compiled in place where /root/reference exists
    (the built .so files travel to the GPU box).  Returns (thal_lib, json_lib) or None."""
    if not (os.path.exists(REF_THAL) and os.path.exists(REF_JSON)):
        return None
    if "t" not in _ref:
        T = C.CDLL(REF_THAL)
        T.ref_thal_init.argtypes = [C.c_char_p] + [C.c_double] * 5
        J = C.CDLL(REF_JSON)
        _ref["t"], _ref["j"] = T, J
    assert _ref["t"].ref_thal_init(PRIMER3_CONFIG.encode(), temp_c, mv, dv, dna_conc, dntp) == 0
    return _ref["t"], _ref["j"]


def build():
    srcs = [os.path.join(_ODIR, f) for f in ("oracle_capi.cpp", "fm9.hpp", "hunt_ref.hpp", "search_ref.hpp", "padlock_ref.hpp", "Makefile")]
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
        L.orc_search.argtypes = [C.c_void_p, C.POINTER(C.c_uint32), C.POINTER(C.c_char_p), C.c_uint32, C.c_char_p, C.c_uint64,
                                 C.POINTER(SearchParams), C.c_char_p, C.c_char_p, C.c_char_p, C.c_void_p, C.c_void_p,
                                 C.POINTER(C.c_int), C.POINTER(C.c_uint64)]
        L.orc_search.restype = C.c_void_p
        L.orc_use_ref_json.argtypes = [C.c_char_p]
        _lib = L
        # every JSON object of the writers through the reference's own nlohmann::json when oracle/_ref has it
        if os.path.exists(REF_JSON) and os.environ.get("ORC_JSON", "ref") != "own":
            L.orc_use_ref_json(REF_JSON.encode())
    return _lib


def use_ref_json(on: bool) -> bool:
    """Switch the oracle's JSON writers between the reference's nlohmann (oracle/_ref/libjsonref.so) and the restated one."""
    return bool(lib().orc_use_ref_json(REF_JSON.encode() if on else None))


def ref_json_in_use() -> bool:
    return bool(lib().orc_ref_json_in_use())


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


def neighbors_fast(query: str, dist: int, indel: bool, maxsize: int = 10000):
    """the hash-set form used for large distance-2 parity runs (oracle/hunt_ref.hpp neighbors_fast)"""
    cnt = C.c_uint64()
    L = lib()
    L.orc_neighbors2.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_uint32, C.c_int, C.POINTER(C.c_uint64)]
    L.orc_neighbors2.restype = C.c_void_p
    s = _take(L.orc_neighbors2(query.encode(), dist, int(indel), maxsize, 1, C.byref(cnt))).decode()
    out = s.split("\n")[:-1] if s else []
    assert len(out) == cnt.value
    return out


def fast_neighbors(on: bool):
    """hunt_one / hunt_timed enumerate neighbourhoods with neighbors_fast (only for order-independent cases)"""
    lib().orc_fast_neighbors(int(on))


def needle(a1: str, a2: str):
    r0, r1, tg = C.c_void_p(), C.c_void_p(), C.c_uint32()
    sc = lib().orc_needle(a1.encode(), a2.encode(), C.byref(r0), C.byref(r1), C.byref(tg))
    return sc, _take(r0).decode(), _take(r1).decode(), tg.value


def needle_hunt(window: str, query: str):
    """needle() followed by hunt's column stripping (hunter.h:391-401, _trailGap :69-77): (score, refalign, queryalign, leading
    query-gap columns dropped)"""
    sc, r0, r1, tg = needle(window, query)
    stop = len(r1) - tg
    ra, qa, lead, in_lead = [], [], 0, True
    for j in range(stop):
        if r1[j] != "-":
            in_lead = False
        if in_lead:
            lead += 1
        else:
            ra.append(r0[j])
            qa.append(r1[j])
    return sc, "".join(ra), "".join(qa), lead


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
        try:
            self.close()
        except Exception:  # interpreter shutdown
            pass

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

    def hunt_parallel(self, seqlen, seqname, seqs, workers=32, **kw):
        """hunt() over contiguous slices of the queries on host threads (ctypes releases the GIL, the handle is read-only): the
        checker's loop is single-threaded like the reference's and a cap-firing 25-mer at distance 2 costs it seconds.  Returns
        (json lines per query, hits per query index in push order)."""
        from concurrent.futures import ThreadPoolExecutor
        n = len(seqs)
        workers = max(1, min(workers, n))
        per = (n + workers - 1) // workers
        parts = [(i, seqs[i:i + per]) for i in range(0, n, per)]

        def run(part):
            base, sub = part
            js, hits = self.hunt(seqlen, seqname, sub, want_hits=True, **kw)
            d = {}
            for h in hits:
                d.setdefault(base + h[0], []).append(h[1:])
            return base, js.split("\n")[:-1], d
        lines, allhits = [None] * n, {}
        with ThreadPoolExecutor(max_workers=workers) as ex:
            for base, ls, d in ex.map(run, parts):
                lines[base:base + len(ls)] = ls
                allhits.update(d)
        return lines, allhits

    def search(self, seqlen, seqname, text: bytes, fasta: str, genome="", outfile="", hamming=False, pruneprimer=None,
               cutTemp=45.0, maxProdSize=15000, cutofPen=-1.0, penDiff=0.6, penMis=0.4, penLen=0.001, kmer=15, distance=1,
               maxNeighborhood=10000, max_locations=10000):
        """`dicey search` through the restated silica.h driver + the reference's own thal()/json dump (oracle/_ref).
        Returns (json_text, exit_code)."""
        libs = ref_libs()
        if libs is None:
            raise RuntimeError("oracle/_ref is not built")
        T, J = libs
        nseq = len(seqlen)
        sl = (C.c_uint32 * nseq)(*seqlen)
        sn = (C.c_char_p * nseq)(*[s.encode() for s in seqname])
        p = SearchParams(int(hamming), int(pruneprimer is not None), cutTemp, maxProdSize, cutofPen, penDiff, penMis, penLen, kmer,
                         distance, maxNeighborhood, pruneprimer or 0, max_locations)
        rc, jl = C.c_int(), C.c_uint64()
        tf = C.cast(T.ref_thal, C.c_void_p)
        jf = C.cast(J.ref_json_dump_double, C.c_void_p)
        jp = lib().orc_search(self.h, sl, sn, nseq, text, len(text), C.byref(p), genome.encode(), outfile.encode(), fasta.encode(),
                              tf, jf, C.byref(rc), C.byref(jl))
        return _take(jp, jl.value).decode(), rc.value

    def padlock(self, chrname, chrseq, gtf_text: str, barcodes_text: str, genes=(), compute_all=False, input_fasta=False, absent=False,
                json=False, hamming=False, probe_mode=False, overlapping=False, distance=1, armlen=20, tmdiff=2, gcmin=0.4, gcmax=0.6,
                ucsc="Unknown", anchor="TGCGTCTATTTAGTGGAGCC", spacerleft="TCCTC", spacerright="TCTTT", feature="exon", idname="gene_id",
                genome="", infile="", outfile="out.tsv", barcodes="", gtf="", jsonfile="", mv=50.0, dv=1.5, dna_conc=50.0, dntp=0.6):
        """`dicey padlock` through the restated padlock.h/gtf.h driver + the reference's own thal() (oracle/_ref).
        chrseq: the FASTA sequences as stored.  Returns (tsv, json_text, stderr_text, exit_code)."""
        libs = ref_libs(37.0, mv, dv, dna_conc, dntp)
        if libs is None:
            raise RuntimeError("oracle/_ref is not built")
        T, _ = libs

        class P(C.Structure):
            _fields_ = [(n, C.c_int32) for n in ("json", "hamming", "probe_mode", "overlapping", "compute_all", "input_fasta", "absent")] + \
                       [(n, C.c_uint32) for n in ("distance", "armlen", "tmdiff")] + [("gcmin", C.c_double), ("gcmax", C.c_double)]
        p = P(int(json), int(hamming), int(probe_mode), int(overlapping), int(compute_all), int(input_fasta), int(absent), distance, armlen,
              tmdiff, gcmin, gcmax)
        strs = [ucsc, anchor, spacerleft, spacerright, feature, idname, genome, infile, outfile, barcodes, gtf, jsonfile]
        sa = (C.c_char_p * len(strs))(*[x.encode() for x in strs])
        ga = (C.c_char_p * max(1, len(genes)))(*[g.encode() for g in genes])
        n = len(chrname)
        cn = (C.c_char_p * n)(*[x.encode() for x in chrname])
        cs = (C.c_char_p * n)(*[x.encode() if isinstance(x, str) else x for x in chrseq])
        rc, jo, eo = C.c_int(), C.c_void_p(), C.c_void_p()
        L = lib()
        L.orc_padlock.restype = C.c_void_p
        L.orc_padlock.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_char_p,
                                  C.c_char_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        tp = L.orc_padlock(self.h, C.byref(p), sa, ga, len(genes), cn, cs, n, gtf_text.encode(), barcodes_text.encode(),
                           C.cast(T.ref_thal, C.c_void_p), C.byref(rc), C.byref(jo), C.byref(eo))
        out = []
        for ptr in (tp, jo.value, eo.value):
            out.append(C.string_at(ptr).decode())
            L.orc_free(C.c_void_p(ptr))
        return out[0], out[1], out[2], rc.value

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
