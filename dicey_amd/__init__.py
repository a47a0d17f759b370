"""dicey_amd — MI355X-native in-silico-PCR search path (drop-in for the `dicey hunt` hot path).

Thin Python mirror of the C ABI in include/dicey_gpu.h.  Names follow the reference's seam
(SURVEY.md §8(b)): an `FmIndex` is what `load_from_checked_file(csa_wt<>)` gives the reference
(src/hunter.h:253-256); `count` / `locate` / `extract` are sdsl's free functions the reference calls
(src/hunter.h:353,355,371); `hunt` is the per-query loop of src/hunter.h:291-437 for a whole batch.
All compute happens in hand-written HIP kernels inside libdiceygpu.so — there is no fallback path.
"""
import ctypes as C
import os
from dataclasses import dataclass, field
from typing import List, Optional, Sequence

from . import _capi
from ._capi import DgError, DG_Q_DIST_ADJUSTED, DG_Q_MAX_MATCHES, DG_Q_NBHD_EXCEEDED, DG_Q_TOO_SHORT

__all__ = ["FmIndex", "Thal", "search_sites", "DnaHit", "QueryResult", "HuntBatch", "build_index", "DgError"]


@dataclass
class DnaHit:  # src/hunter.h:53-66
    score: int
    chr: int
    start: int
    strand: str
    refalign: str
    queryalign: str


@dataclass
class QueryResult:
    sequence: str  # upper-cased, non-ACGT replaced by N (src/hunter.h:306-307)
    distance: int  # after the clamp of src/hunter.h:312-315
    flags: int
    nondna: int
    hits: List[DnaHit] = field(default_factory=list)  # reference PUSH order (before std::sort, src/hunter.h:440)

    def messages(self, max_locations: int, max_neighborhood: int) -> List[str]:
        """The msg vector of src/hunter.h:296-437 in the order the reference pushes it."""
        if self.flags & DG_Q_TOO_SHORT:
            return ["Error: Input sequence is shorter than 10 nucleotides!"]
        m = ["Warning: Non-DNA character in nucleotide sequence detected and replaced by 'N'!"] * self.nondna
        if self.flags & DG_Q_DIST_ADJUSTED:
            m.append("Warning: Distance was adjusted to sequence length!")
        if self.flags & DG_Q_NBHD_EXCEEDED:
            m.append(f"Warning: Neighborhood size exceeds {max_neighborhood} candidates. Only first {max_neighborhood} "
                     "neighbors are searched, results are likely incomplete!")
        if self.flags & DG_Q_MAX_MATCHES:
            m.append(f"Warning: More than {max_locations} matches found. Only first {max_locations} matches are "
                     "reported, results are likely incomplete!")
        return m


@dataclass
class HuntBatch:
    queries: List[QueryResult]
    counters: dict
    timings_ms: dict


def _pack(items: Sequence[bytes]):
    off = (C.c_uint64 * (len(items) + 1))()
    t = 0
    for i, b in enumerate(items):
        off[i] = t
        t += len(b)
    off[len(items)] = t
    return b"".join(items), off


class FmIndex:
    """The FM-index `dicey index` wrote, resident in one GPU's HBM."""

    def __init__(self, fm9_path: str, device: int = 0, selfcheck: bool = True, kmer_table: bool = True, big_table: bool = False, compact: bool = False,
                 pre5: bool = True, _lib=None):
        self._L = _lib or _capi.load()
        self._h = C.c_void_p()
        flags = (0 if selfcheck else _capi.DG_OPEN_NO_SELFCHECK) | (0 if kmer_table else _capi.DG_OPEN_NO_KMER_TABLE) | \
            (_capi.DG_OPEN_BIG_TABLE if big_table else 0) | (_capi.DG_OPEN_COMPACT if compact else 0) | (0 if pre5 else _capi.DG_OPEN_NO_PRE5)
        _capi.check(self._L, self._L.dg_index_open(fm9_path.encode(), device, flags, C.byref(self._h)))

    def share(self) -> "FmIndex":
        """A second handle on the same resident index (own stream and workspaces) for a concurrent host thread; close it
        before this one."""
        other = FmIndex.__new__(FmIndex)
        other._L = self._L
        other._h = C.c_void_p()
        _capi.check(self._L, self._L.dg_index_share(self._h, C.byref(other._h)))
        return other

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.dg_index_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    @property
    def handle(self):
        return self._h

    def stats(self) -> dict:
        s = _capi.IndexStats()
        _capi.check(self._L, self._L.dg_index_stats(self._h, C.byref(s)))
        return {"n": s.n, "sigma": s.sigma, "file_bytes": s.file_bytes, "hbm_bytes": s.hbm_bytes,
                "load_seconds": s.load_seconds, "derive_seconds": s.derive_seconds,
                "code_len": {b: s.code_len[b] for b in range(256) if s.code_len[b]}}

    def size(self) -> int:  # fm_index.size(), src/hunter.h:368
        return self.stats()["n"]

    def count(self, patterns: Sequence[bytes]) -> List[int]:
        buf, off = _pack(patterns)
        out = (C.c_uint64 * max(1, len(patterns)))()
        _capi.check(self._L, self._L.dg_count(self._h, buf, off, len(patterns), out))
        return list(out[:len(patterns)])

    def neighborhood_count(self, seqs: Sequence[bytes], distance: int = 1, hamming: bool = False,
                           max_neighborhood: int = 10000) -> List[tuple]:
        """(forward, reverse-complement) occurrence totals over the neighbourhood of every sequence (src/padlock.h:392-421)."""
        buf, off = _pack(seqs)
        n = len(seqs)
        fw = (C.c_uint64 * max(1, n))()
        rv = (C.c_uint64 * max(1, n))()
        _capi.check(self._L, self._L.dg_neighborhood_count(self._h, distance, 1 if hamming else 0, max_neighborhood, buf, off, n, fw, rv))
        return [(fw[i], rv[i]) for i in range(n)]

    def locate(self, patterns: Sequence[bytes]) -> List[List[int]]:
        buf, off = _pack(patterns)
        lp = C.POINTER(_capi.Locations)()
        _capi.check(self._L, self._L.dg_locate(self._h, buf, off, len(patterns), C.byref(lp)))
        try:
            L = lp.contents
            return [list(L.pos[L.off[i]:L.off[i + 1]]) for i in range(len(patterns))]
        finally:
            self._L.dg_locations_free(lp)

    def extract(self, ranges: Sequence[tuple]) -> List[bytes]:
        n = len(ranges)
        lo = (C.c_uint64 * max(1, n))(*[r[0] for r in ranges])
        hi = (C.c_uint64 * max(1, n))(*[r[1] for r in ranges])
        off = (C.c_uint64 * max(1, n))()
        t = 0
        for i, (a, b) in enumerate(ranges):
            off[i] = t
            t += b - a + 1
        out = C.create_string_buffer(max(1, t))
        _capi.check(self._L, self._L.dg_extract(self._h, lo, hi, n, out, off))
        raw = out.raw
        return [raw[off[i]:off[i] + (ranges[i][1] - ranges[i][0] + 1)] for i in range(n)]

    def _unpack(self, rp, buf=None, off=None) -> HuntBatch:
        R = rp.contents
        if R.compact:  # ABI 5: compact records -> the classic arrays, from the caller's own query bytes (dg_hunt_expand)
            _capi.check(self._L, self._L.dg_hunt_expand(rp, buf, off))
        qs = []
        seqbuf = C.string_at(R.qseq, R.qoff[R.nq]) if R.nq else b""
        _capi.check(self._L, self._L.dg_hunt_rows(rp))  # the two rows of every hit from its compact description
        st = R.aln_stride
        for i in range(R.nq):
            q = QueryResult(seqbuf[R.qoff[i]:R.qoff[i + 1]].decode("latin-1"), R.qdistance[i], R.qflags[i], R.qnondna[i])
            for h in range(R.hit_off[i], R.hit_off[i + 1]):
                H = R.hits[h]
                ra = C.string_at(C.addressof(R.refalign.contents) + h * st, H.aln_len).decode("latin-1")
                qa = C.string_at(C.addressof(R.queryalign.contents) + h * st, H.aln_len).decode("latin-1")
                q.hits.append(DnaHit(H.score, H.chr, H.start, chr(H.strand), ra, qa))
            qs.append(q)
        ctr = {"ext_steps": R.ctr_ext_steps, "leaves": R.ctr_leaves, "sa_reads": R.ctr_sa_reads,
               "win_bytes": R.ctr_win_bytes, "tab_reads": R.ctr_tab_reads, "filter_probes": R.ctr_filter_probes, "nhits": R.nhits}
        tm = {"total": R.ms_total, "search": R.ms_search, "select": R.ms_select, "locate": R.ms_locate,
              "verify": R.ms_verify}
        return HuntBatch(qs, ctr, tm)

    @staticmethod
    def _params(distance, hamming, forward_only, max_locations, max_neighborhood, compact, max_query_len):
        if compact is None:  # the compact delivery of ABI 5 is what this mirror uses unless told otherwise (the GPU suite runs both)
            compact = not os.environ.get("DICEY_CLASSIC_RESULTS")
        return _capi.HuntParams(distance, int(hamming), int(forward_only), max_locations, max_neighborhood, int(max_query_len),
                                _capi.DG_HUNT_COMPACT if compact else 0)

    def hunt(self, queries: Sequence[str], seqlen: Sequence[int], distance: int = 1, hamming: bool = False,
             forward_only: bool = False, max_locations: int = 1000, max_neighborhood: int = 10000, compact=None,
             max_query_len: int = 0) -> HuntBatch:
        """seqlen[i] = faidx length + 1 (src/util.h:201)."""
        buf, off = _pack([q.encode("latin-1") if isinstance(q, str) else q for q in queries])
        sl = (C.c_uint32 * len(seqlen))(*seqlen)
        p = self._params(distance, hamming, forward_only, max_locations, max_neighborhood, compact, max_query_len)
        rp = C.POINTER(_capi.HuntResult)()
        _capi.check(self._L, self._L.dg_hunt(self._h, C.byref(p), sl, len(seqlen), buf, off, len(queries), C.byref(rp)))
        try:
            return self._unpack(rp, buf, off)
        finally:
            self._L.dg_hunt_result_free(rp)

    def hunt_submit(self, queries: Sequence[str], seqlen: Sequence[int], distance: int = 1, hamming: bool = False,
                    forward_only: bool = False, max_locations: int = 1000, max_neighborhood: int = 10000, compact=None,
                    max_query_len: int = 0):
        """dg_hunt_submit: the batch runs on a helper thread of the library; collect with hunt_wait (three per handle at a time,
        waited for in the order they were submitted)."""
        buf, off = _pack([q.encode("latin-1") if isinstance(q, str) else q for q in queries])
        sl = (C.c_uint32 * len(seqlen))(*seqlen)
        p = self._params(distance, hamming, forward_only, max_locations, max_neighborhood, compact, max_query_len)
        t = C.c_void_p()
        _capi.check(self._L, self._L.dg_hunt_submit(self._h, C.byref(p), sl, len(seqlen), buf, off, len(queries), C.byref(t)))
        return (t, buf, off)

    def hunt_wait(self, ticket) -> HuntBatch:
        t, buf, off = ticket
        rp = C.POINTER(_capi.HuntResult)()
        _capi.check(self._L, self._L.dg_hunt_wait(t, C.byref(rp)))
        try:
            return self._unpack(rp, buf, off)
        finally:
            self._L.dg_hunt_result_free(rp)


class Thal:
    """primer3 thal() as `dicey search` uses it (src/silica.h:316-329,437,511): END1, temponly, 37 C."""

    def __init__(self, config_dir: str, mv: float = 50.0, dv: float = 1.5, dntp: float = 0.6, dna_conc: float = 50.0,
                 device: int = 0, _lib=None):
        self._L = _lib or _capi.load()
        self._h = C.c_void_p()
        _capi.check(self._L, self._L.dg_thal_open(config_dir.encode(), mv, dv, dntp, dna_conc, device, C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None) and self._h.value:
            self._L.dg_thal_close(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def tm(self, pairs):
        """pairs: sequence of (oligo1, oligo2) strings -> list of (temp, align_end_1, align_end_2)"""
        items = []
        for a, b in pairs:
            items += [a.encode(), b.encode()]
        buf, off = _pack(items)
        n = len(pairs)
        t = (C.c_double * max(1, n))()
        e1 = (C.c_int32 * max(1, n))()
        e2 = (C.c_int32 * max(1, n))()
        _capi.check(self._L, self._L.dg_thal_batch(self._h, buf, off, n, t, e1, e2))
        return [(t[i], e1[i], e2[i]) for i in range(n)]


def search_sites(ix: "FmIndex", th: "Thal", primers: Sequence[str], seqlen: Sequence[int], kmer: int = 15, distance: int = 1,
                 hamming: bool = False, max_locations: int = 10000, max_neighborhood: int = 10000, cut_temp: float = 45.0):
    """Binding sites of a primer batch (src/silica.h:429-573).  Returns (sites, match_temp, pflags, nhits): sites are dicts
    in the reference's push order (primer-major, forward-strand hits first)."""
    L = ix._L
    buf, off = _pack([p.encode() for p in primers])
    sl = (C.c_uint32 * len(seqlen))(*seqlen)
    p = _capi.SearchParams(distance, int(hamming), max_locations, max_neighborhood, kmer, cut_temp)
    rp = C.POINTER(_capi.SearchResult)()
    _capi.check(L, L.dg_search_sites(ix.handle, th._h, C.byref(p), sl, len(seqlen), buf, off, len(primers), C.byref(rp)))
    try:
        R = rp.contents
        sites = []
        for i in range(R.nsites):
            s = R.sites[i]
            g = C.string_at(C.addressof(R.genome_pool.contents) + s.genome_off, s.genome_len).decode("latin-1")
            sites.append({"ref": s.ref, "pos": s.pos, "primer": s.primer, "on_for": bool(s.on_for), "temp": s.temp,
                          "perf_temp": s.perf_temp, "genome": g})
        search_sites.last_ms_device = R.ms_device
        return sites, [R.match_temp[i] for i in range(R.nprimers)], [R.pflags[i] for i in range(R.nprimers)], R.nhits
    finally:
        L.dg_search_result_free(rp)


def padlock_scan(ix: "FmIndex", th: "Thal", exons: Sequence[bytes], armlen: int = 20, distance: int = 1, hamming: bool = False,
                 tmdiff: int = 2, gc_min: float = 0.4, gc_max: float = 0.6):
    """Per-position values of a batch of exons (src/padlock.h:321-428; see dg_padlock_scan in include/dicey_gpu.h).
    Returns a dict of numpy arrays (pos_off, arm_gc, arm_tm, probe_gc, probe_tm, arm_count, arm_nbcount) + work counters."""
    import numpy as np
    L = ix._L
    buf, off = _pack(exons)
    p = _capi.PadlockParams(armlen, distance, 1 if hamming else 0, tmdiff, gc_min, gc_max)
    rp = C.POINTER(_capi.PadlockResult)()
    _capi.check(L, L.dg_padlock_scan(ix.handle, th._h, C.byref(p), buf, off, len(exons), C.byref(rp)))
    # The arrays are VIEWS of the library's result (128 MB for 1 000 genes: copying them cost a bench step 15 %): every view's buffer
    # object holds the owner, which hands the result back to the library when the last view is gone.
    owner = _PadlockOwner(L, rp)
    R = rp.contents
    n = R.npos

    def view(ptr, cnt, ctype, dt):
        if cnt == 0:
            return np.empty(0, dt)
        cbuf = (ctype * cnt).from_address(C.addressof(ptr.contents))
        cbuf._dicey_owner = owner
        arr = np.frombuffer(cbuf, dtype=dt, count=cnt)
        arr.flags.writeable = False
        return arr

    return {"pos_off": view(R.pos_off, R.nexons + 1, C.c_uint64, np.uint64), "arm_gc": view(R.arm_gc, n, C.c_double, np.float64),
            "arm_tm": view(R.arm_tm, n, C.c_double, np.float64), "probe_gc": view(R.probe_gc, n, C.c_double, np.float64),
            "probe_tm": view(R.probe_tm, n, C.c_double, np.float64), "arm_count": view(R.arm_count, n, C.c_int64, np.int64),
            "arm_nbcount": view(R.arm_nbcount, n, C.c_int64, np.int64), "n_arm_thal": int(R.n_arm_thal), "n_probe_thal": int(R.n_probe_thal),
            "n_arms_counted": int(R.n_arms_counted)}


class _PadlockOwner:
    """keeps a dg_padlock_result alive for the numpy views made of it (padlock_scan)"""

    def __init__(self, lib, rp):
        self._lib, self._rp = lib, rp

    def __del__(self):
        try:
            self._lib.dg_padlock_result_free(self._rp)
        except Exception:
            pass


def build_index(text: bytes, out_fm9: str, device: int = 0, _lib=None):
    """GPU counterpart of `dicey index` (src/index.h:97-123): text = SEQ1\\nSEQ2\\n...SEQk\\n, upper-case."""
    L = _lib or _capi.load()
    _capi.check(L, L.dg_index_build(text, len(text), device, out_fm9.encode()))


def check_fm9(path: str, deep: bool = True) -> dict:
    """dg_fm9_check: the acceptance check of an index file on the host (no device): every section of the sdsl csa_wt<> file accounted
    for and held against the others (src/index.h:121-122 writes it, src/hunter.h:253-256 reads it).  Returns the report (a dict with
    "ok", "sections", and "error" naming the first section that is off); never raises for a file that merely fails the check."""
    import json
    L = _capi.load()
    buf = C.create_string_buffer(1 << 16)
    rc = L.dg_fm9_check(path.encode(), 1 if deep else 0, buf, len(buf))
    rep = json.loads(buf.value.decode() or "{}")
    rep["rc"] = rc
    return rep
