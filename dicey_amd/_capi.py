"""ctypes binding of libdiceygpu.so (include/dicey_gpu.h).

The library is hand-written HIP for gfx950 and is the ONLY compute path of this package: if it is missing, or no
HIP device is present, every call fails loudly.  There is no CPU or PyTorch fallback.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DICEY_LIB") or os.path.join(_HERE, "libdiceygpu.so")  # DICEY_LIB: a development build (tools/)

DG_OK = 0
DG_OPEN_NO_SELFCHECK = 1
DG_OPEN_NO_KMER_TABLE = 2
DG_OPEN_COMPACT = 4
DG_OPEN_NO_PRE5 = 16
DG_OPEN_BIG_TABLE = 8
DG_Q_TOO_SHORT, DG_Q_DIST_ADJUSTED, DG_Q_MAX_MATCHES, DG_Q_NBHD_EXCEEDED = 1, 2, 4, 8
DG_HUNT_COMPACT = 1
DG_HUNT_PHASE_TIMES = 2


class DgError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"libdiceygpu error {code}: {msg}")
        self.code = code


class IndexStats(C.Structure):
    _fields_ = [("n", C.c_uint64), ("sigma", C.c_uint32), ("code_len", C.c_uint32 * 256), ("file_bytes", C.c_uint64),
                ("hbm_bytes", C.c_uint64), ("load_seconds", C.c_double), ("derive_seconds", C.c_double)]


class HuntParams(C.Structure):
    _fields_ = [("distance", C.c_uint32), ("hamming", C.c_int32), ("forward_only", C.c_int32),
                ("max_locations", C.c_uint64), ("max_neighborhood", C.c_uint32), ("max_query_len", C.c_uint32), ("flags", C.c_uint32)]


class Hit(C.Structure):
    _fields_ = [("score", C.c_int32), ("chr", C.c_uint32), ("start", C.c_uint32), ("query", C.c_uint32),
                ("aln_len", C.c_uint16), ("strand", C.c_uint8), ("reserved", C.c_uint8)]


class HuntResult(C.Structure):
    _fields_ = [("nq", C.c_size_t), ("nhits", C.c_uint64), ("hit_off", C.POINTER(C.c_uint64)), ("hits", C.POINTER(Hit)),
                ("ops_per_hit", C.c_uint32), ("aln_stride", C.c_uint32), ("ops", C.POINTER(C.c_uint32)),
                ("refalign", C.POINTER(C.c_char)), ("queryalign", C.POINTER(C.c_char)),
                ("qflags", C.POINTER(C.c_uint32)), ("qdistance", C.POINTER(C.c_uint32)), ("qnondna", C.POINTER(C.c_uint32)),
                ("qseq", C.POINTER(C.c_uint8)), ("qoff", C.POINTER(C.c_uint64)),
                ("ctr_ext_steps", C.c_uint64), ("ctr_leaves", C.c_uint64), ("ctr_sa_reads", C.c_uint64),
                ("ctr_win_bytes", C.c_uint64), ("ctr_tab_reads", C.c_uint64), ("ms_total", C.c_double), ("ms_search", C.c_double),
                ("ms_select", C.c_double), ("ms_locate", C.c_double), ("ms_verify", C.c_double),
                ("d_hits", C.c_void_p), ("d_ops", C.c_void_p),
                ("ctr_filter_probes", C.c_uint64), ("ms_search_flat", C.c_double), ("owner_", C.c_void_p),
                ("compact", C.c_uint32), ("nseq", C.c_uint32), ("chits", C.POINTER(C.c_uint32)), ("qinfo", C.POINTER(C.c_uint32)),
                ("seq_start", C.POINTER(C.c_uint64)), ("expanded_", C.c_void_p),
                ("ms_cap", C.c_double), ("cap_queries_device", C.c_uint64), ("cap_queries_host", C.c_uint64), ("cap_patterns", C.c_uint64),
                ("t_search_begin_ms", C.c_double), ("t_search_end_ms", C.c_double), ("t_base_gen", C.c_uint32), ("flat_kernel_form", C.c_uint32),
                ("stream", C.c_void_p), ("d_block", C.c_void_p), ("d_block_bytes", C.c_uint64),
                ("verify_kernel_form", C.c_uint32), ("reserved7", C.c_uint32)]


class SearchParams(C.Structure):
    _fields_ = [("distance", C.c_uint32), ("hamming", C.c_int32), ("max_locations", C.c_uint64), ("max_neighborhood", C.c_uint32),
                ("kmer", C.c_uint32), ("cut_temp", C.c_double)]


class Site(C.Structure):
    _fields_ = [("ref", C.c_uint32), ("pos", C.c_uint32), ("primer", C.c_uint32), ("on_for", C.c_uint8), ("reserved", C.c_uint8 * 3),
                ("temp", C.c_double), ("perf_temp", C.c_double), ("genome_off", C.c_uint64), ("genome_len", C.c_uint32), ("pad", C.c_uint32)]


class SearchResult(C.Structure):
    _fields_ = [("nprimers", C.c_size_t), ("nsites", C.c_uint64), ("sites", C.POINTER(Site)), ("genome_pool", C.POINTER(C.c_char)),
                ("pflags", C.POINTER(C.c_uint32)), ("match_temp", C.POINTER(C.c_double)), ("nhits", C.c_uint64), ("ms_device", C.c_double),
                ("ms_fm_search", C.c_double), ("ms_site_stage", C.c_double), ("ctr_ext_steps", C.c_uint64), ("ctr_tab_reads", C.c_uint64),
                ("ctr_filter_probes", C.c_uint64), ("ctr_sa_reads", C.c_uint64)]


class Locations(C.Structure):
    _fields_ = [("npat", C.c_size_t), ("off", C.POINTER(C.c_uint64)), ("pos", C.POINTER(C.c_uint64))]


# every symbol include/dicey_gpu.h declares; tests/test_capi_symbols.py checks the list against the header
class PadlockParams(C.Structure):
    _fields_ = [("armlen", C.c_uint32), ("distance", C.c_uint32), ("hamming", C.c_int32), ("tmdiff", C.c_uint32),
                ("gc_min", C.c_double), ("gc_max", C.c_double)]


class PadlockResult(C.Structure):
    _fields_ = [("nexons", C.c_uint64), ("npos", C.c_uint64), ("pos_off", C.POINTER(C.c_uint64)),
                ("arm_gc", C.POINTER(C.c_double)), ("arm_tm", C.POINTER(C.c_double)), ("probe_gc", C.POINTER(C.c_double)),
                ("probe_tm", C.POINTER(C.c_double)), ("arm_count", C.POINTER(C.c_int64)), ("arm_nbcount", C.POINTER(C.c_int64)),
                ("n_arm_thal", C.c_uint64), ("n_probe_thal", C.c_uint64), ("n_arms_counted", C.c_uint64)]


SYMBOLS = ["dg_index_open", "dg_index_close", "dg_index_stats", "dg_count", "dg_locate", "dg_locations_free",
           "dg_extract", "dg_hunt", "dg_hunt_result_free", "dg_hunt_device", "dg_index_build",
           "dg_index_build_device", "dg_last_error", "dg_abi_version", "dg_device_count",
           "dg_thal_open", "dg_thal_close", "dg_thal_batch", "dg_search_sites", "dg_search_result_free",
           "dg_neighborhood_count", "dg_padlock_scan", "dg_padlock_result_free", "dg_index_share",
           "dg_neighbors", "dg_buffer_free", "dg_hit_rows", "dg_hunt_rows", "dg_hunt_submit", "dg_hunt_wait", "dg_hunt_device_submit",
           "dg_chit_unpack", "dg_normalize_query", "dg_hunt_expand", "dg_index_stream", "dg_fm9_check"]

_lib = None


def load(path=None):
    """Load libdiceygpu.so (built by __graft_entry__.build() / `make -C dicey_amd/csrc`). Raises if absent."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or LIB_PATH
    if not os.path.exists(p):
        raise ImportError(f"{p} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). dicey_amd has no CPU fallback.")
    import sys
    if "torch" in sys.modules:  # torch ships its own HIP runtime, which only finds the GPU if it initialises first
        try:
            sys.modules["torch"].cuda.is_available()
        except Exception:
            pass
    L = C.CDLL(p)
    vp, u64p, u8p, u32p = C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint8), C.POINTER(C.c_uint32)
    L.dg_last_error.restype = C.c_char_p
    L.dg_index_open.argtypes = [C.c_char_p, C.c_int, C.c_uint32, C.POINTER(vp)]
    L.dg_fm9_check.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, C.c_size_t]
    L.dg_index_share.argtypes = [vp, C.POINTER(vp)]
    L.dg_index_close.argtypes = [vp]
    L.dg_index_stream.argtypes = [vp]
    L.dg_index_stream.restype = vp
    L.dg_index_close.restype = None
    L.dg_index_stats.argtypes = [vp, C.POINTER(IndexStats)]
    L.dg_count.argtypes = [vp, C.c_char_p, u64p, C.c_size_t, u64p]
    L.dg_locate.argtypes = [vp, C.c_char_p, u64p, C.c_size_t, C.POINTER(C.POINTER(Locations))]
    L.dg_locations_free.argtypes = [C.POINTER(Locations)]
    L.dg_locations_free.restype = None
    L.dg_extract.argtypes = [vp, u64p, u64p, C.c_size_t, C.c_char_p, u64p]
    L.dg_padlock_scan.argtypes = [vp, vp, C.POINTER(PadlockParams), C.c_char_p, u64p, C.c_size_t, C.POINTER(C.POINTER(PadlockResult))]
    L.dg_padlock_result_free.argtypes = [C.POINTER(PadlockResult)]
    L.dg_padlock_result_free.restype = None
    L.dg_neighborhood_count.argtypes = [vp, C.c_uint32, C.c_int, C.c_uint32, C.c_char_p, u64p, C.c_size_t, u64p, u64p]
    L.dg_hunt.argtypes = [vp, C.POINTER(HuntParams), u32p, C.c_uint32, C.c_char_p, u64p, C.c_size_t,
                          C.POINTER(C.POINTER(HuntResult))]
    L.dg_hunt_device.argtypes = [vp, C.POINTER(HuntParams), u32p, C.c_uint32, vp, vp, C.c_size_t, C.c_uint64, C.c_int,
                                 C.POINTER(C.POINTER(HuntResult))]
    L.dg_hunt_result_free.argtypes = [C.POINTER(HuntResult)]
    L.dg_hunt_result_free.restype = None
    L.dg_hunt_rows.argtypes = [C.POINTER(HuntResult)]
    L.dg_hunt_submit.argtypes = [vp, C.POINTER(HuntParams), u32p, C.c_uint32, C.c_char_p, u64p, C.c_size_t, C.POINTER(vp)]
    L.dg_hunt_wait.argtypes = [vp, C.POINTER(C.POINTER(HuntResult))]
    L.dg_hunt_device_submit.argtypes = [vp, C.POINTER(HuntParams), u32p, C.c_uint32, vp, vp, C.c_size_t, C.c_uint64, C.c_int, C.POINTER(vp)]
    L.dg_chit_unpack.argtypes = [C.POINTER(HuntResult), C.c_uint64, C.c_uint32, C.POINTER(Hit), C.POINTER(u32p)]
    L.dg_normalize_query.argtypes = [C.c_char_p, C.c_uint32, C.c_char_p, u32p]
    L.dg_hunt_expand.argtypes = [C.POINTER(HuntResult), C.c_char_p, u64p]
    L.dg_hit_rows.argtypes = [C.POINTER(Hit), u32p, C.c_uint32, C.c_char_p, C.c_uint32, C.c_char_p, C.c_char_p]
    L.dg_index_build.argtypes = [C.c_char_p, C.c_uint64, C.c_int, C.c_char_p]
    L.dg_index_build_device.argtypes = [vp, C.c_uint64, C.c_int, C.c_char_p]
    L.dg_thal_open.argtypes = [C.c_char_p, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(vp)]
    L.dg_thal_close.argtypes = [vp]
    L.dg_thal_close.restype = None
    L.dg_thal_batch.argtypes = [vp, C.c_char_p, u64p, C.c_size_t, C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.dg_search_sites.argtypes = [vp, vp, C.POINTER(SearchParams), u32p, C.c_uint32, C.c_char_p, u64p, C.c_size_t,
                                  C.POINTER(C.POINTER(SearchResult))]
    L.dg_search_result_free.argtypes = [C.POINTER(SearchResult)]
    L.dg_search_result_free.restype = None
    L.dg_neighbors.argtypes = [C.c_char_p, C.c_uint32, C.c_uint32, C.c_int, C.c_uint32, C.POINTER(vp), u64p, C.POINTER(C.c_int)]
    L.dg_buffer_free.argtypes = [vp]
    L.dg_buffer_free.restype = None
    if path is None:
        _lib = L
    return L


def check(L, rc):
    if rc != DG_OK:
        raise DgError(rc, (L.dg_last_error() or b"").decode(errors="replace"))
