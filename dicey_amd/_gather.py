"""ctypes mirror of include/dicey_gather.h (libdiceygather.so): the gather of per-GPU hit lists over RCCL, used by bench.py and
the tests.  No fallback: the library loads or the import of its users fails."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
SYMBOLS = ["dg_comm_unique_id", "dg_comm_open", "dg_comm_open_tcp", "dg_comm_close", "dg_gather_submit", "dg_gather_finish",
           "dg_gather_last", "dg_gather_last_to_host", "dg_comm_max_u64", "dg_comm_barrier", "dg_gather_last_error"]
ID_BYTES = 128
_lib = None


class GatherError(RuntimeError):
    pass


def load(path=None):
    global _lib
    if _lib is not None and path is None:
        return _lib
    p = path or os.path.join(_HERE, "libdiceygather.so")
    if not os.path.exists(p):
        raise OSError(f"{p} is not built (make -C dicey_amd/csrc)")
    L = C.CDLL(p)
    L.dg_gather_last_error.restype = C.c_char_p
    L.dg_comm_open.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
    L.dg_comm_open_tcp.argtypes = [C.c_int, C.c_int, C.c_int, C.c_uint64, C.c_int, C.POINTER(C.c_void_p)]
    L.dg_comm_close.argtypes = [C.c_void_p]
    L.dg_gather_submit.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64]
    L.dg_gather_finish.argtypes = [C.c_void_p, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]
    L.dg_gather_last.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_void_p), C.POINTER(C.c_uint64)]
    L.dg_gather_last_to_host.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.dg_comm_max_u64.argtypes = [C.c_void_p, C.c_uint64, C.POINTER(C.c_uint64)]
    L.dg_comm_barrier.argtypes = [C.c_void_p]
    L.dg_comm_unique_id.argtypes = [C.c_char_p]
    if path is None:
        _lib = L
    return L


def check(L, rc):
    if rc != 0:
        raise GatherError("libdiceygather error %d: %s" % (rc, (L.dg_gather_last_error() or b"").decode()))


class Comm:
    """One rank's end of the gather.  tcp_port: the host-memory test transport; else RCCL with the given unique id."""

    def __init__(self, nranks, rank, capacity, root=0, device=0, unique_id=None, tcp_port=None):
        self.L = load()
        self.nranks, self.rank, self.root = nranks, rank, root
        h = C.c_void_p()
        if tcp_port is not None:
            check(self.L, self.L.dg_comm_open_tcp(tcp_port, nranks, rank, capacity, root, C.byref(h)))
        else:
            check(self.L, self.L.dg_comm_open(unique_id, nranks, rank, device, capacity, root, C.byref(h)))
        self.h = h

    @staticmethod
    def unique_id() -> bytes:
        L = load()
        buf = C.create_string_buffer(ID_BYTES)
        check(L, L.dg_comm_unique_id(buf))
        return buf.raw

    def submit(self, ptr, nbytes, stream=None):
        check(self.L, self.L.dg_gather_submit(self.h, C.c_void_p(stream or 0), C.c_void_p(ptr or 0), nbytes))

    def finish(self):
        b, s = C.c_uint64(), C.c_uint64()
        check(self.L, self.L.dg_gather_finish(self.h, C.byref(b), C.byref(s)))
        return b.value, s.value

    def last(self, r) -> bytes:
        n = C.c_uint64()
        p = C.c_void_p()
        check(self.L, self.L.dg_gather_last(self.h, r, C.byref(p), C.byref(n)))
        buf = C.create_string_buffer(max(1, n.value))
        check(self.L, self.L.dg_gather_last_to_host(self.h, r, buf, n.value, C.byref(n)))
        return buf.raw[:n.value]

    def max_u64(self, v) -> int:
        out = C.c_uint64()
        check(self.L, self.L.dg_comm_max_u64(self.h, int(v), C.byref(out)))
        return out.value

    def barrier(self):
        check(self.L, self.L.dg_comm_barrier(self.h))

    def close(self):
        if self.h:
            check(self.L, self.L.dg_comm_close(self.h))
            self.h = None
