#!/usr/bin/env python3
"""Host-side scaling of the capped neighbourhood enumeration (dg_neighbors, no GPU needed): ms per 25-mer strand at edit distance 2
with 1 ... N threads.  Measurement aid for DESIGN.md "the cap"."""
import ctypes as C, os, random, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dicey_amd import _capi
L = _capi.load()
rng = random.Random(3)
qs = ["".join(rng.choice("ACGT") for _ in range(25)).encode() for _ in range(1024)]


def one(q):
    out = C.c_void_p(); cnt = C.c_uint64(); fired = C.c_int()
    assert L.dg_neighbors(q, len(q), 2, 0, 10000, C.byref(out), C.byref(cnt), C.byref(fired)) == 0
    L.dg_buffer_free(out)


for nt in (1, 8, 32, 64, 128, 256):
    n = min(len(qs), max(16, nt * 4))
    parts = [qs[i:n:nt] for i in range(nt)]
    t = time.time()
    th = [threading.Thread(target=lambda p=p: [one(q) for q in p]) for p in parts]
    [x.start() for x in th]
    [x.join() for x in th]
    dt = time.time() - t
    print("%3d threads: %6.2f ms per strand of wall (%d strands, %.2f s)" % (nt, dt / n * 1e3, n, dt), flush=True)
