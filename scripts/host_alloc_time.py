"""What a pinned result block costs to get and to give back: gr_host_alloc / gr_host_free of an end-point block (637 MB at 2048²),
first and repeated, against touching a fresh pageable array of the same size.

    python scripts/host_alloc_time.py [MiB]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G
from gradus_jl_amd import _lib

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 608
n = mib << 20
ens = G.EnsembleMI355X(0)
L = _lib.load()
for k in range(4):
    p = C.c_void_p()
    t0 = time.perf_counter()
    _lib.check(L.gr_host_alloc(ens.ctx.handle, n, C.byref(p)))
    t1 = time.perf_counter()
    _lib.check(L.gr_host_free(ens.ctx.handle, p))
    t2 = time.perf_counter()
    print(f"gr_host_alloc({mib} MiB) {1e3 * (t1 - t0):8.2f} ms   gr_host_free {1e3 * (t2 - t1):8.2f} ms")
for k in range(2):
    t0 = time.perf_counter()
    a = np.empty(n, dtype=np.uint8)
    a[::4096] = 1
    t1 = time.perf_counter()
    print(f"np.empty + first touch of every page ({mib} MiB, one thread) {1e3 * (t1 - t0):8.2f} ms")
    del a
