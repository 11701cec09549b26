"""ctypes binding of tests/host_harness.cpp: the HIP integrator compiled for the host (g++)."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
SO = os.path.join(HERE, "libhost_harness.so")
SRC = [os.path.join(HERE, "host_harness.cpp"), os.path.join(ROOT, "gradus.jl_amd", "csrc", "gr_device.hpp"),
       os.path.join(ROOT, "include", "gradus_mi355x.h")]


def build():
    if not os.path.exists(SO) or any(os.path.getmtime(s) > os.path.getmtime(SO) for s in SRC):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", SO, SRC[0]])
    return SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _lib.hh_step_log.restype = C.c_int64
    return _lib


def render_endpoints(G, config):
    L = G._lib
    cfg, pl = config.abi_config(), config.abi_plane()
    n = pl.width * pl.height
    rg = L.gr_range(0, n, max(n, 1), 1)
    out = np.zeros(n, dtype=L.POINT_DTYPE)
    lib().hh_render_endpoints(C.byref(cfg), C.byref(pl), C.byref(rg), C.c_void_p(out.ctypes.data))
    return out


def render(G, config, pf):
    from gradus_jl_amd.rendering import abi_pointfunction

    L = G._lib
    cfg, pl = config.abi_config(), config.abi_plane()
    n = pl.width * pl.height
    rg = L.gr_range(0, n, max(n, 1), 1)
    s, keep = abi_pointfunction(pf)
    img = np.zeros(n)
    lib().hh_render(C.byref(cfg), C.byref(pl), C.byref(rg), C.byref(s), C.c_void_p(img.ctypes.data))
    return img.reshape(pl.width, pl.height).T


def trace_endpoints(G, config):
    L = G._lib
    cfg = config.abi_config()
    v = np.ascontiguousarray(config.velocity, dtype=np.float64)
    x = np.ascontiguousarray(config.position, dtype=np.float64)
    n = v.shape[0]
    out = np.zeros(n, dtype=L.POINT_DTYPE)
    lib().hh_trace_endpoints(C.byref(cfg), C.c_void_p(x.ctypes.data), C.c_int64(0 if x.ndim == 1 else 4),
                             C.c_void_p(v.ctypes.data), C.c_int64(n), C.c_void_p(out.ctypes.data))
    return out


def step_log(G, config, i, cap=100000):
    L = G._lib
    cfg, pl = config.abi_config(), config.abi_plane()
    out = np.zeros(1, dtype=L.POINT_DTYPE)
    t, h = np.zeros(cap), np.zeros(cap)
    n = lib().hh_step_log(C.byref(cfg), C.byref(pl), C.c_int64(i), C.c_void_p(out.ctypes.data),
                          C.c_void_p(t.ctypes.data), C.c_void_p(h.ctypes.data), C.c_int64(cap))
    return out[0], t[:n], h[:n]
