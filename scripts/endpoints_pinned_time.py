"""The drop-in boundary proper at 2048²: `gr_render_endpoints` (637 MB of 152-B records back to the host) into a block
the library pinned (gr_host_alloc, ABI 5: what the Julia shim wraps as its Vector{GeodesicPoint}) against the caller's own
pageable memory, fresh and warm.  Wall time of the blocking call and the device-side split of gr_stats (kernel_ms / call_ms).

    python scripts/endpoints_pinned_time.py [size]"""
import json, math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ctypes as C
import numpy as np
import gradus_jl_amd as G
from gradus_jl_amd import _lib

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=N, image_height=N, alpha_lims=(-60, 60),
                             beta_lims=(-35, 35), ensemble=ens)
acfg, pl = cfg.abi_config(), cfg.abi_plane()
n = N * N
rg = _lib.gr_range(0, n, n, 1)
L = _lib.load()
st = _lib.gr_stats()


def call(buf):
    t0 = time.perf_counter()
    _lib.check(L.gr_render_endpoints(ens.ctx.handle, C.byref(acfg), C.byref(pl), C.byref(rg), buf.ctypes.data, C.byref(st)))
    return (time.perf_counter() - t0) * 1e3, st.kernel_ms, st.call_ms


out = {"size": N, "bytes": n * 152}
call(np.zeros(n, dtype=_lib.POINT_DTYPE))                  # context warm-up (allocations, first launch)
blk = _lib.PinnedBlock(ens.ctx, n * 152)
pinned = blk.array(_lib.POINT_DTYPE, n)
out["pinned"] = [call(pinned) for _ in range(5)]                   # the kernel stores across the link itself (direct_host, default)
ens.set("direct_host", 0)
pinned2 = _lib.PinnedBlock(ens.ctx, n * 152)
banded = pinned2.array(_lib.POINT_DTYPE, n)
out["pinned_banded"] = [call(banded) for _ in range(5)]            # staging buffer in HBM, bands on two streams, copy engine
assert pinned.tobytes() == banded.tobytes()
ens.set("lds_points", 0)
out["pinned_banded_lane_stores"] = [call(banded) for _ in range(5)]   # ... with each lane storing its own 152 bytes
assert pinned.tobytes() == banded.tobytes()
ens.set("lds_points", 1)
ens.set("direct_host", 1)
warm = np.zeros(n, dtype=_lib.POINT_DTYPE)
warm[:] = 0                                                   # pages exist
out["pageable_warm"] = [call(warm) for _ in range(5)]
out["pageable_fresh"] = [call(np.empty(n, dtype=_lib.POINT_DTYPE)) for _ in range(5)]
ref = warm.copy()
assert pinned.tobytes() == ref.tobytes()
for k in ("pinned", "pinned_banded", "pinned_banded_lane_stores", "pageable_warm", "pageable_fresh"):
    a = np.array(out[k])
    print(f"{k:26s} wall {np.median(a[:, 0]):7.2f} ms   kernel_ms {np.median(a[:, 1]):7.2f}   call_ms {np.median(a[:, 2]):7.2f}")
print(json.dumps(out))
