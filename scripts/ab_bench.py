"""Interleaved A/B timing of two builds of libgradus_mi355x.so in ONE process (cdna guide §5.4 rule 24).

    python scripts/ab_bench.py libA.so libB.so [--size 2048] [--rounds 12] [--shard W:R]
Reports median / min device time of the trace kernel (gr_stats.kernel_ms of gr_render)."""
import argparse, ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G
from gradus_jl_amd import _lib as L
from gradus_jl_amd.rendering import abi_pointfunction

ap = argparse.ArgumentParser()
ap.add_argument("libs", nargs="+")
ap.add_argument("--size", type=int, default=2048)
ap.add_argument("--rounds", type=int, default=12)
ap.add_argument("--shard", default=None)
ap.add_argument("--workload", default="kerr", choices=["kerr", "johannsen"])
ap.add_argument("--set", action="append", default=[], help="key=value knob for every lib")
args = ap.parse_args()

if args.workload == "kerr":
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 1000.0, math.radians(75), 0.0])
    pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
else:   # BASELINE config C4; shadow pf so that no plunging table (device trace) is needed here
    m = G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0)
    x = np.array([0.0, 1000.0, math.radians(70), 0.0])
    pf = G.ConstPointFunctions.shadow()
cfgo = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=args.size, image_height=args.size,
                              alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=G.EnsembleMI355X.__new__(G.EnsembleMI355X))
cfg, pl = cfgo.abi_config(), cfgo.abi_plane()
s, keep = abi_pointfunction(pf)
n = args.size * args.size
if args.shard:
    w, r = (int(t) for t in args.shard.split(":"))
    rg = G.shard_plan(args.size, args.size, w, r).ray_range()
else:
    rg = L.gr_range(0, n, n, 1)
img = np.zeros(rg.count)

libs = []
for path in args.libs:
    lib = C.CDLL(os.path.abspath(path))
    h = C.c_void_p()
    lib.gr_ctx_create.argtypes = [C.c_int32, C.POINTER(C.c_void_p)]
    assert lib.gr_ctx_create(0, C.byref(h)) == 0
    lib.gr_ctx_set.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    for kv in args.set:
        k, v = kv.split("=")
        assert lib.gr_ctx_set(h, k.encode(), int(v)) == 0
    lib.gr_render.argtypes = [C.c_void_p] + [C.c_void_p] * 4 + [C.c_void_p, C.c_void_p]
    lib.gr_abi_version.restype = C.c_int32
    libs.append((path, lib, h, lib.gr_abi_version()))

# ABI <= 4: gr_stats.kernel_ms spans kernel + D2H; ABI 5 splits it into kernel_ms (kernel) and call_ms (kernel + D2H).
# The column compared across builds is kernel + D2H; the kernel alone is listed for ABI-5 builds.
times = {p: [] for p, _, _, _ in libs}
ktimes = {p: [] for p, _, _, _ in libs}
st = L.gr_stats()
for rnd in range(args.rounds + 2):
    for path, lib, h, abi in libs:
        rc = lib.gr_render(h, C.byref(cfg), C.byref(pl), C.byref(s), C.byref(rg), img.ctypes.data, C.byref(st))
        assert rc == 0, rc
        if rnd >= 2:
            times[path].append(st.call_ms if abi >= 5 else st.kernel_ms)
            if abi >= 5:
                ktimes[path].append(st.kernel_ms)
for path in times:
    t = np.array(times[path])
    k = np.array(ktimes[path]) if ktimes[path] else None
    extra = f"  kernel alone median {np.median(k):8.4f} min {k.min():8.4f}" if k is not None else ""
    print(f"{os.path.basename(path):40s} kernel+D2H median {np.median(t):8.4f} ms  min {t.min():8.4f}  max {t.max():8.4f}  (n={t.size}){extra}")
