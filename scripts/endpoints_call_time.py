"""What a CALLER sees: `ensemble_solve_tracing_problem` at 2048² (the drop-in boundary, 637 MB of GeodesicPoints), result array
allocated inside the call as the bindings do, previous result dropped before the next call.  Three ways to own the result:
a block the library pinned with the pool of freed blocks (default), the same with the pool off (page-locking paid per call),
the caller's pageable memory.

    python scripts/endpoints_call_time.py [size]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gradus_jl_amd as G

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=N, image_height=N, alpha_lims=(-60, 60),
                             beta_lims=(-35, 35), ensemble=ens)


def calls(k):
    out = []
    pts = None
    for _ in range(k):
        pts = None                                   # the previous result is garbage now (Python frees it at once)
        t0 = time.perf_counter()
        pts, st = G.ensemble_solve_tracing_problem(ens, cfg, stats=True)
        out.append(((time.perf_counter() - t0) * 1e3, st["kernel_ms"], st["call_ms"]))
    return out


for label, pool, pinned in (("pinned, pool on (default)", 4096, "1"), ("pinned, pool off", 0, "1"), ("pageable", 4096, "0")):
    ens.set("pinned_pool_mib", pool)
    os.environ["GRADUS_MI355X_PINNED_RESULTS"] = pinned
    r = calls(6)
    print(f"{label:28s} first call {r[0][0]:7.1f} ms; then wall " + " ".join(f"{a[0]:6.1f}" for a in r[1:]) +
          f"   (device part of the last: kernel {r[-1][1]:.1f} call {r[-1][2]:.1f})")
ens.set("pinned_pool_mib", 4096)
