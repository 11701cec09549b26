#!/usr/bin/env python3
"""Attribution of the status flips whose ORACLE class is robust.

BASELINE config C2 (Kerr a = 0.998, 1024², ThinDisc(r_isco, 50)): the device and the oracle disagree on the class
(hit / miss) of ~800 of 1 048 576 pixels; on about half of them the oracle's own class survives a nudge of its tolerance
(1e-9 -> 0.9e-9), so "the pixel is ill-conditioned" does not explain them.  This script explains them, on the CPU:

  * device side = the kernel's integrator compiled for the host (tests/host_harness.cpp: the same gr_device.hpp, bit-level
    differences from hipcc's contraction only), oracle = oracle/gradus_oracle.c at 1e-9 and at 0.9e-9;
  * for every robust flip the side that HIT gives the crossing: its end point (r, θ = acos(±gtol), v^θ) and the step it
    happened in (step logs of both sides: λ_k and h_k).  The disc condition c = |r cosθ| - gtol r is negative only inside
    the wedge |cosθ| < gtol, which the ray crosses in Δλ_w = 2 gtol / |v^θ| (r cancels); the ContinuousCallback tests c at the
    step's ends and at 6 interior points (Θ = j/7).  A crossing whose wedge passage is shorter than the sample spacing
    h/7 of the step THE OTHER SIDE took there can fall between two samples: sample-grid aliasing, the reference's own
    blind spot (/root/reference/src/geometry/bootstrap.jl:43-54, interp_points = 8);
  * flips whose crossing sits at the disc's rim (ρ within one sample spacing of r_in or r_out: the condition jumps between
    its plateau c = 1 and c < 0 there) are the rim class;
  * anything else would be a bug.

Writes profiles/r3_robust_flips.json (summary + the per-pixel table).

    python scripts/robust_flips.py [size=1024]
"""
import ctypes as C
import json
import math
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gradus_jl_amd as G          # noqa: E402
import harness as Hh               # noqa: E402
from oracle import oracle as O     # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
A = 0.998
X = np.array([0.0, 1000.0, math.radians(75.0), 0.0])
ALIMS, BLIMS = (-60.0, 60.0), (-35.0, 35.0)
GTOL = 1e-2


def device_logic_endpoints(cfg, n_threads=8):
    """the kernel's integrator on the host, column blocks in parallel (ctypes releases the GIL)"""
    L = G._lib
    acfg, pl = cfg.abi_config(), cfg.abi_plane()
    n = pl.width * pl.height
    out = np.zeros(n, dtype=L.POINT_DTYPE)
    lib = Hh.lib()
    cols = np.linspace(0, pl.width, n_threads + 1).astype(int)

    def work(k):
        first, cnt = cols[k] * pl.height, (cols[k + 1] - cols[k]) * pl.height
        if cnt == 0:
            return
        rg = L.gr_range(int(first), int(cnt), int(cnt), 1)
        lib.hh_render_endpoints(C.byref(acfg), C.byref(pl), C.byref(rg), C.c_void_p(out.ctypes.data + int(first) * out.itemsize))

    th = [threading.Thread(target=work, args=(k,)) for k in range(n_threads)]
    [t.start() for t in th]
    [t.join() for t in th]
    return out


def main():
    t0 = time.time()
    m = G.KerrMetric(1.0, A)
    isco = m.isco()
    d = G.ThinDisc(isco, 50.0)
    cfg = G.render_configuration(m, X, d, 2000.0, image_width=S, image_height=S, alpha_lims=ALIMS, beta_lims=BLIMS,
                                 ensemble=G.EnsembleMI355X.__new__(G.EnsembleMI355X))
    dev = device_logic_endpoints(cfg)
    print(f"device-logic end points: {time.time() - t0:.0f} s", flush=True)
    ocfg = {tol: O.make_config("kerr", (1.0, A), disc=(isco, 50.0), lambda_max=2000.0, abstol=tol, reltol=tol) for tol in (1e-9, 0.9e-9)}
    v = O.render_velocities(ocfg[1e-9], X, ALIMS, BLIMS, S, S)
    orc = O.trace(ocfg[1e-9], X, v)
    orc2 = O.trace(ocfg[0.9e-9], X, v)
    print(f"oracle at 1e-9 and 0.9e-9: {time.time() - t0:.0f} s", flush=True)

    hit_d, hit_o, hit_o2 = dev["status"] == 2, orc["status"] == 2, orc2["status"] == 2
    flips = hit_d != hit_o
    robust = flips & (hit_o == hit_o2)
    idx = np.flatnonzero(robust)
    print(f"{S}²: flips {int(flips.sum())}, of which the oracle's class is robust {idx.size}, oracle vs nudged oracle {int((hit_o != hit_o2).sum())}", flush=True)

    acfg, pl = cfg.abi_config(), cfg.abi_plane()
    rows = []
    for i in idx:
        i = int(i)
        hitter = "device" if hit_d[i] else "oracle"
        p = dev[i] if hit_d[i] else orc[i]
        lam, r, th, vth, vr = float(p["lambda_max"]), float(p["x"][1]), float(p["x"][2]), float(p["v"][2]), float(p["v"][1])
        rho = r * abs(math.sin(th))
        # the steps both sides took around λ_hit
        _, tdev, hdev = Hh.step_log(G, cfg, i)               # (λ after each attempted step, step size h) of the device logic
        _, torc, _ = O.trace_steps(ocfg[1e-9], X, v[i])       # λ of every accepted step of the oracle
        def step_at(ts, lam):
            ts = np.asarray(ts)
            k = int(np.searchsorted(ts, lam, side="left"))
            k = min(max(k, 1), ts.size - 1)
            return float(ts[k] - ts[k - 1])
        tacc = np.concatenate([[0.0], np.unique(tdev)])
        h_dev, h_orc = step_at(tacc, lam), step_at(np.concatenate([[0.0], torc]) if torc[0] != 0.0 else torc, lam)
        h_missing = h_orc if hitter == "device" else h_dev      # the step of the side that did NOT see the crossing
        wedge = 2.0 * GTOL / max(abs(vth), 1e-300)              # affine length of the passage through |cosθ| < gtol
        frac = wedge / h_missing                                # as a fraction of that step; the samples are 1/7 apart
        # radial motion across one sample spacing: does the rim (c jumps to its plateau 1) lie inside it?
        drho = abs(vr) * h_missing / 7.0 + abs(r * math.cos(th) * vth) * h_missing / 7.0
        rim = min(abs(rho - isco), abs(rho - 50.0))
        if frac < 1.0 / 7.0:
            kind = "aliasing: wedge passage shorter than the sample spacing"
        elif rim <= 2.0 * drho:
            kind = "rim: crossing within two sample spacings of the disc's edge"
        elif frac < 2.0 / 7.0:
            kind = "aliasing (marginal): wedge passage between one and two sample spacings"
        else:
            kind = "other"
        rows.append({"pixel": i, "alpha_index": i // S, "beta_index": i % S, "hit_on": hitter, "lambda_hit": lam, "rho_hit": rho,
                     "v_theta": vth, "h_device_logic": h_dev, "h_oracle": h_orc, "wedge_passage_over_step": frac,
                     "distance_to_rim": rim, "radial_travel_per_sample": drho, "class": kind})
    kinds = {}
    for rrow in rows:
        kinds[rrow["class"]] = kinds.get(rrow["class"], 0) + 1
    summary = {"image": f"{S}x{S}", "status_flips": int(flips.sum()), "robust_flips": int(idx.size),
               "oracle_vs_nudged_oracle_flips": int((hit_o != hit_o2).sum()),
               "hit_on_device_only": int((robust & hit_d).sum()), "hit_on_oracle_only": int((robust & hit_o).sum()),
               "classes": kinds,
               "wedge_passage_over_step_quantiles": ({q: float(np.quantile([r_["wedge_passage_over_step"] for r_ in rows], q)) for q in (0.0, 0.5, 0.9, 1.0)}
                                                     if rows else None),
               "device_side": "tests/host_harness.cpp (the kernel's gr_device.hpp compiled for the host)",
               "note": "a crossing is seen if one of the 8 samples per step (Θ = j/7) falls inside the wedge |cosθ| < gtol: a passage shorter "
                       "than 1/7 of the step can be missed by either integrator depending on where ITS steps fall"}
    print(json.dumps(summary, indent=1))
    with open(os.path.join(ROOT, "profiles", "r3_robust_flips.json"), "w") as f:
        json.dump({"summary": summary, "pixels": rows}, f, indent=1)


if __name__ == "__main__":
    main()
