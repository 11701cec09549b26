#!/usr/bin/env python3
"""Soak for GR_DISC_MESH: N random scenes (metric, observer, tolerance, a random mesh of one of four kinds) through the C ABI
against the oracle.  A mesh hit ends at a step end, so the acceptance is tests/test_mesh_geometry.py's: same decision (flips
counted), free rays to rtol, hits stop within 4 in affine time of the oracle's stop and ON the oracle's mesh-less trajectory.

    python scripts/soak_mesh.py [n_scenes=200] [seed=1] [only=scene]
SOAK_HOST=1: the kernel logic compiled for the host (tests/host_harness.cpp) instead of the device (replays without a GPU);
with `only`, the scene is also traced WITHOUT its mesh on both sides, to tell mesh flips from the metric's own.
"""
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gradus_jl_amd as G
from oracle import oracle
from test_mesh_geometry import box, octahedron, shards, slab

n_scenes = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
only = int(sys.argv[3]) if len(sys.argv) > 3 else None
HOST = os.environ.get("SOAK_HOST") == "1"
rng = np.random.default_rng(seed)
U = lambda a, b: float(rng.uniform(a, b))
if HOST:
    import harness as Hh

    ens = G.EnsembleMI355X.__new__(G.EnsembleMI355X)
    ens.knobs = {}
    ens.set = lambda k, v: None
else:
    ens = G.EnsembleMI355X(0)


def device_trace(m, x, v, d, lam, tol):
    if HOST:
        return Hh.trace_endpoints(G, G.tracing_configuration(m, x, v, d, (0.0, lam), ensemble=ens, abstol=tol, reltol=tol))
    args = (m, x, v, d, (0.0, lam)) if d is not None else (m, x, v, (0.0, lam))
    return G.tracegeodesics(*args, ensemble=ens, abstol=tol, reltol=tol)

fam = [
    ("kerr", lambda: (1.0, U(-0.99, 0.99)), G.KerrMetric),
    ("johannsen", lambda: (1.0, U(0, 0.9), U(-1, 2), U(-1, 1), U(-1, 1), U(-1, 2)), G.JohannsenMetric),
    ("johannsen-psaltis", lambda: (1.0, U(0, 0.8), U(-0.5, 1)), G.JohannsenPsaltisMetric),
    ("kerr-newman", lambda: (lambda a: (1.0, a, U(0, math.sqrt(1 - a * a) * 0.95)))(U(0, 0.9)), G.KerrNewmanMetric),
    ("bumblebee", lambda: (1.0, U(0, 0.29), U(-0.5, 1)), G.BumblebeeMetric),
]


def random_mesh():
    kind = int(rng.integers(4))
    if kind == 0:
        return "slab", slab(U(2.0, 4.0), U(8.0, 20.0), int(rng.integers(3, 8)), int(rng.integers(12, 40)), U(0.8, 2.5))
    if kind == 1:
        return "box", box((U(-8, 8), U(-8, 8), U(-4, 4)), U(3.0, 6.0))
    if kind == 2:
        return "octahedron", octahedron((U(-8, 8), U(-8, 8), U(-4, 4)), U(2.5, 5.0))
    return "shards", shards(int(rng.integers(200, 900)), U(8.0, 20.0), 10.0 ** U(1.5, 4.0), int(rng.integers(1 << 30)))


tot = flips = hits = free_bad = hit_far = traj_bad = traj_checked = stalled = 0
worst_free = worst_traj = 0.0
bad_scenes = []
for sc in range(n_scenes):
    name, mkp, cls = fam[int(rng.integers(len(fam)))]
    params = mkp()
    m = cls(*params)
    robs = 10.0 ** U(1.7, 3.0)
    x = np.array([0.0, robs, math.radians(U(15, 165)), 0.0])
    tol = 10.0 ** int(rng.integers(-11, -6))
    kind, mesh = random_mesh()
    n = 40
    aa, bb = np.meshgrid(np.linspace(-U(10, 22), U(10, 22), n), np.linspace(-U(8, 16), U(8, 16), n))
    v = G.map_impact_parameters(m, x, aa.ravel(), bb.ravel())
    ens.set("kernel", int(rng.integers(2)))
    if only is not None and sc != only:
        continue
    lam = 2.0 * robs + 200.0
    got = device_trace(m, x, v, G.MeshAccretionGeometry(mesh), lam, tol)
    ocfg = lambda **kw: oracle.make_config(name, params, abstol=tol, reltol=tol, **kw)
    if only is not None:
        g0, r0 = device_trace(m, x, v, None, lam, tol), oracle.trace(ocfg(lambda_max=lam), x, v)
        d0 = g0["status"] != r0["status"]
        print(f"scene {sc} without its mesh: {int(d0.sum())} status differences; status pairs (device, oracle): "
              f"{sorted(set(zip(g0['status'][d0].tolist(), r0['status'][d0].tolist())))}; flagged rays device {int((g0['flags'] & 0xFFFF != 0).sum())} oracle {int((r0['flags'] & 0xFFFF != 0).sum())}")
    ref = oracle.trace(ocfg(disc={"mesh": mesh}, lambda_max=lam), x, v)
    rtol = max(1e-6, 1e3 * tol)
    # rays that stall at a horizon the chart does not reach (step below dtmin: flagged, e.g. Bumblebee with spin) end where the
    # last bits decide, with or without a mesh (replay a scene with `only`): counted apart
    clean = ((ref["flags"] & 0xFFFF) == 0) & ((got["flags"] & 0xFFFF) == 0)
    differ = (got["status"] != ref["status"]) | ((ref["status"] == 2) & (np.abs(got["lambda_max"] - ref["lambda_max"]) >= 4.0))
    stalled += int((differ & ~clean).sum())
    mism = differ & clean
    free = ~differ & (ref["status"] == 3) & clean
    err = np.zeros(free.sum())
    for f in ("x", "v"):
        err = np.maximum(err, (np.abs(got[f][free] - ref[f][free]) / np.maximum(np.abs(ref[f][free]), 1.0)).max(axis=1))
    hit = ~differ & (ref["status"] == 2)
    terr = 0.0
    for i in np.flatnonzero(hit)[::9]:
        cut = oracle.trace(ocfg(lambda_max=float(got["lambda_max"][i])), x, v[i:i + 1])
        e = max((np.abs(got[f][i] - cut[f][0]) / np.maximum(np.abs(cut[f][0]), 1.0)).max() for f in ("x", "v"))
        traj_checked += 1
        if not (e < rtol) or cut["status"][0] != 3:
            traj_bad += 1
        terr = max(terr, e)
    tot += got["status"].size
    flips += int(mism.sum())
    hits += int(hit.sum())
    nbad = int((err >= rtol).sum())
    free_bad += nbad
    worst_free = max(worst_free, float(np.median(err)) if err.size else 0.0)
    worst_traj = max(worst_traj, terr)
    if mism.sum() > max(2, got["status"].size // 100) or nbad > 0.02 * max(1, free.sum()) or terr >= rtol:
        bad_scenes.append((sc, name, tuple(round(p, 4) for p in params), kind, len(mesh), f"flips={int(mism.sum())} free>rtol={nbad} traj={terr:.2e} tol={tol:.0e} robs={robs:.1f}"))
print(f"{n_scenes} scenes, {tot} rays: {hits} mesh hits agreed, {flips} flips ({100.0 * flips / tot:.4f} %; {stalled} more among rays flagged as stalled), free rays beyond rtol {free_bad}, "
      f"hit states checked against the oracle's trajectory {traj_checked}, off it {traj_bad} (worst {worst_traj:.2e}); scenes outside the tests' acceptance: {len(bad_scenes)}")
for b in bad_scenes[:40]:
    print("  ", b)
