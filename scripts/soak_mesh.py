#!/usr/bin/env python3
"""Soak for GR_DISC_MESH: N random scenes (metric, observer, tolerance, a random mesh of one of four kinds) through the C ABI
against the oracle.  A mesh hit ends at a step end, so the acceptance is tests/test_mesh_geometry.py's: same decision (flips
counted), free rays to rtol, hits stop within 4 in affine time of the oracle's stop and ON the oracle's mesh-less trajectory.

    python scripts/soak_mesh.py [n_scenes=200] [seed=1]
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
rng = np.random.default_rng(seed)
U = lambda a, b: float(rng.uniform(a, b))
ens = G.EnsembleMI355X(0)
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
    return "shards", shards(int(rng.integers(200, 1500)), U(8.0, 20.0), 10.0 ** U(1.5, 4.0), int(rng.integers(1 << 30)))


tot = flips = hits = free_bad = hit_far = traj_bad = traj_checked = 0
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
    lam = 2.0 * robs + 200.0
    got = G.tracegeodesics(m, x, v, G.MeshAccretionGeometry(mesh), (0.0, lam), ensemble=ens, abstol=tol, reltol=tol)
    ocfg = lambda **kw: oracle.make_config(name, params, abstol=tol, reltol=tol, **kw)
    ref = oracle.trace(ocfg(disc={"mesh": mesh}, lambda_max=lam), x, v)
    rtol = max(1e-6, 1e3 * tol)
    mism = (got["status"] != ref["status"]) | ((ref["status"] == 2) & (np.abs(got["lambda_max"] - ref["lambda_max"]) >= 4.0))
    clean = ((ref["flags"] & 0xFFFF) == 0) & ((got["flags"] & 0xFFFF) == 0)
    free = ~mism & (ref["status"] == 3) & clean
    err = np.zeros(free.sum())
    for f in ("x", "v"):
        err = np.maximum(err, (np.abs(got[f][free] - ref[f][free]) / np.maximum(np.abs(ref[f][free]), 1.0)).max(axis=1))
    hit = ~mism & (ref["status"] == 2)
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
print(f"{n_scenes} scenes, {tot} rays: {hits} mesh hits agreed, {flips} flips ({100.0 * flips / tot:.4f} %), free rays beyond rtol {free_bad}, "
      f"hit states checked against the oracle's trajectory {traj_checked}, off it {traj_bad} (worst {worst_traj:.2e}); scenes outside the tests' acceptance: {len(bad_scenes)}")
for b in bad_scenes[:40]:
    print("  ", b)
