"""Oracle step counts of a corona's sky rays (DESIGN_measurements.md §M18, §M19): how well does a quantity known BEFORE the trace
(from a ray's initial x, v) predict how many steps the ray takes -- i.e. can the longest waves be launched first and can the 64
rays of a wave be made to take the same number of steps?

  python scripts/corona_lanes.py [lamp|disc] [n_samples=1000000] [chunk=4096]

lamp: LampPostModel(h = 10) over Kerr a = 0.998 (one position, 0.01 rad off the axis); disc: DiscCorona(co-rotating, r = 10, h = 5), a
position per sample.  Prints, per chunk of consecutive samples, the correlation of the predictor of k_sky_velocities_dealt
(gradus_mi355x.hip, restated below in numpy) with the oracle's step counts, the residual by position bucket, and the lane utilisation
of 64-ray waves formed in sample order / by class / by (class, position bucket).  CPU only (test infrastructure: the oracle)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gradus_jl_amd as G
from oracle import oracle as O

O.lib()
which = sys.argv[1] if len(sys.argv) > 1 else "lamp"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
chunk = int(sys.argv[3]) if len(sys.argv) > 3 else 4096
K = G.corona
a = 0.998
m = G.KerrMetric(1.0, a)
model = G.LampPostModel(h=10.0) if which == "lamp" else G.DiscCorona(G.SourceVelocities.co_rotating, 10.0, 5.0, seed=1)
s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
xs, vs, vsrc = K.sample_position_direction_velocity(m, model, s, n)
ocfg = O.make_config("kerr", (1.0, a), disc=(0.0, 500.0), lambda_max=5000.0)


def predicted_steps(x, v, two_sided=False):
    """sky_cost_class's number: 16 Δϕ + 36 ln(sin θ₀ / sin θ_min) along the flat-space straight line (two_sided: the variant with both
    ends of the run in ln sin θ, measured equal on the device and not shipped)"""
    r, th = x[:, 1], x[:, 2]
    er, eth, eph = v[:, 1], r * v[:, 2], r * np.sin(th) * v[:, 3]
    nrm = np.sqrt(er ** 2 + eth ** 2 + eph ** 2)
    er, eth, eph = er / nrm, eth / nrm, eph / nrm
    s_, c_ = np.sin(th), np.cos(th)
    Px, Pz = r * s_, r * c_
    dx, dy, dz = er * s_ + eth * c_, eph, er * c_ - eth * s_
    dphi = np.arctan2(np.abs(dy), dx)
    Pd = Px * dx + Pz * dz
    nx, ny, nz = -Pz * dy, Pz * dx - Px * dz, Px * dy
    s_ext = np.maximum(np.abs(nz) / np.sqrt(nx * nx + ny * ny + nz * nz + 1e-300), 1e-12)
    a0, a1 = dz * r * r - Pz * Pd, Pz - Pd * dz
    s0 = np.maximum(np.abs(s_), 1e-12)
    ok = (((a0 > 0) & (a1 > 0)) | ((a0 < 0) & (a1 < 0))) & (s_ext < s0)
    if not two_sided:
        return 16 * dphi + 36 * np.where(ok, np.log(s0 / s_ext), 0.0)
    s_dir = np.maximum(np.sqrt(dx * dx + dy * dy), 1e-12)
    return 16 * dphi + 18 * np.where(ok, np.log(s0 / s_ext) + np.log(s_dir / s_ext), np.abs(np.log(s_dir / s0)))


def utilisation(steps, order):
    w = steps[order][:steps.size // 64 * 64].reshape(-1, 64)
    return w.mean() / w.max(axis=1).mean()


for first in [int(f * n) for f in (0.0, 0.05, 0.25, 0.45, 0.52, 0.58, 0.7, 0.9)] + [n - chunk]:
    sl = slice(first, first + chunk)
    x = xs[sl] if xs.ndim == 2 else np.tile(xs, (chunk, 1))
    ref, st = O.trace(ocfg, x if which == "disc" else x[0], vs[sl], stats=True)
    steps = (st["accepted"] + st["rejected"]).astype(np.float64)
    sin0 = np.abs(np.sin(x[:, 2]))
    pb = np.minimum(7, (sin0 * 8).astype(int))
    line = f"{first:8d}: steps mean {steps.mean():5.0f} min {steps.min():4.0f} max {steps.max():4.0f}"
    for name, two in (("shipped", False), ("two-sided", True)):
        c = predicted_steps(x, vs[sl], two)
        cl = np.minimum(47 if two else 31, (c / 12).astype(int))
        res = steps - c
        line += (f" | {name}: corr {np.corrcoef(c, steps)[0, 1]:.3f} residual {res.mean():4.0f} ± {res.std():4.1f}; lanes: sample order "
                 f"{utilisation(steps, np.arange(steps.size)):.2f}, by class {utilisation(steps, np.argsort(-cl, kind='stable')):.2f}, "
                 f"by (class, position bucket) {utilisation(steps, np.lexsort((pb, -cl))):.2f}")
    print(line, flush=True)
