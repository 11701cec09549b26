"""Time a render through GR_METRIC_TABULATED next to the same metric's fused kernels (one MI355X).
    python scripts/tabmetric_bench.py [--sizes 1024,2048] [--reps 3] [--grid 8,32] [--out gpurun_out/tabmetric.json]
kernel_ms is gr_stats.kernel_ms of the blocking host call (start of the call's device work -> end of the trace kernel); the
first render of each case (table upload, longest-first tile order learned) is not counted."""
import argparse
import json
import math
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401,E402  (one HIP runtime per process: torch first, as in tests/conftest.py)

import gradus_jl_amd as G  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--sizes", default="1024,2048")
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--grid", default="24,96")
    ap.add_argument("--metrics", default="kerr,johannsen")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    m_r, n_theta = (int(t) for t in args.grid.split(","))
    ens = G.EnsembleMI355X(0)
    x = np.array([0.0, 1000.0, math.radians(75.0), 0.0])
    bases = {"kerr": G.KerrMetric(1.0, 0.998), "johannsen": G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0)}
    rows = []
    for name in args.metrics.split(","):
        base = bases[name]
        tm = G.TabulatedMetric(base, m_r=m_r, n_theta=n_theta, max_refinements=0)
        d = G.ThinDisc(base.isco(), 50.0)
        for size in (int(s) for s in args.sizes.split(",")):
            imgs = {}
            for label, m in (("fused", base), ("tabulated", tm)):
                pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
                ms = []
                for rep in range(args.reps + 1):
                    a, b, img, st = G.rendergeodesics(m, x, d, 2000.0, image_width=size, image_height=size, alpha_lims=(-60, 60),
                                                      beta_lims=(-35, 35), pf=pf, ensemble=ens, stats=True)
                    if rep:
                        ms.append(st["kernel_ms"])
                imgs[label] = img
                rows.append({"metric": name, "size": size, "path": label, "kernel_ms": min(ms), "kernel_ms_all": ms,
                             "steps_per_ray": (st["accepted_steps"] + st["rejected_steps"]) / st["rays"],
                             "rays_per_s": size * size / (min(ms) * 1e-3)})
                print(rows[-1], flush=True)
            both = ~np.isnan(imgs["fused"]) & ~np.isnan(imgs["tabulated"])
            rel = np.abs(imgs["tabulated"][both] / imgs["fused"][both] - 1.0)
            rows.append({"metric": name, "size": size, "path": "tabulated vs fused",
                         "flips": int(np.sum(np.isnan(imgs["fused"]) != np.isnan(imgs["tabulated"]))),
                         "rel_median": float(np.median(rel)), "rel_q999": float(np.quantile(rel, 0.999)), "rel_max": float(rel.max()),
                         "n_over_1e-6": int(np.sum(rel > 1e-6)), "table_mb": tm.table.nbytes / 1e6, "grid": [tm.m_r, tm.n_theta],
                         "fit_errors": list(tm.errors)})
            print(rows[-1], flush=True)
    if args.out:
        os.makedirs(os.path.dirname(args.out), exist_ok=True)
        with open(args.out, "w") as f:
            json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
