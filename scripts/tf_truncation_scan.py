#!/usr/bin/env python3
"""f-4 evidence: the eleven recorded transfer-function statistics of the reference
(test/smoke-tests/cunningham-transfer-functions.jl:25-39) against THIS build's samples, (i) with all 114 samples as the
reference's current source stores them, (ii) with the stored calls of the two golden-section searches truncated to their
first k_min / k_max calls (114 -> 80 + k_min + k_max samples: what `N = count(data.mask); data.data[:, 1:N]` of
cunningham-transfer-functions.jl:303-334 keeps if Optim's search returns before its 16 iterations).

Runs on the host build of the tangent integrator (tests/harness.py; no GPU), reference root finder, tangents in the error
norm (the library's default).  Writes profiles/r3_tf_truncation.json.

    python scripts/tf_truncation_scan.py
"""
import json
import math
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import gradus_jl_amd as G      # noqa: E402
import harness as Hh           # noqa: E402

GOLD = {(3, 4.0): 0.14048899037409682, (35, 4.0): 0.10846177995555085, (74, 4.0): 0.05550300700779827,
        (85, 4.0): 0.03602870590038378, (30, 4.0): 0.11958152396826184, (30, 7.0): 0.12205125501900763,
        (30, 10.0): 0.1265019201038228, (30, 15.0): 0.12875961522283233, (30, 300.0): 0.13378948600255888,
        (30, 800.0): 0.13470290875241375, (30, 1000.0): 0.13319637850028626}
TOL = {k: (1e-2 * v if k[1] >= 300 else 1e-3) for k, v in GOLD.items()}      # rtol 1e-2 for the large radii, atol 1e-3 else


def statistic(th, g, J, re, keep):
    gg, JJ = g[keep], J[keep]
    gs = (gg - gg.min()) / np.ptp(gg)
    f = gg * np.sqrt(gs * (1 - gs)) * np.ptp(gg) * JJ / (math.pi * re)
    return float(np.mean(f * gs))


def main():
    out = {"note": "statistic = mean(f g*) of cunningham_transfer_function(KerrMetric(1, 0.998), x = (0, 1e5, angle, 0), ThinDisc, r_e; N = 80); "
                   "k_min / k_max = stored calls of the g_min / g_max golden-section search (17 = all, the current source)",
           "cases": []}
    for ang in (3, 35, 74, 85, 30):
        radii = [r for (a, r) in GOLD if a == ang]
        x = np.array([0.0, 100_000.0, math.radians(ang), 0.0])
        m, tr = Hh.tangent_tracer(G, 0.998, x, 2 * x[1])
        raw = []
        G.transfer_functions.cunningham_transfer_functions(m, x, G.ThinDisc(0.0, float("inf")), radii, N=80, tracer=tr,
                                                           root_finder="reference", _raw=raw)
        D = raw[0][0]
        for k, re in enumerate(radii):
            th, g, J, t = D[k]
            gold = GOLD[(ang, re)]
            scan = []
            for kmin in range(18):
                for kmax in range(18):
                    keep = np.ones(114, bool)
                    keep[80 + kmin:97] = False                       # g_min search: columns 80..96 in call order
                    for j in range(kmax, 17):                        # g_max search: columns 113, 112, ... in call order
                        keep[113 - j] = False
                    scan.append((abs(statistic(th, g, J, re, keep) - gold), kmin, kmax))
            scan.sort()
            full = statistic(th, g, J, re, np.ones(114, bool))
            only_min = sorted((abs(statistic(th, g, J, re, np.r_[np.ones(80 + kk, bool), np.zeros(17 - kk, bool), np.ones(17, bool)]) - gold), kk)
                              for kk in range(18))
            e = {"angle_deg": ang, "r_e": re, "recorded": gold, "tolerance_of_the_reference_test": TOL[(ang, re)],
                 "all_114_samples": full, "diff_114": full - gold, "within_tolerance_114": abs(full - gold) <= TOL[(ang, re)],
                 "best_truncation": {"k_min": scan[0][1], "k_max": scan[0][2], "samples": 80 + scan[0][1] + scan[0][2], "abs_diff": scan[0][0]},
                 "best_with_full_gmax_search": {"k_min": only_min[0][1], "abs_diff": only_min[0][0]},
                 "next_best": [{"k_min": a, "k_max": b, "abs_diff": d} for d, a, b in scan[1:4]]}
            out["cases"].append(e)
            print(f"({ang:2d}, {re:6.1f}) 114: {full - gold:+.2e} {'ok ' if e['within_tolerance_114'] else 'OUT'} | best ({scan[0][1]:2d},{scan[0][2]:2d}) {scan[0][0]:.1e}"
                  f" | g_max full: k_min {only_min[0][1]:2d} {only_min[0][0]:.1e}")
    with open(os.path.join(ROOT, "profiles", "r3_tf_truncation.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
