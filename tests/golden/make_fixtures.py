"""Generates tests/golden/*.npz: per-ray endpoint records and redshift images produced by the CPU
oracle (oracle/gradus_oracle.c) AFTER it has been pinned on the reference's golden values
(tests/test_oracle_golden.py).  The GPU tests compare the HIP path with these files as well as
with the live oracle, so parity can be checked even where the oracle cannot be rebuilt.

    python tests/golden/make_fixtures.py
"""
import math
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import oracle as O   # noqa: E402


def main():
    # (1) the reference's smoke-test scene, Kerr a = 0 + ThinDisc(0, 40), 20 x 20
    x = np.array([0.0, 100.0, math.radians(85), 0.0])
    cfg = O.make_config("kerr", (1.0, 0.0), disc=(0.0, 40.0), lambda_max=200.0)
    v = O.render_velocities(cfg, x, (-9.5, 9.5), (-9.5, 9.5), 20, 20)
    pts = O.trace(cfg, x, v, nthreads=1)
    np.savez_compressed(os.path.join(HERE, "kerr_a0_thindisc_20x20_endpoints.npz"), points=pts,
                        x_obs=x, alims=(-9.5, 9.5), blims=(-9.5, 9.5), W=20, H=20, lambda_max=200.0,
                        metric="kerr", params=(1.0, 0.0), disc=(0.0, 40.0))
    # (2) BASELINE config C1: Kerr a = 0.998, 64 x 64, redshift ∘ filter_intersected
    cfg = O.make_config("kerr", (1.0, 0.998), disc=(0.0, 40.0), lambda_max=200.0)
    isco = O.isco(cfg)
    img, pts = O.rendergeodesics(cfg, x, (-9.5, 9.5), (-9.5, 9.5), 64, 64, pf_id=O.PF_REDSHIFT,
                                 filter_id=O.FILTER_INTERSECTED, r_isco=isco, nthreads=1, return_points=True)
    np.savez_compressed(os.path.join(HERE, "kerr_a0998_c1_64x64_redshift.npz"), image=img, status=pts["status"],
                        lambda_max_per_ray=pts["lambda_max"], x_obs=x, alims=(-9.5, 9.5), blims=(-9.5, 9.5), W=64, H=64,
                        lambda_max=200.0, metric="kerr", params=(1.0, 0.998), disc=(0.0, 40.0), r_isco=isco)
    print("wrote fixtures to", HERE)


if __name__ == "__main__":
    main()
