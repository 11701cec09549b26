"""Dual numbers through the integrator (jacobian_∂αβ_∂gr, src/tracing/precision-solvers.jl:401-451): the
TANGENT flavour of the device integrator (real = value + ∂/∂α + ∂/∂β, gradus.jl_amd/csrc/gr_tangent.hpp) compiled for the
host, against central differences of the oracle at tolerance 1e-12.  Runs without a GPU."""
import math

import numpy as np
import pytest

import harness as Hh

ISCO = 1.2369706551751847
X = np.array([0.0, 1000.0, math.radians(30), 0.0])
RAYS = [(5.0, 4.0), (-6.0, 3.0), (2.0, -7.0), (9.0, 1.0), (-3.0, -8.0)]


def _oracle_surface(oracle, al, be, tol=1e-12):
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc={"datum": 0.0}, lambda_max=4000.0, abstol=tol, reltol=tol)
    p = oracle.trace(cfg, X, oracle.map_impact_parameters(cfg, X, [al], [be]))[0]
    g = oracle.apply_pf(cfg, np.array([p]), 4000.0, pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_INTERSECTED, r_isco=ISCO)[0]
    return p["x"][1] * abs(math.sin(p["x"][2])), g, p["lambda_max"]


def _oracle_fixed_lambda(oracle, al, be, lam, tol=1e-12):
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc=None, lambda_max=lam, abstol=tol, reltol=tol)
    p = oracle.trace(cfg, X, oracle.map_impact_parameters(cfg, X, [al], [be]))
    p["status"] = 2
    g = oracle.apply_pf(cfg, p, 2 * lam, pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_NONE, r_isco=ISCO)[0]
    return p[0]["x"][1] * abs(math.sin(p[0]["x"][2])), g


def _tangent(G, tol):
    m = G.KerrMetric(1.0, 0.998)
    cfg = G.tracing_configuration(m, X, np.zeros((1, 4)), G.DatumPlane(0.0), 4000.0, abstol=tol, reltol=tol)
    pf = G.ConstPointFunctions.redshift(m, X)
    a = np.array([r[0] for r in RAYS])
    b = np.array([r[1] for r in RAYS])
    return Hh.ray_tangent(G, cfg, pf, a, b)


def test_tangents_equal_central_differences_of_the_oracle(G, oracle):
    out = _tangent(G, 1e-11)
    h = 1e-5
    for k, (al, be) in enumerate(RAYS):
        rho0, g0, lam0 = _oracle_surface(oracle, al, be)
        assert out[k, 7] == 2
        assert out[k, 0] == pytest.approx(g0, rel=1e-8) and out[k, 1] == pytest.approx(rho0, rel=1e-8)
        fd = {}
        for name, (da, db) in (("a", (h, 0.0)), ("b", (0.0, h))):
            rp, gp, _ = _oracle_surface(oracle, al + da, be + db)
            rm, gm, _ = _oracle_surface(oracle, al - da, be - db)
            fd["g" + name], fd["r" + name] = (gp - gm) / (2 * h), (rp - rm) / (2 * h)
        got = {"ga": out[k, 2], "gb": out[k, 3], "ra": out[k, 4], "rb": out[k, 5]}
        scale_g = max(abs(fd["ga"]), abs(fd["gb"]))
        scale_r = max(abs(fd["ra"]), abs(fd["rb"]))
        for key in ("ga", "gb"):
            assert abs(got[key] - fd[key]) < 2e-5 * scale_g, (al, be, key, got[key], fd[key])
        for key in ("ra", "rb"):
            assert abs(got[key] - fd[key]) < 2e-5 * scale_r, (al, be, key, got[key], fd[key])
        det_ad = got["ra"] * got["gb"] - got["rb"] * got["ga"]
        det_fd = fd["ra"] * fd["gb"] - fd["rb"] * fd["ga"]
        assert det_ad == pytest.approx(det_fd, rel=1e-4)


def test_the_event_time_term_is_not_optional(G, oracle):
    """The Jacobian of the SURFACE map (where the ray meets the disc) against the Jacobian of the end state at the fixed
    affine time of the base ray: they differ by tens of per cent, and the transfer functions of the reference agree with the
    former (this build's difference-based Jacobians, which trace every perturbed ray to the surface, reproduce its recorded
    values at rₑ >= 7 to 1e-3)."""
    out = _tangent(G, 1e-11)
    h = 1e-5
    ratios = []
    for k, (al, be) in enumerate(RAYS[:3]):
        _, _, lam0 = _oracle_surface(oracle, al, be)
        J = np.zeros((2, 2))
        for c, (da, db) in enumerate(((h, 0.0), (0.0, h))):
            rp, gp = _oracle_fixed_lambda(oracle, al + da, be + db, lam0)
            rm, gm = _oracle_fixed_lambda(oracle, al - da, be - db, lam0)
            J[0, c], J[1, c] = (rp - rm) / (2 * h), (gp - gm) / (2 * h)
        det_surface = out[k, 4] * out[k, 3] - out[k, 5] * out[k, 2]
        ratios.append(np.linalg.det(J) / det_surface)
    assert max(abs(r - 1.0) for r in ratios) > 0.1, ratios


def test_tangent_build_takes_the_steps_of_the_plain_build(G, oracle):
    """With values-only step control (`tangent_norm` 0) the values of the tangent build == the plain host build of the same
    integrator (the controller and every branch look at values only), at the reference's tolerance; with the tangents in
    the error norm (the default since round 3) the steps are shorter where the tangents are stiff and the values agree to
    the tolerance level."""
    m = G.KerrMetric(1.0, 0.998)
    cfg = G.tracing_configuration(m, X, np.zeros((1, 4)), G.DatumPlane(0.0), 4000.0)
    pf = G.ConstPointFunctions.redshift(m, X)
    a = np.array([r[0] for r in RAYS])
    b = np.array([r[1] for r in RAYS])
    with_norm = Hh.ray_tangent(G, cfg, pf, a, b)
    Hh.lib_tangent().hht_set_tangent_norm(0)
    try:
        out = Hh.ray_tangent(G, cfg, pf, a, b)
    finally:
        Hh.lib_tangent().hht_set_tangent_norm(1)
    np.testing.assert_allclose(with_norm[:, 0:2], out[:, 0:2], rtol=1e-6)
    assert np.max(np.abs(with_norm[:, 1] / out[:, 1] - 1.0)) > 1e-13          # and they really are different step sequences
    v = np.array([G.map_impact_parameters(m, X, ai, bi) for ai, bi in zip(a, b)]).reshape(-1, 4)
    cfg2 = G.tracing_configuration(m, X, v, G.DatumPlane(0.0), 4000.0)
    pts = Hh.trace_endpoints(G, cfg2)
    rho = pts["x"][:, 1] * np.abs(np.sin(pts["x"][:, 2]))
    np.testing.assert_allclose(out[:, 1], rho, rtol=1e-11)
    np.testing.assert_allclose(out[:, 6], pts["x"][:, 0], rtol=1e-11)
    np.testing.assert_array_equal(out[:, 7].astype(int), pts["status"])


@pytest.mark.parametrize("nr,nt", [(16, 24), (19, 13), (8, 8), (5, 30)])
def test_separable_ray_sets_equal_explicit_arrays(G, nr, nt):
    """gr_rayset.sep_* (a PolarPlane handed over as three small tables): the kernel code's index map `Ray::sep_index`,
    tiled and column-major, whole tiles and ragged edges, against explicit α / β arrays in the order the Python mirror
    `lineprofiles._sep_index` states -- bit-identical rays, so bit-identical results."""
    from gradus_jl_amd.lineprofiles import _sep_index

    m = G.KerrMetric(1.0, 0.9)
    cfg = G.tracing_configuration(m, X, np.zeros((1, 4)), G.DatumPlane(0.0), 4000.0)
    pf = G.ConstPointFunctions.redshift(m, X)
    r = np.geomspace(2.0, 30.0, nr)
    th = np.linspace(0.0, 2 * math.pi, nt, endpoint=False)
    cs, sn = np.cos(th), np.sin(th)
    for tiled in (True, False):
        sep = Hh.ray_tangent_separable(G, cfg, pf, r, cs, sn, tiled)
        i, j = _sep_index(np.arange(nr * nt), nr, nt, tiled and nr >= 8 and nt >= 8)
        assert sorted(zip(i.tolist(), j.tolist())) == [(a, b) for a in range(nr) for b in range(nt)]      # a permutation
        ref = Hh.ray_tangent(G, cfg, pf, r[i] * cs[j], r[i] * sn[j])
        assert np.array_equal(np.nan_to_num(sep, nan=-1.0), np.nan_to_num(ref, nan=-1.0)), (nr, nt, tiled)
    if nr == 16:
        # the tiled order walks 8 x 8 tiles: the first 64 rays are rows 0..7 of columns 0..7, rows fastest
        i, j = _sep_index(np.arange(64), nr, nt, True)
        assert i.tolist() == list(range(8)) * 8 and j.tolist() == [c for c in range(8) for _ in range(8)]
        # and it is the permutation lineprofiles._tile_order applies to explicit arrays
        from gradus_jl_amd.lineprofiles import _tile_order

        perm = _tile_order(nr, nt)
        i, j = _sep_index(np.arange(nr * nt), nr, nt, True)
        assert np.array_equal(perm, i + nr * j)


@pytest.mark.parametrize("angle", [3, 30, 74])
def test_jacobians_satisfy_the_area_identity_of_the_ring_image(G, angle):
    """A check of the dual-number Jacobians that needs neither difference quotients nor the reference: the image of the ring
    ρ = rₑ encloses the area A(rₑ) = ½∮r(θ)² dθ (from the root finder alone), and dA/drₑ = ∮ds/|∇ρ| = ∮|∂(α,β)/∂(rₑ,g)| |dg|
    along the ring -- the first form uses (∂ρ/∂α, ∂ρ/∂β), the second is the transfer function's own Jacobian (its
    normalisation: ∫f/(g√(g✶(1-g✶))) dg✶ over both branches = dA/drₑ / (π rₑ)).  a = 0.998, rₑ = 4, observer at 10⁵."""
    from gradus_jl_amd import transfer_functions as TF

    x = np.array([0.0, 100_000.0, math.radians(angle), 0.0])
    m, tr = Hh.tangent_tracer(G, 0.998, x, 2 * x[1])
    th = np.linspace(0.0, 2 * math.pi, 1441)[:-1]

    def ring(r_e):
        return TF.find_offsets_for_radius_newton_ad(tr, np.full(th.size, r_e), th, r_min=m.inner_radius())

    def area(r_e):
        r = ring(r_e)[0]
        return 0.5 * np.sum(r * r) * (th[1] - th[0])

    dA = (area(4.0 + 1e-3) - area(4.0 - 1e-3)) / 2e-3
    r, _, g, tan = ring(4.0)
    ga, gb, ra, rb = tan[:, 2], tan[:, 3], tan[:, 4], tan[:, 5]
    al, be = r * np.cos(th), r * np.sin(th)
    ds = np.hypot(np.diff(np.r_[al, al[0]]), np.diff(np.r_[be, be[0]]))
    grad = np.hypot(ra, rb)
    assert np.sum(ds / (0.5 * (grad + np.roll(grad, -1)))) == pytest.approx(dA, rel=5e-5)
    Jinv = 1.0 / np.abs(ra * gb - rb * ga)
    dg = np.abs(np.diff(np.r_[g, g[0]]))
    # |J| ~ 1/|θ - θ*| at the two extrema of g, |dg| ~ |θ - θ*| dθ: integrable, the trapezoid sum is good to a per cent
    assert np.sum(0.5 * (Jinv + np.roll(Jinv, -1)) * dg) == pytest.approx(dA, rel=1.5e-2)
