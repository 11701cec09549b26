"""CPU-side checks: the C-ABI library loads and exports every declared symbol, struct layouts
match the header, and the host-side mirror of the reference API agrees with the oracle."""
import ctypes as C
import math
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol(G):
    hdr = open(os.path.join(ROOT, "include", "gradus_mi355x.h")).read()
    declared = set(re.findall(r"\b(gr_[a-z_]+)\s*\(", hdr))
    declared -= {"gr_ctx"}
    lib = C.CDLL(G._lib.LIB_PATH)
    assert declared == set(G._lib.EXPORTS)
    for name in declared:
        assert hasattr(lib, name), name
    lib.gr_abi_version.restype = C.c_int32
    assert lib.gr_abi_version() == G._lib.ABI_VERSION == 8


def test_struct_layouts(G):
    L = G._lib
    assert L.POINT_DTYPE.itemsize == 152            # GeodesicPoint{Float64,Nothing}
    assert C.sizeof(L.gr_config) == 8 + 64 + 8 * 10 + 8 + 8 + 8 + 32 + 16 + 32 + 8 + 16 + 8 + 4 * 56 + 16      # + comp_n, comp[4]; + metric_table, metric_table_n (ABI 7)
    assert C.sizeof(L.gr_metric_segment) == 72 and C.sizeof(L.gr_metric_break) == 16          # ABI 8
    assert C.sizeof(L.gr_metric_grid) == 24 + 8 * 4 + 24 + 8 + L.GR_METRIC_MAX_SEG * 72          # ABI 8: + n_seg, n_rows, seg[]
    assert C.sizeof(L.gr_plane) == 8 * (4 + 16 + 4) + 16 + 8
    assert C.sizeof(L.gr_range) == 32
    assert C.sizeof(L.gr_stats) == 96            # ABI 6: + enqueue_ms
    assert C.sizeof(L.gr_pointfunction) == 8 + 16 + 8 + 32 + 8 + 32          # + has_u_src, u_src (ABI 7)
    assert C.sizeof(L.gr_rayset) == 8 * (4 + 16) + 8 * 4 + 8 + 8 * 5 + 8 + 24 + 16 + 8 + 8 + 16 + 8          # + sky_* (ABI 7), sky_first / sky_total / sky_rows (ABI 8)


def test_no_device_means_loud_failure(G):
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(G.GradusMI355XError) as ei:
        G.Context(0)
    assert ei.value.code == -3   # GR_ERR_NO_DEVICE: there is no CPU fallback
    with pytest.raises(G.GradusMI355XError):
        G.rendergeodesics(G.KerrMetric(), np.array([0.0, 100.0, 1.4, 0.0]), 200.0, image_width=4, image_height=4)


def test_status_codes(G):
    assert [int(s) for s in G.StatusCodes] == [0, 1, 2, 3]
    assert G.StatusCodes.IntersectedWithGeometry == 2


def test_host_observer_setup_matches_oracle(G, oracle):
    x = np.array([0.0, 1000.0, math.radians(75), 0.0])
    for m, (name, params) in (
        (G.KerrMetric(1.0, 0.998), ("kerr", (1.0, 0.998))),
        (G.KerrMetric(1.0, 0.0), ("kerr", (1.0, 0.0))),
        (G.JohannsenMetric(1.0, 0.7, 2.0, 0.0, 0.0, 1.0), ("johannsen", (1.0, 0.7, 2.0, 0.0, 0.0, 1.0))),
    ):
        cfg = oracle.make_config(name, params)
        np.testing.assert_allclose(G.lnrbasis_matrix(m, x), oracle.lnrbasis(cfg, x), rtol=1e-12, atol=1e-15)
        np.testing.assert_allclose(G.lnrframe_matrix(m, x), oracle.lnrframe(cfg, x), rtol=1e-12, atol=1e-15)
        v = G.map_impact_parameters(m, x, np.array([1.0, -3.0]), np.array([2.0, 0.5]))
        np.testing.assert_allclose(v, oracle.map_impact_parameters(cfg, x, [1.0, -3.0], [2.0, 0.5]), rtol=1e-12)
        g, _, _ = oracle.metric_jacobian(cfg, x[1], x[2])
        np.testing.assert_allclose(m.metric_components(x[1], x[2]), g, rtol=1e-13)
        assert m.inner_radius() == pytest.approx(oracle.inner_radius(params[0], params[1]))


def test_kerr_isco_matches_reference_table(G):
    assert G.KerrMetric(1.0, 0.0).isco() == 6.0
    assert G.KerrMetric(1.0, 1.0).isco() == 1.0
    assert G.KerrMetric(1.0, 0.998).isco() == pytest.approx(1.2369706551751847, abs=1e-12)
    assert G.KerrMetric(1.0, -0.998).isco() == pytest.approx(8.99437445480357, abs=1e-12)


def test_dilaton_axion_metric_and_isco(G, oracle):
    """DilatonAxion: ISCOs recorded in test/smoke-tests/special-radii.jl:15-36 (atol 1e-5 there), components
    against the oracle's dual-number restatement, and the horizon radius formula."""
    assert G.DilatonAxion(M=1.0, a=0.6, β=-0.5, b=0.1).isco() == pytest.approx(29.701502242023523, abs=1e-5)
    assert G.DilatonAxion(M=1.0, a=0.0, b=0.0, β=0.0).isco() == pytest.approx(6.0, abs=1e-5)
    p = (1.0, 0.5, 0.2, 1.0)
    m = G.DilatonAxion(*p)
    cfg = oracle.make_config("dilaton-axion", p)
    for r, th in ((7.0, 1.1), (4.2, 0.3), (30.0, 2.5)):
        np.testing.assert_allclose(m.metric_components(r, th), oracle.metric_jacobian(cfg, r, th)[0], rtol=1e-13, atol=1e-15)
    assert m.isco() == pytest.approx(oracle.isco(cfg), rel=1e-10)
    assert cfg.r_inner == pytest.approx(1.01 * m.inner_radius())
    # β = 0, b = 0 is Kerr
    k = G.KerrMetric(1.0, 0.5)
    np.testing.assert_allclose(G.DilatonAxion(1.0, 0.5, 0.0, 0.0).metric_components(5.0, 1.0), k.metric_components(5.0, 1.0),
                               rtol=1e-13)
    with pytest.raises(ValueError):
        G.DilatonAxion(M=1.0, a=0.6, β=-0.5, b=0.1).inner_radius()        # no horizon for these couplings


def test_planes(G):
    assert G.trajectory_count(G.PolarPlane(G.LinearGrid(), Nr=10, Nθ=10)) == 100
    pl = G.CartesianPlane(G.LinearGrid(), x_min=0.1, y_min=0.1, Nx=12, Ny=12)
    a, b = G.impact_parameters(pl)
    assert a.size == G.trajectory_count(pl) == 121
    assert a.min() == -150.0 and a.max() == 150.0 and (a == 0.1).sum() == 11
    g = G.GeometricGrid()(1.0, 250.0, 5)
    assert g[0] == 1.0 and g[-1] == pytest.approx(250.0)


def test_configuration_errors_mirror_reference(G):
    m = G.KerrMetric()
    x = np.array([0.0, 100.0, 1.4, 0.0])
    with pytest.raises(ValueError, match="trajectories must be defined"):
        G.tracing_configuration(m, x, lambda i: np.zeros(4), None, 200.0)
    with pytest.raises(ValueError, match="save_on"):
        G.tracing_configuration(m, x, np.zeros((2, 4)), None, 200.0, save_on=True)
    with pytest.raises(AssertionError, match="α limits must be sorted"):
        G.render_configuration(m, x, 200.0, image_width=4, image_height=4, alpha_lims=(1, -1), beta_lims=(-1, 1))
    cfg = G.tracing_configuration(m, x, np.zeros((3, 4)), G.ThinDisc(0.0, 40.0), (0.0, 200.0)).abi_config()
    assert cfg.disc_id == 1 and cfg.disc_r_out == 40.0 and cfg.gtol == 1e-2 and cfg.abstol == 1e-9
    assert cfg.r_inner == pytest.approx(2.02) and cfg.r_outer == 12000.0 and cfg.maxiters == 1_000_000


def test_pointfunction_composition(G):
    CPF = G.ConstPointFunctions
    pf = CPF.affine_time() @ CPF.filter_early_term()
    assert pf.fusable and pf.device_pf == 0 and pf.device_filter == 1
    gp = np.zeros(1, dtype=G.POINT_DTYPE)[0]
    gp["lambda_max"] = 5.0
    assert pf(None, gp, 10.0) == 5.0
    assert math.isnan(pf(None, gp, 5.0))
    custom = G.PointFunction(lambda m, gp, t, **kw: gp["x"][1]) @ CPF.filter_intersected()
    assert not custom.fusable
    assert math.isnan(custom(None, gp, 10.0))


def test_thick_disc_surface_vectors():
    """test/discs/test-geometry.jl: unit tangent of the ShakuraSunyaev surface (atol 1e-5 there)."""
    import gradus_jl_amd as G
    from gradus_jl_amd.precision_solvers import _cartesian_surface_normal, _cartesian_tangent_vector

    m = G.KerrMetric(1.0, 0.9)
    d = G.ShakuraSunyaev.for_metric(m)
    np.testing.assert_allclose(_cartesian_tangent_vector(d, 2.6), [0.689693957000099, 0.0, 0.724100991352412], atol=1e-5)
    np.testing.assert_allclose(_cartesian_tangent_vector(d, 6.6), [0.9679192396299138, 0.0, 0.2512615083021063], atol=1e-5)
    np.testing.assert_allclose(_cartesian_tangent_vector(d, 1.0), [1, 0, 0], atol=1e-5)
    n = _cartesian_surface_normal(d, 2.6)
    assert abs(float(n @ _cartesian_tangent_vector(d, 2.6))) < 1e-12 and n[2] > 0


def _build_c_client(tmp_path):
    import subprocess

    exe = str(tmp_path / "c_abi_smoke")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "c_abi_smoke.c"), "-ldl", "-lm", "-o", exe])
    return exe


def test_header_is_plain_c_and_library_refuses_without_a_device(G, tmp_path):
    """include/gradus_mi355x.h compiles as C11 (static asserts on the struct sizes included); the C client loads the
    library, checks the ABI version and -- on a host without a GPU -- gets GR_ERR_NO_DEVICE, not a CPU result."""
    import subprocess

    import torch

    exe = _build_c_client(tmp_path)
    rc = subprocess.run([exe, G._lib.LIB_PATH], capture_output=True, text=True)
    if torch.cuda.is_available():
        assert rc.returncode == 0, rc.stdout + rc.stderr
    else:
        assert rc.returncode == 3 and "no HIP device" in rc.stdout, rc.stdout + rc.stderr


def test_record_index_of_an_lds_unit():
    """points_epilogue (gr_kernels.hpp) maps unit u of a wave's 64 x 19 end-point units to its record with (u * 3450) >> 16:
    exact for every unit, and sizeof(gr_point) is 19 eight-byte units."""
    from gradus_jl_amd import _lib

    assert _lib.POINT_DTYPE.itemsize == 19 * 8
    assert all((u * 3450) >> 16 == u // 19 for u in range(64 * 19))
