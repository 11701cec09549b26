"""ONE host thread, SEVERAL devices: the *_multi entry points of include/gradus_mi355x.h (ABI 6).

The GPU box has one MI355X, so the contexts of these tests live on the same device -- which exercises everything the
call does on the host (the deal of the rays, per-context staging, the order of enqueues, copies and waits, the
kernels' stores at global indices into one pinned block) and leaves only the physical concurrency of distinct devices
untested (tests/test_gpu_parity.py::test_multi_device_render_on_distinct_devices runs where there are two).

Two properties are asserted for every entry point of the boundary (src/tracing/tracing.jl:151-196 and its callers):
  * the bytes equal those of the one-context entry point, for 1, 2 and 4 contexts, into pageable AND into
    library-pinned results;
  * the enqueue phase does not wait for a device: gr_stats.enqueue_ms of every context is a small fraction of the
    kernels' time ("What's missing" 1b: a D2H copy into pageable memory inside the enqueue loop would run the
    devices one after another).
"""
import ctypes as C
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

X_FAR = np.array([0.0, 1000.0, math.radians(75), 0.0])
ALIMS, BLIMS = (-60.0, 60.0), (-35.0, 35.0)


@pytest.fixture(scope="module")
def multi4(G):
    return G.EnsembleMI355X(devices=[0, 0, 0, 0])


def _ctxs(ens, n):
    from gradus_jl_amd import _lib

    return _lib.ctx_array(ens.contexts[:n])


def _render_config(G, ens, W, H, metric=None, disc="isco"):
    from gradus_jl_amd.rendering import render_configuration

    m = metric or G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0) if disc == "isco" else disc
    return m, render_configuration(m, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS, beta_lims=BLIMS,
                                   ensemble=ens)


def _dest(L, ctx, nbytes, pinned):
    """(array of nbytes uint8, keep-alive): pageable numpy memory or a gr_host_alloc block"""
    from gradus_jl_amd import _lib

    if pinned:
        return _lib.PinnedBlock(ctx, nbytes).array(np.uint8, nbytes)
    return np.zeros(nbytes, dtype=np.uint8)


@pytest.mark.parametrize("pinned", [False, True])
def test_fused_render_multi_equals_one_context(G, ens, multi4, pinned):
    """gr_render_multi for 1 / 2 / 4 contexts == gr_render, into pageable memory (strided copies home in phase 2) and
    into a pinned block (every kernel stores its pixels at their place in the whole image: Cold.out_global)."""
    from gradus_jl_amd import _lib
    from gradus_jl_amd.rendering import abi_pointfunction

    L = _lib.load()
    W, H = 256, 160
    m, config = _render_config(G, ens, W, H)
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    cfg, pl = config.abi_config(), config.abi_plane()
    s, keep = abi_pointfunction(pf)
    n = W * H
    ref = np.zeros(n)
    st = _lib.gr_stats()
    rg = _lib.gr_range(0, n, n, 1)
    _lib.check(L.gr_render(ens.ctx.handle, C.byref(cfg), C.byref(pl), C.byref(s), C.byref(rg), ref.ctypes.data, C.byref(st)))
    assert np.isfinite(ref).sum() > n // 10
    for k in (1, 2, 4):
        arr, sts = _ctxs(multi4, k)
        raw = _dest(L, multi4.ctx, 8 * n, pinned)
        raw[:] = 0xFF
        _lib.check(L.gr_render_multi(arr, k, C.byref(cfg), C.byref(pl), C.byref(s), 0, raw.ctypes.data, sts))
        assert raw.tobytes() == ref.tobytes(), f"{k} contexts, pinned={pinned}"
        assert sum(x.rays for x in sts) == n and all(x.rays == n // k for x in sts)
        assert all(x.enqueue_ms > 0.0 for x in sts)


@pytest.mark.parametrize("pinned", [False, True])
def test_endpoints_multi_equals_one_context(G, ens, multi4, pinned):
    """gr_render_endpoints_multi (prerendergeodesics / the generic boundary on the render closure) == gr_render_endpoints."""
    from gradus_jl_amd import _lib

    L = _lib.load()
    W, H = 256, 96
    m, config = _render_config(G, ens, W, H)
    cfg, pl = config.abi_config(), config.abi_plane()
    n = W * H
    ref = np.zeros(n, dtype=_lib.POINT_DTYPE)
    st = _lib.gr_stats()
    rg = _lib.gr_range(0, n, n, 1)
    _lib.check(L.gr_render_endpoints(ens.ctx.handle, C.byref(cfg), C.byref(pl), C.byref(rg), ref.ctypes.data, C.byref(st)))
    assert (ref["status"] == 2).sum() > n // 10
    for k in (1, 2, 4):
        arr, sts = _ctxs(multi4, k)
        raw = _dest(L, multi4.ctx, 152 * n, pinned)
        raw[:] = 0xFF
        _lib.check(L.gr_render_endpoints_multi(arr, k, C.byref(cfg), C.byref(pl), 0, raw.ctypes.data, sts))
        got = raw.view(_lib.POINT_DTYPE)
        for f in _lib.POINT_DTYPE.names:          # field by field: the 4 padding bytes' twin (flags) is a field too
            assert np.array_equal(got[f], ref[f]), (k, pinned, f)
        assert sum(x.rays for x in sts) == n
    # the Python boundary takes the same route
    pts = G.rendering.prerendergeodesics(m, X_FAR, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=W, image_height=H,
                                         alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=multi4)[2].points
    assert np.array_equal(np.ascontiguousarray(pts.T).ravel()["x"], ref["x"])


def test_trace_and_rayset_multi_equal_one_context(G, ens, multi4):
    """tracegeodesics on (x, v) arrays and on impact-parameter ray sets, and the per-ray summaries the precision solvers
    read: contiguous shares over 1 / 2 / 3 / 4 contexts give the bytes of the one-context call (n not a multiple of 64,
    so the last share is short and, with 4 contexts, shares differ)."""
    from gradus_jl_amd import _lib
    from gradus_jl_amd.rendering import abi_pointfunction
    from gradus_jl_amd.tracing import map_impact_parameters, tracing_configuration

    L = _lib.load()
    m = G.KerrMetric(1.0, 0.9)
    d = G.ThinDisc(m.isco(), 40.0)
    x = np.array([0.0, 500.0, math.radians(60), 0.0])
    rng = np.random.default_rng(11)
    n = 1000
    al, be = rng.uniform(-30, 30, n), rng.uniform(-20, 20, n)
    v = np.stack([map_impact_parameters(m, x, a, b) for a, b in zip(al, be)])
    config = tracing_configuration(m, x, v, d, (0.0, 1000.0), ensemble=ens)
    cfg = config.abi_config()
    ref = np.zeros(n, dtype=_lib.POINT_DTYPE)
    st = _lib.gr_stats()
    _lib.check(L.gr_trace_endpoints(ens.ctx.handle, C.byref(cfg), x.ctypes.data, 0, v.ctypes.data, n, ref.ctypes.data, C.byref(st)))
    assert (ref["status"] == 2).sum() > 100
    for k in (1, 2, 3, 4):
        arr, sts = _ctxs(multi4, k)
        got = np.zeros(n, dtype=_lib.POINT_DTYPE)
        _lib.check(L.gr_trace_endpoints_multi(arr, k, C.byref(cfg), x.ctypes.data, 0, v.ctypes.data, n, got.ctypes.data, sts))
        for f in _lib.POINT_DTYPE.names:
            assert np.array_equal(got[f], ref[f]), (k, f)
        assert sum(s.rays for s in sts) == n
    # per-ray positions (x_stride = 4)
    xs = np.repeat(x[None, :], n, axis=0).copy()
    arr, sts = _ctxs(multi4, 3)
    got = np.zeros(n, dtype=_lib.POINT_DTYPE)
    _lib.check(L.gr_trace_endpoints_multi(arr, 3, C.byref(cfg), xs.ctypes.data, 4, v.ctypes.data, n, got.ctypes.data, sts))
    assert np.array_equal(got["x"], ref["x"]) and np.array_equal(got["status"], ref["status"])
    # the Python boundary
    got = G.tracegeodesics(m, x, v, d, (0.0, 1000.0), ensemble=multi4)
    assert np.array_equal(got["v"], ref["v"])

    # ---- impact-parameter ray sets
    from gradus_jl_amd.tracing import lnr_momentum_to_global_velocity_matrix

    rs = _lib.gr_rayset()
    Mx = lnr_momentum_to_global_velocity_matrix(m, x)
    for i in range(4):
        rs.x_obs[i] = float(x[i])
        for q in range(4):
            rs.Mx[4 * i + q] = float(Mx[i, q])
    al, be = np.ascontiguousarray(al), np.ascontiguousarray(be)
    rs.alpha, rs.beta, rs.n = al.ctypes.data, be.ctypes.data, n
    pf, keep = abi_pointfunction(G.ConstPointFunctions.redshift(m, x))
    ref_pts = np.zeros(n, dtype=_lib.POINT_DTYPE)
    _lib.check(L.gr_rayset_endpoints(ens.ctx.handle, C.byref(cfg), C.byref(rs), ref_pts.ctypes.data, C.byref(st)))
    ref_sum = np.zeros((n, 4))
    _lib.check(L.gr_ray_summary(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), ref_sum.ctypes.data, C.byref(st)))
    ref_tan = np.zeros((n, 8))
    _lib.check(L.gr_ray_tangent(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), ref_tan.ctypes.data, C.byref(st)))
    ref_gr = np.zeros((n, 2))
    _lib.check(L.gr_redshift_radius(ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), m.isco(), 40.0, ref_gr.ctypes.data,
                                    C.byref(st)))
    for k in (2, 4):
        arr, sts = _ctxs(multi4, k)
        got = np.zeros(n, dtype=_lib.POINT_DTYPE)
        _lib.check(L.gr_rayset_endpoints_multi(arr, k, C.byref(cfg), C.byref(rs), got.ctypes.data, sts))
        assert np.array_equal(got["x"], ref_pts["x"]) and np.array_equal(got["lambda_max"], ref_pts["lambda_max"])
        o = np.zeros((n, 4))
        _lib.check(L.gr_ray_summary_multi(arr, k, C.byref(cfg), C.byref(rs), C.byref(pf), o.ctypes.data, sts))
        assert o.tobytes() == ref_sum.tobytes()
        o = np.zeros((n, 8))
        _lib.check(L.gr_ray_tangent_multi(arr, k, C.byref(cfg), C.byref(rs), C.byref(pf), o.ctypes.data, sts))
        assert o.tobytes() == ref_tan.tobytes()
        o = np.zeros((n, 2))
        _lib.check(L.gr_redshift_radius_multi(arr, k, C.byref(cfg), C.byref(rs), C.byref(pf), m.isco(), 40.0, o.ctypes.data, sts))
        assert o.tobytes() == ref_gr.tobytes()
    # a separable plane in ray order (tracegeodesics(m, x, plane, ...)): shares are ranges of sep_first
    plane = G.PolarPlane(G.GeometricGrid(), Nr=37, Nθ=29, r_min=1.0, r_max=40.0)
    one = G.tracegeodesics(m, x, plane, d, (0.0, 1000.0), ensemble=ens)
    many = G.tracegeodesics(m, x, plane, d, (0.0, 1000.0), ensemble=multi4)
    assert one.size == 37 * 29 and np.array_equal(one["x"], many["x"]) and np.array_equal(one["status"], many["status"])


def test_lineprofile_multi_equals_one_context(G, ens, multi4):
    """gr_lineprofile_multi: the rays of a PolarPlane dealt block-cyclically (strips of 8 x 8 tiles), one histogram per
    context, added on the host.  Every ray is binned exactly once (the counters add up) and the histogram equals the
    one-context one to rounding (the order in which a bin receives its rays differs, as between any two launches)."""
    from gradus_jl_amd import _lib

    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 120.0)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    bins = np.linspace(0.1, 1.5, 180)
    for (nr, nt) in ((256, 256), (100, 61)):          # whole tiles / ragged edges
        plane = G.PolarPlane(G.GeometricGrid(), Nr=nr, Nθ=nt, r_min=1.0, r_max=120.0)
        kw = dict(plane=plane, maxrₑ=100.0, stats=True)
        _, f1, s1 = G.lineprofile(bins, G.PowerLawEmissivity(3.0), m, u, d, G.BinningMethod(), ensemble=ens, **kw)
        _, f4, s4 = G.lineprofile(bins, G.PowerLawEmissivity(3.0), m, u, d, G.BinningMethod(), ensemble=multi4, **kw)
        assert s4["rays"] == s1["rays"] == nr * nt
        assert s4["status_count"] == s1["status_count"] and s4["accepted_steps"] == s1["accepted_steps"]
        assert f1.sum() == pytest.approx(1.0, abs=1e-12)
        np.testing.assert_allclose(f4, f1, rtol=1e-11, atol=1e-15)
    # ray arrays instead of a separable plane: contiguous shares
    import os

    os.environ["GRADUS_MI355X_SEPARABLE_RAYS"] = "0"
    try:
        _, g1 = G.lineprofile(bins, G.PowerLawEmissivity(3.0), m, u, d, G.BinningMethod(), ensemble=ens, plane=plane, maxrₑ=100.0)
        _, g4 = G.lineprofile(bins, G.PowerLawEmissivity(3.0), m, u, d, G.BinningMethod(), ensemble=multi4, plane=plane, maxrₑ=100.0)
    finally:
        del os.environ["GRADUS_MI355X_SEPARABLE_RAYS"]
    np.testing.assert_allclose(g4, g1, rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(g1, f1, rtol=1e-11, atol=1e-15)


def test_transfer_function_through_a_multi_ensemble(G, ens, multi4):
    """The Cunningham transfer function's launches (gr_ray_tangent / gr_ray_summary on a few hundred rays each) through an
    ensemble of four contexts: the same numbers as on one (contiguous shares, same kernels, same bytes per ray)."""
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 100_000.0, math.radians(60), 0.0])
    kw = dict(N=40, chart=G.chart_for_metric(m, 2 * x[1], closest_approach=1.005))
    one = G.cunningham_transfer_function(m, x, G.ThinDisc(0.0, float("inf")), 7.0, ensemble=ens, **kw)
    many = G.cunningham_transfer_function(m, x, G.ThinDisc(0.0, float("inf")), 7.0, ensemble=multi4, **kw)
    assert one.f.size == many.f.size and np.all(np.isfinite(one.f))
    assert many.f.tobytes() == one.f.tobytes() and many.g_star.tobytes() == one.g_star.tobytes()


def test_multi_argument_errors(G, ens, multi4):
    from gradus_jl_amd import _lib
    from gradus_jl_amd.tracing import separable_rayset, tracing_configuration

    L = _lib.load()
    m, config = _render_config(G, ens, 64, 64)
    cfg, pl = config.abi_config(), config.abi_plane()
    h = multi4.contexts[0].handle
    twice = (C.c_void_p * 2)(h, h)
    out = np.zeros(64 * 64, dtype=_lib.POINT_DTYPE)
    assert L.gr_render_endpoints_multi(twice, 2, C.byref(cfg), C.byref(pl), 0, out.ctypes.data, None) == -1
    assert b"twice" in L.gr_last_error()
    arr, sts = _ctxs(multi4, 3)
    assert L.gr_render_endpoints_multi(arr, 3, C.byref(cfg), C.byref(pl), 0, out.ctypes.data, sts) == -1      # 64 columns over 3
    # per-ray outputs of a TILED separable set have no ray order to share out
    plane = G.PolarPlane(G.GeometricGrid(), Nr=16, Nθ=16, r_min=1.0, r_max=40.0)
    rs, keep = separable_rayset(m, X_FAR, plane, tiled=True)
    pts = np.zeros(256, dtype=_lib.POINT_DTYPE)
    arr, sts = _ctxs(multi4, 2)
    assert L.gr_rayset_endpoints_multi(arr, 2, C.byref(cfg), C.byref(rs), pts.ctypes.data, sts) == -1
    assert b"ray order" in L.gr_last_error()
    # nothing was left in flight: the contexts still work
    rs, keep = separable_rayset(m, X_FAR, plane, tiled=False)
    assert L.gr_rayset_endpoints_multi(arr, 2, C.byref(cfg), C.byref(rs), pts.ctypes.data, sts) == 0
    assert sum(s.rays for s in sts) == 256


@pytest.mark.parametrize("pinned", [False, True])
def test_enqueue_phase_does_not_wait_for_a_device(G, ens, multi4, pinned):
    """VERDICT r3 1(b).  Four contexts, a 2048² fused render and a 1024² end-point render: the host time each context's
    enqueue took is a small fraction of the kernels' time, into pageable memory (whose copies are queued only after
    EVERY kernel has been launched) and into pinned memory (no copies at all).  Were a blocking copy part of the
    enqueue loop, enqueue_ms of context 0 would be about its kernel's duration."""
    from gradus_jl_amd import _lib
    from gradus_jl_amd.rendering import abi_pointfunction

    L = _lib.load()
    k = 4
    arr, sts = _ctxs(multi4, k)
    report = {}
    for what, (W, H) in (("image", (2048, 2048)), ("endpoints", (1024, 1024))):
        m, config = _render_config(G, ens, W, H)
        cfg, pl = config.abi_config(), config.abi_plane()
        n = W * H
        pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
        s, keep = abi_pointfunction(pf)
        raw = _dest(L, multi4.ctx, (8 if what == "image" else 152) * n, pinned)
        for rep in range(2):          # the first call allocates scratch and loads code objects
            if what == "image":
                _lib.check(L.gr_render_multi(arr, k, C.byref(cfg), C.byref(pl), C.byref(s), 0, raw.ctypes.data, sts))
            else:
                _lib.check(L.gr_render_endpoints_multi(arr, k, C.byref(cfg), C.byref(pl), 0, raw.ctypes.data, sts))
        enq = [x.enqueue_ms for x in sts]
        ker = [x.kernel_ms for x in sts]
        call = [x.call_ms for x in sts]
        report[what] = (enq, ker, call)
        print(f"  {what} pinned={pinned}: enqueue_ms {['%.3f' % e for e in enq]} kernel_ms {['%.2f' % q for q in ker]} "
              f"call_ms {['%.2f' % q for q in call]}")
        assert max(enq) < 0.1 * max(ker), (what, enq, ker)
        assert sum(enq) < 0.25 * max(ker), (what, enq, ker)
        assert sum(x.rays for x in sts) == n


def test_image_width_not_divisible_by_the_number_of_devices(G, ens):
    """three contexts and a 64-wide image -- the library deals whole columns and refuses 64 / 3; the host side takes
    the largest leading subset of the ensemble that divides the width (here two) instead of raising, for the fused render
    and for the end points, and the pixels / records are those of one context."""
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    three = G.EnsembleMI355X(devices=[0, 0, 0])
    assert len(three.contexts_for_width(64)) == 2 and len(three.contexts_for_width(63)) == 3 and len(three.contexts_for_width(61)) == 1
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=64, image_height=64, alpha_lims=ALIMS, beta_lims=BLIMS)
    _, _, one = G.rendergeodesics(m, X_FAR, d, 2000.0, pf=pf, ensemble=ens, **kw)
    _, _, many = G.rendergeodesics(m, X_FAR, d, 2000.0, pf=pf, ensemble=three, **kw)
    assert one.tobytes() == many.tobytes()
    _, _, c1 = G.prerendergeodesics(m, X_FAR, d, 2000.0, ensemble=ens, **kw)
    _, _, c3 = G.prerendergeodesics(m, X_FAR, d, 2000.0, ensemble=three, **kw)
    assert np.asarray(c1.points).tobytes() == np.asarray(c3.points).tobytes()


def test_tabulated_metric_through_a_multi_ensemble(G, ens, multi4):
    """A user-defined metric (GR_METRIC_TABULATED) over several contexts: every context stages its own copy of the table; the
    fused image, the end points of a ray array and a corona's sky rays equal the one-context results bit for bit."""
    tab = G.TabulatedMetric(G.KerrMetric(1.0, 0.9))
    d = G.ThinDisc(tab.isco(), 50.0)
    pf = G.ConstPointFunctions.redshift(tab, X_FAR, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=128, image_height=96, alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf)
    _, _, ref = G.rendergeodesics(tab, X_FAR, d, 2000.0, ensemble=ens, **kw)
    _, _, img = G.rendergeodesics(tab, X_FAR, d, 2000.0, ensemble=multi4, **kw)
    assert np.isfinite(ref).sum() > 500 and img.tobytes() == ref.tobytes()
    rng = np.random.default_rng(3)
    vs = G.map_impact_parameters(tab, X_FAR, rng.uniform(-20, 20, 3001), rng.uniform(-12, 12, 3001))
    a = G.tracegeodesics(tab, X_FAR, vs, d, 2000.0, ensemble=ens)
    b = G.tracegeodesics(tab, X_FAR, vs, d, 2000.0, ensemble=multi4)
    assert a.tobytes() == b.tobytes()
    # a corona's sky source is one context's work: a multi ensemble uses its first context
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    p1 = G.emissivity_profile(tab, G.ThinDisc(0.0, 200.0), G.LampPostModel(h=8.0), sampler=s, n_samples=4000, N=20, ensemble=ens)
    p4 = G.emissivity_profile(tab, G.ThinDisc(0.0, 200.0), G.LampPostModel(h=8.0), sampler=s, n_samples=4000, N=20, ensemble=multi4)
    np.testing.assert_array_equal(p1.ε, p4.ε)
