"""Parity of the HIP path (through the C ABI) with the CPU oracle.  Needs an MI355X."""
import math

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

X_SMOKE = np.array([0.0, 100.0, math.radians(85), 0.0])
X_FAR = np.array([0.0, 1000.0, math.radians(75), 0.0])
ALIMS, BLIMS = (-60.0, 60.0), (-35.0, 35.0)

# fp64 tolerance of the north star: redshift map rtol 1e-6 (integration tolerance 1e-9)
RTOL = 1e-6


def _metric(G, name, params):
    return {"kerr": G.KerrMetric, "johannsen": G.JohannsenMetric, "morris-thorne": G.MorrisThorneWormhole,
            "bumblebee": G.BumblebeeMetric, "kerr-newman": G.KerrNewmanMetric,
            "johannsen-psaltis": G.JohannsenPsaltisMetric, "dilaton-axion": G.DilatonAxion}[name](*params)


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize(
    "name,params,disc,expected",
    [
        ("kerr", (1.0, 0.0), None, 9009.452876609641),
        ("johannsen", (1.0, 0.0, 0.0, 0.0, 0.0, 0.0), None, 9009.448935932085),
        ("kerr", (1.0, 0.0), (0.0, 40.0), 38412.08347901267),
        ("johannsen", (1.0, 0.0, 0.0, 0.0, 0.0, 0.0), (0.0, 40.0), 38412.08386562321),
        ("bumblebee", (1.0, 0.0, 0.0), None, 9009.452384885506),
        ("kerr-newman", (1.0, 0.0, 0.0), None, 9009.451384824908),
        ("morris-thorne", (1.0,), (0.0, 40.0), 9375.430228131403),
        ("bumblebee", (1.0, 0.0, 0.0), (0.0, 40.0), 38412.0832157869),
        ("kerr-newman", (1.0, 0.0, 0.0), (0.0, 40.0), 38412.08517225652),
    ],
)
def test_reference_fingerprints_on_device(G, ens, kernel, name, params, disc, expected):
    """test/smoke-tests/rendergeodesics.jl:43-67 run through the HIP path."""
    ens.set("kernel", kernel)
    m = _metric(G, name, params)
    args = (G.ThinDisc(*disc), 200.0) if disc else (200.0,)
    _, _, img = G.rendergeodesics(m, X_SMOKE, *args, image_width=20, image_height=20, alpha_lims=(-9.5, 9.5),
                                  beta_lims=(-9.5, 9.5), ensemble=ens)
    # Morris-Thorne + thin disc: a few huge steps in a nearly flat exterior; the oracle's own fingerprint moves by
    # 1.7e-6 / 8.9e-6 when its tolerance is nudged by -/+10 % (tests/test_kernel_logic_host.py), so 1e-5 there
    assert float(np.nansum(img)) == pytest.approx(expected, rel=1e-5 if (name == "morris-thorne" and disc) else 1e-6)


def _compare_points(G, O, got, ref, rtol=RTOL, median=1e-11):
    """Endpoint parity.  Status must agree except for a few edge pixels (rays grazing the disc rim
    or a thin higher-order image; chaotic near the photon orbit).  Rays that reach the chart,
    the disc or λ_max must agree to `rtol`.  Rays swallowed by the hole stop at whichever step
    first lands inside 1.01 r₊, where v^t and t diverge, so only λ and r are meaningful there."""
    mism = got["status"] != ref["status"]
    assert mism.sum() <= max(2, got.size // 500), f"{mism.sum()} status mismatches"
    ok = ~mism
    np.testing.assert_array_equal(got["flags"][ok], 0)
    np.testing.assert_allclose(got["x_init"][ok], ref["x_init"][ok], rtol=1e-12, atol=1e-15)
    np.testing.assert_allclose(got["v_init"][ok], ref["v_init"][ok], rtol=1e-11, atol=1e-15)
    inner = ok & (ref["status"] == O.WITHIN_INNER_BOUNDARY)
    rest = ok & ~inner
    np.testing.assert_allclose(got["lambda_max"][rest], ref["lambda_max"][rest], rtol=rtol)
    errs = []
    for f in ("x", "v"):
        scale = np.maximum(np.abs(ref[f][rest]), 1.0)
        e = np.abs(got[f][rest] - ref[f][rest]) / scale
        assert e.max() < rtol, (f, e.max())
        errs.append(e)
    # typical agreement is far below the tolerance: rounding-level differences only
    assert np.median(np.concatenate(errs, axis=1)) < median
    if inner.any():
        # one step more or less when a step ends within rounding of 1.01 r₊ (steps there are ~1e-3)
        np.testing.assert_allclose(got["lambda_max"][inner], ref["lambda_max"][inner], rtol=1e-4)
        np.testing.assert_allclose(got["x"][inner, 1], ref["x"][inner, 1], rtol=1e-2)


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("disc", [None, "isco"])
def test_endpoints_match_oracle_64(G, oracle, ens, kernel, disc):
    ens.set("kernel", kernel)
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0) if disc else None
    args = (d, 2000.0) if d else (2000.0,)
    W = H = 64
    _, _, cache = G.prerendergeodesics(m, X_FAR, *args, image_width=W, image_height=H, alpha_lims=ALIMS,
                                       beta_lims=BLIMS, ensemble=ens)
    got = np.ascontiguousarray(cache.points.T).ravel()
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc=(m.isco(), 50.0) if disc else None, lambda_max=2000.0)
    ref = oracle.trace(cfg, X_FAR, oracle.render_velocities(cfg, X_FAR, ALIMS, BLIMS, W, H))
    _compare_points(G, oracle, got, ref)


@pytest.mark.parametrize("kernel", [0, 1])
def test_redshift_image_matches_oracle_128(G, oracle, ens, kernel):
    """BASELINE config C2 geometry at 128x128: fused redshift ∘ filter_intersected."""
    ens.set("kernel", kernel)
    m = G.KerrMetric(1.0, 0.998)
    isco = m.isco()
    W = H = 128
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    _, _, img, st = G.rendergeodesics(m, X_FAR, G.ThinDisc(isco, 50.0), 2000.0, image_width=W, image_height=H,
                                      alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf, ensemble=ens, stats=True)
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc=(isco, 50.0), lambda_max=2000.0)
    ref = oracle.rendergeodesics(cfg, X_FAR, ALIMS, BLIMS, W, H, pf_id=oracle.PF_REDSHIFT,
                                 filter_id=oracle.FILTER_INTERSECTED, r_isco=isco)
    assert st["rays"] == W * H and st["flagged_rays"] == 0
    nan_mismatch = np.isnan(img) != np.isnan(ref)
    # rim pixels only: the 8-point event sampling is step-sequence dependent
    assert nan_mismatch.sum() <= 0.002 * W * H
    both = ~np.isnan(img) & ~np.isnan(ref)
    assert both.sum() > 1000
    np.testing.assert_allclose(img[both], ref[both], rtol=RTOL)


def test_reference_pointfunction_smoke_scene(G, oracle, ens):
    """test/smoke-tests/pointfunctions.jl: KerrMetric() (M = 1, a = 0), observer at r = 100, 85°, ThinDisc(10, 40),
    100 x 100 pixels over ±50, unfiltered redshift -- the reference only asserts that it runs; here every pixel against
    the oracle."""
    m = G.KerrMetric(1.0, 0.0)
    u = np.array([0.0, 100.0, math.radians(85), 0.0])
    pf = G.ConstPointFunctions.redshift(m, u)
    _, _, img, st = G.rendergeodesics(m, u, G.ThinDisc(10.0, 40.0), 200.0, image_width=100, image_height=100,
                                      alpha_lims=(-50, 50), beta_lims=(-50, 50), pf=pf, ensemble=ens, stats=True)
    cfg = oracle.make_config("kerr", (1.0, 0.0), disc=(10.0, 40.0), lambda_max=200.0)
    ref = oracle.rendergeodesics(cfg, u, (-50, 50), (-50, 50), 100, 100, pf_id=oracle.PF_REDSHIFT,
                                 filter_id=oracle.FILTER_NONE, r_isco=m.isco())
    assert st["rays"] == 10_000 and st["flagged_rays"] == 0
    assert (np.isnan(img) != np.isnan(ref)).sum() <= 20
    both = ~np.isnan(img) & ~np.isnan(ref)
    assert both.sum() > 1000
    # unfiltered: every ray carries a value, also the 50 that end at the horizon (the redshift formula evaluated there is
    # ill-conditioned) and the pixel row that grazes the disc edge-on (g passes through 0): 51 of the 10 000 pixels are not
    # 1e-6-close between the host build of the kernels and the oracle; everything else is
    rel = np.abs(img[both] / ref[both] - 1.0)
    assert (rel > RTOL).sum() <= 80 and np.median(rel) < 1e-9


def test_fused_image_equals_endpoints_plus_apply(G, ens):
    """rendergeodesics(pf) == apply(pf, prerendergeodesics(...)) (test/smoke-tests/prerendergeodesics.jl:33-42)."""
    ens.set("kernel", 1)
    m = G.KerrMetric(1.0, 0.9)
    d = G.ThinDisc(0.0, 40.0)
    kw = dict(image_width=48, image_height=40, alpha_lims=(-20, 20), beta_lims=(-15, 15), ensemble=ens)
    for pf in (G.ConstPointFunctions.shadow(),
               G.ConstPointFunctions.redshift(m, X_SMOKE) @ G.ConstPointFunctions.filter_intersected()):
        _, _, img = G.rendergeodesics(m, X_SMOKE, d, 200.0, pf=pf, **kw)
        _, _, cache = G.prerendergeodesics(m, X_SMOKE, d, 200.0, **kw)
        img2 = G.apply(pf, cache)
        np.testing.assert_array_equal(np.isnan(img), np.isnan(img2))
        ok = ~np.isnan(img)
        np.testing.assert_allclose(img[ok], img2[ok], rtol=1e-12)


def test_endpoints_returned_in_bands_equal_one_copy(G, ens):
    """gr_render_endpoints on a large plane: traced and copied back in bands of whole 8-column strips on two streams
    (later bands computing while earlier ones travel), the destination pre-faulted by helper threads -- byte for byte
    the records of the single launch + single copy, also for a width that leaves a ragged last band, and the statistics
    add up."""
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    for W, H in ((1536, 1536), (1100, 2048)):
        cfg = G.render_configuration(m, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=ens)
        out = {}
        for pipe in (4, 7, 0):
            ens.set("pipeline", pipe)
            pts, st = G.ensemble_solve_tracing_problem(ens, cfg, stats=True)
            out[pipe] = (pts.copy(), st)
        for pipe in (4, 7):
            assert out[pipe][0].tobytes() == out[0][0].tobytes()
            for k in ("rays", "accepted_steps", "rejected_steps", "status_count"):
                assert out[pipe][1][k] == out[0][1][k]
        assert out[4][1]["rays"] == W * H


def test_endpoints_into_a_block_the_library_pinned(G, ens, monkeypatch):
    """gr_host_alloc / gr_host_free (ABI 5; what the Julia shim wraps as its Vector{GeodesicPoint}): the end points of a
    2048² plane (637 MB) returned into page-locked memory are byte for byte those of the pageable call, the copy hides
    under the trace of the later bands (device time of the whole call within 10 % of its kernels; the pageable path pays
    ~38 %), kernel_ms and call_ms of gr_stats tell the two apart, and blocks outlive their context."""
    import ctypes as C

    from gradus_jl_amd import _lib

    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    W = H = 2048
    cfg = G.render_configuration(m, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=ens)
    ens.set("pipeline", 4)
    res = {}
    for mode in ("1", "0", "1", "0"):                      # second round: warm buffers on both paths
        monkeypatch.setenv("GRADUS_MI355X_PINNED_RESULTS", mode)
        pts, st = G.ensemble_solve_tracing_problem(ens, cfg, stats=True)
        res[mode] = (pts, st)
    pinned, pageable = res["1"], res["0"]
    assert isinstance(pinned[0].base, C.Array) or pinned[0].base is not None
    assert pinned[0].tobytes() == pageable[0].tobytes()
    for st in (pinned[1], pageable[1]):
        assert 0.0 < st["kernel_ms"] <= st["call_ms"]
    print(f"  2048² end points: pinned kernel {pinned[1]['kernel_ms']:.2f} call {pinned[1]['call_ms']:.2f} ms; "
          f"pageable kernel {pageable[1]['kernel_ms']:.2f} call {pageable[1]['call_ms']:.2f} ms")
    assert pinned[1]["call_ms"] < 1.10 * pinned[1]["kernel_ms"]
    assert pinned[1]["call_ms"] < pageable[1]["call_ms"]
    # a block survives the context that allocated it and is freed without one
    ctx2 = _lib.Context(0)
    blk = _lib.PinnedBlock(ctx2, 1 << 20)
    a = blk.array(np.float64, 1 << 17)
    a[:] = 3.0
    ctx2.close()
    assert float(a.sum()) == 3.0 * (1 << 17)
    del a, blk
    L = _lib.load()
    assert L.gr_host_free(None, None) == 0
    assert L.gr_host_free(None, C.c_void_p(0x1000)) == -1          # not one of ours: refused, nothing freed


def test_freed_pinned_blocks_are_reused(G, ens):
    """Page-locking costs ten times the call it serves (608 MiB: 113-365 ms to lock, 75 ms to unlock), so gr_host_free parks
    the block in a process-wide pool and the next gr_host_alloc of a similar size takes it from there: same address, no
    page-locking; a much smaller request does not take a large block; "pinned_pool_mib" 0 empties and disables the pool."""
    import time

    from gradus_jl_amd import _lib

    ens.set("pinned_pool_mib", 0)                            # whatever earlier tests left behind goes
    ens.set("pinned_pool_mib", 4096)
    big = 96 << 20
    a = _lib.PinnedBlock(ens.ctx, big)
    pa = a.ptr
    del a
    t0 = time.perf_counter()
    b = _lib.PinnedBlock(ens.ctx, big - (8 << 20))           # within [size / 2, size]: served from the pool
    dt = time.perf_counter() - t0
    assert b.ptr == pa and dt < 2e-3, dt
    v = b.array(np.float64, 1024)
    v[:] = 2.0                                               # still writable host memory
    assert float(v.sum()) == 2048.0
    c = _lib.PinnedBlock(ens.ctx, 1 << 20)                   # far smaller: its own block
    assert c.ptr != pa
    del v, b, c
    ens.set("pinned_pool_mib", 0)                            # pool emptied, nothing is kept any more
    d = _lib.PinnedBlock(ens.ctx, 1 << 20)
    pd = d.ptr
    del d
    e = _lib.PinnedBlock(ens.ctx, 32 << 20)
    f = _lib.PinnedBlock(ens.ctx, 1 << 20)                   # were the 1 MiB block still pooled it would come back here
    assert f.ptr != pd or True                               # (the runtime may reuse the address; what counts: no error, memory usable)
    f.array(np.float64, 16)[:] = 1.0
    del e, f
    # blocks of 8 MiB and more are mappings on transparent huge pages registered with the runtime; "pinned_huge" 0 takes
    # every block from hipHostMalloc instead -- both are ordinary host memory the kernels can store into
    for huge in (0, 1):
        ens.set("pinned_huge", huge)
        g = _lib.PinnedBlock(ens.ctx, 24 << 20)
        w = g.array(np.float64, (24 << 20) // 8)
        w[:] = 3.0
        assert float(w[::4096].sum()) == 3.0 * len(w[::4096])
        del w, g
    ens.set("pinned_pool_mib", 4096)


def test_image_into_a_block_the_library_pinned(G, ens):
    """gr_render into an image allocated with gr_host_alloc: the kernel stores the pixels across the link itself (no
    staging image, no copy; "direct_host") -- the bytes of the staged call, for a whole plane and for a range of it."""
    import ctypes as C

    from gradus_jl_amd import _lib
    from gradus_jl_amd.rendering import abi_pointfunction

    m = G.KerrMetric(1.0, 0.9)
    x = np.array([0.0, 800.0, math.radians(70), 0.0])
    W, H = 264, 200
    cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 40.0), 1600.0, image_width=W, image_height=H,
                                 alpha_lims=(-30, 30), beta_lims=(-20, 20), ensemble=ens)
    pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
    acfg, apl = cfg.abi_config(), cfg.abi_plane()
    apf, keep = abi_pointfunction(pf)
    L = _lib.load()
    for first, count in ((0, W * H), (8 * H, 40 * H)):
        rg = _lib.gr_range(first, count, count, 1)
        plain = np.full(count, -7.0)
        blk = _lib.PinnedBlock(ens.ctx, 8 * count)
        pinned = blk.array(np.float64, count)
        pinned[:] = -7.0
        st = _lib.gr_stats()
        _lib.check(L.gr_render(ens.ctx.handle, C.byref(acfg), C.byref(apl), C.byref(apf), C.byref(rg), plain.ctypes.data, C.byref(st)))
        _lib.check(L.gr_render(ens.ctx.handle, C.byref(acfg), C.byref(apl), C.byref(apf), C.byref(rg), pinned.ctypes.data, C.byref(st)))
        assert st.rays == count
        assert pinned.tobytes() == plain.tobytes()
        assert np.isfinite(plain).sum() > 0.05 * count
        del pinned, blk


def _render_points(G, ens, cfg, first=0, count=None, into=None):
    """gr_render_endpoints on a range of the plane, as records (optionally into a caller-supplied array)"""
    import ctypes as C

    from gradus_jl_amd import _lib

    acfg, pl = cfg.abi_config(), cfg.abi_plane()
    n = pl.width * pl.height
    count = n - first if count is None else count
    rg = _lib.gr_range(first, count, max(count, 1), 1)
    pts = np.zeros(count, dtype=_lib.POINT_DTYPE) if into is None else into
    st = _lib.gr_stats()
    _lib.check(_lib.load().gr_render_endpoints(ens.ctx.handle, C.byref(acfg), C.byref(pl), C.byref(rg), pts.ctypes.data, C.byref(st)))
    return pts, st


def _trace_points(G, ens, cfg, xs, vs):
    """array inputs through gr_trace_endpoints, as records"""
    import ctypes as C

    from gradus_jl_amd import _lib

    acfg = cfg.abi_config()
    n = vs.shape[0]
    pts = np.zeros(n, dtype=_lib.POINT_DTYPE)
    xs, vs = np.ascontiguousarray(xs), np.ascontiguousarray(vs)
    _lib.check(_lib.load().gr_trace_endpoints(ens.ctx.handle, C.byref(acfg), xs.ctypes.data, 4, vs.ctypes.data, n, pts.ctypes.data, None))
    return pts


def test_endpoint_records_sent_by_the_wave_equal_per_lane_stores(G, ens):
    """End-point records of the one-ray-per-lane kernel leave through LDS, a wave's 64 records as runs of consecutive
    addresses (gr_kernels.hpp points_epilogue; gr_ctx_set "lds_points", default 1) -- byte for byte what each lane storing its
    own 152 bytes produces: tiled planes, planes whose height is no multiple of 8, a ragged last wave, a range that starts
    inside the plane, array inputs; and into a block the library pinned the kernel's own stores across the link
    ("direct_host", default 1) equal the staged, banded copy."""
    from gradus_jl_amd import _lib

    m = G.KerrMetric(1.0, 0.9)
    d = G.ThinDisc(m.isco(), 40.0)
    x = np.array([0.0, 800.0, math.radians(70), 0.0])
    kw = dict(alpha_lims=(-30, 30), beta_lims=(-20, 20), ensemble=ens)

    def both(fn):
        out = []
        for v in (1, 0):
            ens.set("lds_points", v)
            out.append(fn())
        ens.set("lds_points", 1)
        assert out[0].tobytes() == out[1].tobytes()
        return out[0]

    for (W, H) in ((96, 80), (37, 29), (64, 1), (1, 200)):
        cfg = G.render_configuration(m, x, d, 1600.0, image_width=W, image_height=H, **kw)
        pts = both(lambda: _render_points(G, ens, cfg)[0])
        assert pts.size == W * H and (pts["status"] > 0).all()
        n0 = (W * H) // 3
        sub = both(lambda: _render_points(G, ens, cfg, first=n0)[0])
        assert sub.tobytes() == pts[n0:].tobytes()
    # array inputs, n not a multiple of 64: the directions of a 40 x 25 plane, ray by ray
    cfga = G.render_configuration(m, x, d, 1600.0, image_width=40, image_height=25, **kw)
    ref = _render_points(G, ens, cfga)[0]
    got = both(lambda: _trace_points(G, ens, cfga, ref["x_init"], ref["v_init"]))
    assert got.size == 1000
    np.testing.assert_array_equal(got["status"], ref["status"])
    np.testing.assert_allclose(got["x"], ref["x"], rtol=1e-9, atol=1e-9)
    # pinned destination: the kernel's own stores across the link against staged bands
    Wb = Hb = 1536
    cfgb = G.render_configuration(m, x, d, 1600.0, image_width=Wb, image_height=Hb, **kw)
    nb = Wb * Hb
    res = []
    for v in (1, 0):
        ens.set("direct_host", v)
        blk = _lib.PinnedBlock(ens.ctx, nb * 152)
        a, st = _render_points(G, ens, cfgb, into=blk.array(_lib.POINT_DTYPE, nb))
        res.append((blk, a, st.kernel_ms, st.call_ms))
    ens.set("direct_host", 1)
    assert res[0][1].tobytes() == res[1][1].tobytes()
    print(f"  {Wb}² end points into pinned memory: direct kernel {res[0][2]:.2f} call {res[0][3]:.2f} ms; "
          f"banded kernel {res[1][2]:.2f} call {res[1][3]:.2f} ms")


def test_endpoint_cache_kept_on_the_device(G, ens):
    """prerendergeodesics(..., keep_on_device=True): `apply` of built-in point functions runs on the records in HBM
    (gr_apply_pointfunction_device) and gives the image of the host route bit for bit; the host copy appears on first use
    and equals the ordinary cache; a Python point function still works (it reads the host copy)."""
    import time

    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    kw = dict(image_width=512, image_height=384, alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=ens)
    _, _, host = G.prerendergeodesics(m, X_FAR, d, 2000.0, **kw)
    _, _, dev = G.prerendergeodesics(m, X_FAR, d, 2000.0, keep_on_device=True, **kw)
    assert dev._points is None and dev.device_points is not None
    for pf in (G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected(), G.ConstPointFunctions.shadow(),
               G.ConstPointFunctions.affine_time()):
        a, b = G.apply(pf, dev), G.apply(pf, host)
        assert a.shape == b.shape == (384, 512) and a.tobytes() == b.tobytes()
    assert dev._points is None                                   # nothing came back but images
    assert dev.points.tobytes() == host.points.tobytes()
    custom = G.PointFunction(lambda mm, gp, t: gp["x"][1])
    np.testing.assert_array_equal(G.apply(custom, dev), G.apply(custom, host))
    t0 = time.perf_counter()
    G.apply(G.ConstPointFunctions.shadow(), dev)
    t1 = time.perf_counter()
    G.apply(G.ConstPointFunctions.shadow(), host)
    t2 = time.perf_counter()
    print(f"apply(shadow) on a 512x384 cache: {1e3 * (t1 - t0):.2f} ms in HBM, {1e3 * (t2 - t1):.2f} ms from the host")


def test_kernels_agree_bitwise(G, ens):
    """Wave-ballot refill must not change any ray: persistent == one-ray-per-lane, bit for bit."""
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    out = []
    for kernel, thr in ((0, 8), (1, 1), (1, 8), (1, 64)):
        ens.set("kernel", kernel).set("refill_threshold", thr)
        _, _, cache = G.prerendergeodesics(m, X_FAR, d, 2000.0, image_width=96, image_height=72, alpha_lims=ALIMS,
                                           beta_lims=BLIMS, ensemble=ens)
        out.append(cache.points.copy())
    ens.set("refill_threshold", 8)
    for o in out[1:]:
        assert o.tobytes() == out[0].tobytes()


def test_pixel_tile_shape_changes_no_ray(G, ens):
    """8 x 8 tiles, 16 x 4 tiles (a wave's stores = four whole 128-byte lines; gr_ctx_set "tile_rows") and no tiling give
    the same image and the same end points bit for bit, also on planes whose height fits only one of the shapes and on a
    sharded range; the LPT tile order (persistent kernel) follows the tile shape."""
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    for (W, H) in ((96, 64), (64, 72), (40, 48)):          # H % 16 == 0 ; H % 8 == 0 only ; both
        imgs, pts = [], []
        for rows, swz, kern in ((8, 1, 0), (16, 1, 0), (8, 0, 0), (16, 1, 1)):
            ens.set("tile_rows", rows).set("swizzle", swz).set("kernel", kern).set("lpt", 2 if kern == 1 else 1)
            _, _, img = G.rendergeodesics(m, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS, beta_lims=BLIMS,
                                          pf=pf, ensemble=ens)
            _, _, cache = G.prerendergeodesics(m, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS,
                                               beta_lims=BLIMS, ensemble=ens)
            imgs.append(img.copy())
            pts.append(cache.points.copy())
        ens.set("tile_rows", 8).set("swizzle", 1).set("kernel", 2).set("lpt", 1)
        for k in range(1, len(imgs)):
            assert imgs[k].tobytes() == imgs[0].tobytes(), (W, H, k)
            assert pts[k].tobytes() == pts[0].tobytes(), (W, H, k)


def test_tracegeodesics_arrays_and_polar_counts(G, oracle, ens):
    """tracegeodesics(m, x, plane, λ) on the device reproduces the reference's exact
    WithinInnerBoundary counts (test/image-planes/test-polar-grids.jl:13-21)."""
    ens.set("kernel", 1)
    m = G.KerrMetric()
    u = np.array([1.0, 1e3, math.pi / 2, 0.0])
    for grid, n in ((G.LinearGrid(), 10), (G.GeometricGrid(), 30), (G.InverseGrid(), 80)):
        pts = G.tracegeodesics(m, u, G.PolarPlane(grid, Nr=10, Nθ=10), (0.0, 2000.0), ensemble=ens)
        assert int(np.sum(pts["status"] == G.StatusCodes.WithinInnerBoundary)) == n
    kw = dict(x_min=0.1, y_min=0.1, Nx=12, Ny=12)
    for grid, n in ((G.LinearGrid(), 1), (G.GeometricGrid(), 25), (G.InverseGrid(), 81)):
        pts = G.tracegeodesics(m, u, G.CartesianPlane(grid, **kw), (0.0, 2000.0), ensemble=ens)
        assert int(np.sum(pts["status"] == G.StatusCodes.WithinInnerBoundary)) == n


def test_tracegeodesics_per_ray_positions(G, oracle, ens):
    """(xs, vs) array input shape of geodesic-problem.jl:141-150, ragged size (not a multiple of 64)."""
    m = G.KerrMetric(1.0, 0.5)
    rng = np.random.default_rng(7)
    n = 333
    xs = np.column_stack([np.zeros(n), rng.uniform(20, 60, n), rng.uniform(0.4, 2.7, n), rng.uniform(0, 6, n)])
    vs = np.stack([G.map_impact_parameters(m, x, a, b) for x, a, b in
                   zip(xs, rng.uniform(-8, 8, n), rng.uniform(-8, 8, n))])
    got = G.tracegeodesics(m, xs, vs, G.ThinDisc(2.0, 30.0), (0.0, 300.0), ensemble=ens)
    cfg = oracle.make_config("kerr", (1.0, 0.5), disc=(2.0, 30.0), lambda_max=300.0)
    ref = oracle.trace(cfg, xs, vs)
    _compare_points(G, oracle, got, ref)


def test_empty_and_single(G, ens):
    m = G.KerrMetric()
    x = X_SMOKE
    pts = G.tracegeodesics(m, x, np.zeros((0, 4)), 200.0, ensemble=ens)
    assert pts.shape == (0,)
    v = G.map_impact_parameters(m, x, 1.0, 1.0)
    pts = G.tracegeodesics(m, x, v, 200.0, ensemble=ens)
    assert pts.shape == (1,) and pts["status"][0] == G.StatusCodes.WithinInnerBoundary
    _, _, img = G.rendergeodesics(m, x, 200.0, image_width=1, image_height=1, alpha_lims=(0, 0), beta_lims=(0, 0),
                                  ensemble=ens)
    assert img.shape == (1, 1)


def test_invalid_arguments_return_errors(G, ens):
    with pytest.raises(G.GradusMI355XError, match="INVALID_ARGUMENT"):
        G.tracegeodesics(G.KerrMetric(), X_SMOKE, np.zeros((2, 4)), 200.0, abstol=-1.0, ensemble=ens)
    with pytest.raises(G.GradusMI355XError, match="INVALID_ARGUMENT"):
        ens.set("kernel", 7)
    ens.set("kernel", 1)


def test_full_size_properties_1024(G, oracle, ens):
    """BASELINE config C2 at full size: conservation laws on every ray and oracle parity on a
    strided subset of the same pixels."""
    ens.set("kernel", 1)
    m = G.KerrMetric(1.0, 0.998)
    isco = m.isco()
    W = H = 1024
    _, _, cache = G.prerendergeodesics(m, X_FAR, G.ThinDisc(isco, 50.0), 2000.0, image_width=W, image_height=H,
                                       alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=ens)
    pts = np.ascontiguousarray(cache.points.T).ravel()
    assert pts.size == W * H and np.all(pts["flags"] == 0)
    assert np.all(pts["status"] != G.StatusCodes.NoStatus) or np.all(pts["lambda_max"][pts["status"] == 3] == 2000.0)

    def EL(x, v):
        s2 = np.sin(x[:, 2]) ** 2
        Sig = x[:, 1] ** 2 + 0.998 ** 2 * (1 - s2)
        w = 2 * x[:, 1] / Sig
        gtt, gtp = w - 1, -0.998 * s2 * w
        gpp = s2 * (x[:, 1] ** 2 + 0.998 ** 2 - 0.998 * gtp)
        return -(gtt * v[:, 0] + gtp * v[:, 3]), gtp * v[:, 0] + gpp * v[:, 3]

    E0, L0 = EL(pts["x_init"], pts["v_init"])
    E1, L1 = EL(pts["x"], pts["v"])
    fin = pts["status"] != G.StatusCodes.WithinInnerBoundary   # near the horizon Δ→0 amplifies rounding
    assert np.max(np.abs(E1[fin] / E0[fin] - 1)) < 1e-6
    assert np.max(np.abs(L1[fin] - L0[fin]) / np.maximum(np.abs(L0[fin]), 1.0)) < 1e-6
    hit = pts["status"] == G.StatusCodes.IntersectedWithGeometry
    assert hit.sum() > 50_000
    assert np.all(np.abs(np.cos(pts["x"][hit, 2])) <= 0.01 + 1e-9)

    # oracle parity on every 16th pixel in both directions (4096 rays)
    idx = (np.arange(0, W, 16)[:, None] * H + np.arange(0, H, 16)[None, :]).ravel()
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc=(isco, 50.0), lambda_max=2000.0)
    v_all = np.concatenate([oracle.render_velocities(cfg, X_FAR, ALIMS, BLIMS, W, H, i0=int(i), n=1) for i in idx])
    ref = oracle.trace(cfg, X_FAR, v_all)
    _compare_points(G, oracle, pts[idx], ref)


def test_full_size_properties_2048_bench_workload(G, oracle, ens):
    """The bench workload itself (BASELINE config C3, 2048² = 4 194 304 rays) through size-independent
    properties: the fused image equals the redshift of the end-point records ray for ray, E and L_z are
    conserved on every ray, every hit lies on the gtol wedge inside the disc's radial range, the
    image is mirror-consistent with the end points' classification; and oracle parity on a 64 x 64
    strided subset of the same pixels.  Default launch shape (one-wave workgroups)."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    isco = m.isco()
    W = H = 2048
    d = G.ThinDisc(isco, 50.0)
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=W, image_height=H, alpha_lims=ALIMS, beta_lims=BLIMS, ensemble=ens)
    _, _, img, st = G.rendergeodesics(m, X_FAR, d, 2000.0, pf=pf, stats=True, **kw)
    assert st["rays"] == W * H and st["flagged_rays"] == 0
    _, _, cache = G.prerendergeodesics(m, X_FAR, d, 2000.0, **kw)
    pts = np.ascontiguousarray(cache.points.T).ravel()
    hit = pts["status"] == G.StatusCodes.IntersectedWithGeometry
    assert np.array_equal(np.isfinite(img).T.ravel(), hit)                   # same rays classified as hits
    again = G.apply(pf, cache)
    np.testing.assert_array_equal(np.isnan(again), np.isnan(img))
    np.testing.assert_allclose(again[~np.isnan(img)], img[~np.isnan(img)], rtol=1e-12)   # fused == endpoints + apply
    assert int(hit.sum()) == st["status_count"][2] and 1_250_000 < hit.sum() < 1_350_000

    s2 = np.sin(pts["x"][:, 2]) ** 2
    def EL(x, v):
        s2 = np.sin(x[:, 2]) ** 2
        Sig = x[:, 1] ** 2 + 0.998 ** 2 * (1 - s2)
        w = 2 * x[:, 1] / Sig
        gtt, gtp = w - 1, -0.998 * s2 * w
        gpp = s2 * (x[:, 1] ** 2 + 0.998 ** 2 - 0.998 * gtp)
        return -(gtt * v[:, 0] + gtp * v[:, 3]), gtp * v[:, 0] + gpp * v[:, 3]

    E0, L0 = EL(pts["x_init"], pts["v_init"])
    E1, L1 = EL(pts["x"], pts["v"])
    fin = pts["status"] != G.StatusCodes.WithinInnerBoundary
    assert np.max(np.abs(E1[fin] / E0[fin] - 1)) < 1e-6
    assert np.max(np.abs(L1[fin] - L0[fin]) / np.maximum(np.abs(L0[fin]), 1.0)) < 1e-6
    assert np.all(np.abs(np.cos(pts["x"][hit, 2])) <= 0.01 + 1e-9)

    def carter(x, v, E, L):
        # Q = p_θ² + cos²θ (L²/sin²θ - a² E²) for a null geodesic, p_θ = Σ v^θ
        c2 = np.cos(x[:, 2]) ** 2
        Sig = x[:, 1] ** 2 + 0.998 ** 2 * c2
        return (Sig * v[:, 2]) ** 2 + c2 * (L * L / (1 - c2) - 0.998 ** 2 * E * E)

    Q0, Q1 = carter(pts["x_init"], pts["v_init"], E0, L0), carter(pts["x"], pts["v"], E1, L1)
    assert np.max(np.abs(Q1[fin] - Q0[fin]) / np.maximum(np.abs(Q0[fin]), 1.0)) < 1e-6      # Carter's constant
    rho = pts["x"][hit, 1] * np.sqrt(s2[hit])
    assert rho.min() >= isco * (1 - 1e-6) and rho.max() <= 50.0 * (1 + 1e-6)     # rim hits sit on the radial edge itself
    # closed-form redshift of a Keplerian emitter seen by a static distant observer: g = 1/(u^t (1 - Ω L/E))
    r = rho
    Om = 1.0 / (r ** 1.5 + 0.998)
    ut = (r ** 1.5 + 0.998) / np.sqrt(r ** 3 - 3 * r ** 2 + 2 * 0.998 * r ** 1.5)
    g_closed = 1.0 / (ut * (1.0 - Om * L1[hit] / E1[hit]))
    g_img = img.T.ravel()[hit]
    # the disc surface is the |cosθ| = 0.01 wedge, not the equator: the closed form (equatorial) agrees to O(gtol²)
    assert np.max(np.abs(g_img / g_closed - 1)) < 3e-3 and np.median(np.abs(g_img / g_closed - 1)) < 3e-4

    idx = (np.arange(0, W, 32)[:, None] * H + np.arange(0, H, 32)[None, :]).ravel()
    cfg = oracle.make_config("kerr", (1.0, 0.998), disc=(isco, 50.0), lambda_max=2000.0)
    v_all = np.concatenate([oracle.render_velocities(cfg, X_FAR, ALIMS, BLIMS, W, H, i0=int(i), n=1) for i in idx])
    ref = oracle.trace(cfg, X_FAR, v_all)
    _compare_points(G, oracle, pts[idx], ref)
    gref = oracle.apply_pf(cfg, ref, 2000.0, pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_INTERSECTED, r_isco=isco)
    both = np.isfinite(gref) & np.isfinite(img.T.ravel()[idx])
    assert both.sum() > 1000
    np.testing.assert_allclose(img.T.ravel()[idx][both], gref[both], rtol=RTOL)      # north-star tolerance


# ---------------- BASELINE config C4: JohannsenMetric + ThinDisc, interpolated redshift ----------------
JOH = (1.0, 0.7, 2.0, 0.0, 0.0, 1.0)     # docs/src/getting-started.md:393


def test_single_geodesic_path_and_plunging_table(G, oracle, ens):
    """tracegeodesics for ONE geodesic (saved path) and the PlungingInterpolation built from it
    (src/orbits/orbit-solving.jl:99-167) against the oracle's plunge."""
    m = G.JohannsenMetric(*JOH)
    isco = m.isco()
    ocfg = oracle.make_config("johannsen", JOH)
    assert isco == pytest.approx(oracle.isco(ocfg), rel=1e-12)
    r, vt, vr, vp = G.interpolate_plunging_velocities(m, ensemble=ens)
    ro, vto, vro, vpo = oracle.plunging_table(ocfg, isco)
    assert abs(len(r) - len(ro)) <= 0.05 * len(ro) and len(r) > 50   # the start at the ISCO is marginally unstable
    assert np.all(np.diff(r) > 0) and r[-1] < isco and r[0] < m.inner_radius() * 1.01
    # same curve v(r): compare on the oracle's nodes inside the common range (linear interpolation
    # between adaptive nodes is only good to ~(Δr)², so this is a 1e-3 check of the curve itself)
    sel = (ro > r[1]) & (ro < r[-2])
    for mine, ref in ((vt, vto), (vr, vro), (vp, vpo)):
        np.testing.assert_allclose(np.interp(ro[sel], r, mine), ref[sel], rtol=2e-3, atol=1e-6)
    # a null geodesic path: first row = initial state, last row = end point record
    x = X_SMOKE
    v = G.map_impact_parameters(m, x, 3.0, 4.0)
    path = G.tracegeodesic_path(m, x, v, G.ThinDisc(2.0, 40.0), 200.0, ensemble=ens)
    assert path.λ[0] == 0.0 and np.all(np.diff(path.λ) > 0)
    np.testing.assert_array_equal(path.x[0], x)
    np.testing.assert_array_equal(path.x[-1], path.point["x"])
    assert path.λ[-1] == path.point["lambda_max"]
    pts = G.tracegeodesics(m, x, v, G.ThinDisc(2.0, 40.0), 200.0, ensemble=ens)
    assert pts[0].tobytes() == path.point.tobytes()


@pytest.mark.parametrize("disc_in", ["isco", 2.0])
def test_johannsen_redshift_matches_oracle(G, oracle, ens, disc_in):
    m = G.JohannsenMetric(*JOH)
    isco = m.isco()
    r_in = isco if disc_in == "isco" else disc_in
    x = np.array([0.0, 1000.0, math.radians(70), 0.0])
    W = H = 96
    pf = G.ConstPointFunctions.redshift(m, x, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
    _, _, img, st = G.rendergeodesics(m, x, G.ThinDisc(r_in, 50.0), 2000.0, image_width=W, image_height=H,
                                      alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf, ensemble=ens, stats=True)
    ocfg = oracle.make_config("johannsen", JOH, disc=(r_in, 50.0), lambda_max=2000.0)
    # the oracle is given the SAME plunging table: the table's nodes are step-sequence dependent
    ref, pts = oracle.rendergeodesics(ocfg, x, ALIMS, BLIMS, W, H, pf_id=oracle.PF_REDSHIFT,
                                      filter_id=oracle.FILTER_INTERSECTED, r_isco=isco, plunge=pf.extra["plunge"],
                                      return_points=True)
    assert st["flagged_rays"] == 0
    assert (np.isnan(img) != np.isnan(ref)).sum() <= 0.002 * W * H
    both = ~np.isnan(img) & ~np.isnan(ref)
    assert both.sum() > 500
    np.testing.assert_allclose(img[both], ref[both], rtol=RTOL)
    if disc_in != "isco":
        rho = pts["x"][:, 1] * np.abs(np.sin(pts["x"][:, 2]))
        assert ((pts["status"] == 2) & (rho < isco)).sum() > 20      # the interpolated branch is exercised


def test_lpt_tile_order_changes_nothing_but_time(G, ens):
    """Longest-first tile scheduling learned from the previous render of the same plane must give a
    bit-identical image (only the order of the work queue changes)."""
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=512, image_height=512, alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf, ensemble=ens)
    ens.set("kernel", 1).set("lpt", 0)
    _, _, ref = G.rendergeodesics(m, X_FAR, d, 2000.0, **kw)
    ens.set("lpt", 2)
    imgs = [G.rendergeodesics(m, X_FAR, d, 2000.0, **kw)[2] for _ in range(3)]   # record, sort+use, use
    ens.set("lpt", 1)
    for im in imgs:
        assert im.tobytes() == ref.tobytes()


# ---------------- BASELINE config C5: BinningMethod line profile ----------------
def _oracle_lineprofile(oracle, G, name, params, u, disc, plane, bins, q, rmin, rmax):
    """lineprofile(..., BinningMethod()) restated from oracle pieces + numpy (line-profiles.jl:152-198)."""
    λ_max = 2.0 * u[1]
    cfg = oracle.make_config(name, params, disc=disc, lambda_max=λ_max, upper_hemisphere=True)
    a, b = G.impact_parameters(plane, u)
    v = oracle.map_impact_parameters(cfg, u, a, b)
    pts = oracle.trace(cfg, u, v)
    g = oracle.apply_pf(cfg, pts, λ_max, pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_INTERSECTED, r_isco=rmin)
    rho = pts["x"][:, 1] * np.abs(np.sin(pts["x"][:, 2]))
    I = (pts["status"] == oracle.INTERSECTED_WITH_GEOMETRY) & (rho >= rmin) & (rho <= rmax)
    areas = G.unnormalized_areas(plane).ravel(order="F")
    f = rho[I] ** (-q) * g[I] ** 3 * areas[I]
    # bucket(Simple(), g, f, bins): last edge <= g, clamped (pinned by test_corona_host.py on the
    # reference's golden emissivity vector); restated here independently of the package
    idx = np.clip(np.searchsorted(bins, g[I], side="right") - 1, 0, bins.size - 1)
    flux = np.bincount(idx, weights=f, minlength=bins.size)
    return flux / flux.sum()


@pytest.mark.parametrize("shape", [(128, 256), (100, 77)])
def test_separable_polar_plane_equals_explicit_rays(G, ens, monkeypatch, shape):
    """A PolarPlane crosses the boundary as three small tables (gr_rayset.sep_*, the device forms α = r_i cos θ_j, ...)
    or as explicit α / β / area arrays: same rays bit for bit, so the same line profile (up to the order of the fp64
    atomic adds) on the fused route and the same (g, ρ) pairs, weighted by the same areas, on the generic route."""
    import time

    m = G.KerrMetric(1.0, 0.998)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(m.isco(), 250.0)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=shape[0], Nθ=shape[1], r_min=1.0, r_max=250.0)
    bins = np.linspace(0.1, 1.5, 180)
    out = {}
    for sep in ("1", "0"):
        monkeypatch.setenv("GRADUS_MI355X_SEPARABLE_RAYS", sep)
        t0 = time.perf_counter()
        _, y_fused, st = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0,
                                       ensemble=ens, stats=True)
        _, y_gen = G.lineprofile(bins, lambda r: r ** -3.0, m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0, ensemble=ens)
        out[sep] = (y_fused, y_gen, st, time.perf_counter() - t0)
    for k in ("accepted_steps", "rejected_steps", "rays"):
        assert out["1"][2][k] == out["0"][2][k]
    np.testing.assert_allclose(out["1"][0], out["0"][0], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(out["1"][1], out["0"][1], rtol=1e-11, atol=1e-15)
    np.testing.assert_allclose(out["1"][0], out["1"][1], rtol=1e-9, atol=1e-13)


@pytest.mark.parametrize("shape", [(8, 8), (5, 30), (257, 513)])
def test_separable_rays_across_kernels_and_precisions(G, ens, monkeypatch, shape):
    """Separable against explicit ray sets for a one-tile plane, an untiled one (fewer than 8 radii) and a ragged one, at a
    tight tolerance (one ray per lane) and a loose one (persistent kernel with refill), fp64 and fp32: the same rays, so the
    same number of integration steps and the same profile."""
    m = G.KerrMetric(1.0, 0.998)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(m.isco(), 250.0)
    bins = np.linspace(0.1, 1.5, 180)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=shape[0], Nθ=shape[1], r_min=1.0, r_max=250.0)
    for tol in (1e-9, 1e-4):
        for prec in (64, 32):
            ens.set("precision", prec)
            out = {}
            for sep in ("1", "0"):
                monkeypatch.setenv("GRADUS_MI355X_SEPARABLE_RAYS", sep)
                _, y, st = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0,
                                         ensemble=ens, stats=True, abstol=tol, reltol=tol)
                out[sep] = (y, st)
            assert out["1"][1]["rays"] == shape[0] * shape[1]
            assert out["1"][1]["accepted_steps"] == out["0"][1]["accepted_steps"], (tol, prec)
            np.testing.assert_allclose(out["1"][0], out["0"][0], rtol=1e-9 if prec == 64 else 1e-4, atol=1e-12)


def test_lineprofile_with_an_emissivity_profile_is_fused(G, ens, monkeypatch):
    """lineprofile(bins, prof::RadialDiscProfile, m, u, d, BinningMethod()): the profile's table is interpolated on the
    device (emissivity_at: clamp to the table's range + NaNLinearInterpolator, src/corona/radial.jl:15-18,
    src/interpolations.jl:1-30) and binned there.  Same profile as the generic route (pairs back to the host, numpy
    interpolation and bucketing) -- with a table that has a NaN node, with hits beyond both ends of the table (clamped),
    and for a lamp-post emissivity profile traced on the device."""
    from gradus_jl_amd import lineprofiles as LP
    from gradus_jl_amd.corona import RadialDiscProfile

    m = G.KerrMetric(1.0, 0.9)
    u = np.array([0.0, 1000.0, math.radians(55), 0.0])
    d = G.ThinDisc(m.isco(), 200.0)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=160, Nθ=200, r_min=1.0, r_max=300.0)
    bins = np.linspace(0.1, 1.6, 120)
    radii = np.geomspace(4.0, 120.0, 40)                      # the disc extends beyond both ends: clamped there
    eps = radii ** -2.7 * (1.0 + 0.3 * np.sin(radii))
    eps[7] = np.nan                                            # a bin no corona ray fell into
    lamp = G.emissivity_profile(m, d, G.LampPostModel(h=8.0), n_samples=400, ensemble=ens)
    for prof in (RadialDiscProfile(radii, eps, np.zeros_like(radii)), lamp):
        _, y_fused, st = G.lineprofile(bins, prof, m, u, d, G.BinningMethod(), plane=plane, maxrₑ=200.0, ensemble=ens, stats=True)
        monkeypatch.setattr(LP, "_emissivity_table", lambda e: None)        # force the generic route
        _, y_gen = G.lineprofile(bins, prof, m, u, d, G.BinningMethod(), plane=plane, maxrₑ=200.0, ensemble=ens)
        monkeypatch.undo()
        assert st["rays"] == 160 * 200 and y_fused.sum() == pytest.approx(1.0, abs=1e-12) and np.count_nonzero(y_fused) > 40
        np.testing.assert_allclose(y_fused, y_gen, rtol=1e-10, atol=1e-15)


def test_tracegeodesics_on_a_polar_plane_forms_its_rays_on_the_device(G, ens, monkeypatch):
    """tracegeodesics(m, u, plane::PolarPlane, d, ...) (the call inside the reference's lineprofile, line-profiles.jl:171-183):
    the plane goes over as its tables and trajectory i is ray i of vec(αs) -- same statuses and end points as with host-built
    (x, v) arrays (whose velocities come from a numpy product instead of the kernel's FMAs: agreement to rounding)."""
    m = G.KerrMetric(1.0, 0.9)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(m.isco(), 80.0)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=37, Nθ=50, r_min=1.0, r_max=120.0)
    got = {}
    for sep in ("1", "0"):
        monkeypatch.setenv("GRADUS_MI355X_SEPARABLE_RAYS", sep)
        got[sep] = G.tracegeodesics(m, u, plane, d, (0.0, 2000.0), ensemble=ens, callback=G.domain_upper_hemisphere())
    a, b = got["1"], got["0"]
    assert a.size == b.size == 37 * 50
    assert (a["status"] != b["status"]).sum() <= 2
    same = a["status"] == b["status"]
    np.testing.assert_allclose(a["v_init"][same], b["v_init"][same], rtol=1e-13, atol=1e-16)
    hit = same & (a["status"] == 2)
    assert hit.sum() > 300
    np.testing.assert_allclose(a["x"][hit], b["x"][hit], rtol=1e-7, atol=1e-9)
    # vec(αs) order: trajectory k is (r index, θ index) = (k % Nr, k // Nr)
    αs, βs = G.impact_parameters(plane, u)
    cfg = G.tracing_configuration(m, u, np.zeros((1, 4)), d, (0.0, 2000.0), ensemble=ens)
    v = G.map_impact_parameters(m, u, αs[[0, 36, 37, 1849]], βs[[0, 36, 37, 1849]])
    np.testing.assert_allclose(a["v_init"][[0, 36, 37, 1849], 1:], v[:, 1:], rtol=1e-12, atol=1e-15)


@pytest.mark.parametrize("world", [2, 3, 8])
def test_sharded_lineprofile_reassembles(G, ens, world):
    """distributed.lineprofile_sharded on one GPU: the partial histograms of every rank's shard (block-cyclic strips of
    the plane's tile order, gr_rayset.sep_first / sep_block / sep_stride) sum to the single-launch histogram, the rays
    and steps add up exactly, and the one-rank call equals lineprofile(...)."""
    import torch

    from gradus_jl_amd.device import lineprofile_device, new_stats, stats_dict
    from gradus_jl_amd.distributed import lineprofile_sharded, ray_shard

    m = G.KerrMetric(1.0, 0.998)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(m.isco(), 250.0)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=200, Nθ=132, r_min=1.0, r_max=250.0)          # ragged: 200 = 25 x 8, 132 = 16.5 x 8
    bins = np.linspace(0.1, 1.5, 180)
    eps = G.PowerLawEmissivity(3)
    st_all = new_stats(torch.device("cuda", 0))
    whole = lineprofile_device(bins, eps, m, u, d, plane, maxrₑ=250.0, ensemble=ens, stats=st_all)
    parts, rays, steps = torch.zeros_like(whole), 0, 0
    for r in range(world):
        st = new_stats(torch.device("cuda", 0))
        parts += lineprofile_device(bins, eps, m, u, d, plane, shard=ray_shard(plane, world, r), maxrₑ=250.0, ensemble=ens, stats=st)
        sd = stats_dict(st)
        rays += sd["rays"]
        steps += sd["accepted_steps"]
    torch.cuda.synchronize()
    sa = stats_dict(st_all)
    assert rays == sa["rays"] == 200 * 132 and steps == sa["accepted_steps"]
    np.testing.assert_allclose(parts.cpu().numpy(), whole.cpu().numpy(), rtol=1e-11, atol=1e-18)
    if world == 2:
        # an emissivity profile (table) through the device-resident route equals the host-buffer route
        from gradus_jl_amd.corona import RadialDiscProfile

        rr = np.geomspace(2.0, 200.0, 30)
        prof = RadialDiscProfile(rr, rr ** -2.5, np.zeros_like(rr))
        _, yt = lineprofile_sharded(bins, prof, m, u, d, plane, maxrₑ=250.0, ensemble=ens)
        _, yt_ref = G.lineprofile(bins, prof, m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0, ensemble=ens)
        np.testing.assert_allclose(yt, yt_ref, rtol=1e-11, atol=1e-18)
        _, y = lineprofile_sharded(bins, eps, m, u, d, plane, maxrₑ=250.0, ensemble=ens)
        _, y_ref = G.lineprofile(bins, eps, m, u, d, G.BinningMethod(), plane=plane, maxrₑ=250.0, ensemble=ens)
        np.testing.assert_allclose(y, y_ref, rtol=1e-11, atol=1e-18)


def test_lineprofile_binning_matches_oracle_and_reference_edges(G, oracle, ens):
    """test/line-profiles/test-binning.jl:5-32 on the device (fused and generic paths) + oracle parity."""
    m = G.KerrMetric(M=1.0, a=0.6)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(m.isco(), 250.0)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=100, Nθ=400)
    bins = np.linspace(0.1, 1.3, 100)
    x, y = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, d, G.BinningMethod(), plane=plane, ensemble=ens)
    x2, y2 = G.lineprofile(bins, lambda r: r ** -3.0, m, u, d, G.BinningMethod(), plane=plane, ensemble=ens)
    np.testing.assert_allclose(y, y2, rtol=1e-9, atol=1e-15)          # fused == generic path
    assert y.sum() == pytest.approx(1.0)
    g_low = x[np.argmax(y > 0)]
    g_high = x[len(y) - 1 - np.argmax(y[::-1] > 0) - 1]
    assert g_low == pytest.approx(0.355, abs=0.05)
    assert g_high == pytest.approx(1.2, abs=0.05)
    ref = _oracle_lineprofile(oracle, G, "kerr", (1.0, 0.6), u, (m.isco(), 250.0), plane, bins, 3.0, m.isco(), 50.0)
    # a rim ray switching bins moves ~1e-4 of the flux; bulk agreement is much tighter
    assert np.abs(y - ref).sum() < 2e-3
    assert np.max(np.abs(y - ref)) < 5e-4


def test_fp32_kernels_track_fp64(G, ens):
    """The fp32 instantiation (gr_ctx_set("precision", 32)) at tol 1e-5 against fp64 at 1e-9 on the
    C2 scene: same classification up to rim pixels, redshift to ~1e-3."""
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=128, image_height=128, alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf, ensemble=ens)
    ens.set("kernel", 1).set("precision", 64)
    _, _, ref = G.rendergeodesics(m, X_FAR, d, 2000.0, **kw)
    ens.set("precision", 32)
    try:
        _, _, img, st = G.rendergeodesics(m, X_FAR, d, 2000.0, abstol=1e-5, reltol=1e-5, stats=True, **kw)
    finally:
        ens.set("precision", 64)
    assert st["rays"] == 128 * 128 and st["flagged_rays"] <= 0.01 * st["rays"]
    assert (np.isnan(img) != np.isnan(ref)).sum() <= 0.02 * img.size
    both = ~np.isnan(img) & ~np.isnan(ref)
    assert both.sum() > 1000
    assert np.median(np.abs(img[both] / ref[both] - 1)) < 2e-4
    assert np.percentile(np.abs(img[both] / ref[both] - 1), 99) < 2e-2


@pytest.mark.parametrize("world", [2, 8])
def test_sharded_ranges_reassemble_to_single_render(G, ens, world):
    """The multi-GPU decomposition on ONE device: render every rank's block-cyclic gr_range through
    the device entry point, undo the deal exactly as gather_image does, and compare bit for bit with
    the unsharded render (the RCCL gather itself is covered with gloo in test_distributed_cpu.py)."""
    import torch
    from gradus_jl_amd import device as gdev

    ens.set("kernel", 1).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    W, H = 256, 192
    cfg = G.render_configuration(m, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS,
                                 beta_lims=BLIMS, ensemble=ens)
    dev = torch.device("cuda", 0)
    full = torch.empty(W * H, dtype=torch.float64, device=dev)
    gdev.render_device(cfg, pf, full)
    slabs = []
    for rank in range(world):
        plan = G.shard_plan(W, H, world, rank)
        local = torch.empty(plan.count, dtype=torch.float64, device=dev)
        gdev.render_device(cfg, pf, local, plan.ray_range())
        slabs.append(local)
    torch.cuda.synchronize()
    plan = G.shard_plan(W, H, world, 0)
    re = torch.stack(slabs).view(world, plan.n_blocks, plan.block).permute(1, 0, 2).reshape(-1)
    assert re.cpu().numpy().tobytes() == full.cpu().numpy().tobytes()
    img = full.view(W, H).t().cpu().numpy()
    _, _, ref = G.rendergeodesics(m, X_FAR, d, 2000.0, image_width=W, image_height=H, alpha_lims=ALIMS,
                                  beta_lims=BLIMS, pf=pf, ensemble=ens)
    assert img.tobytes() == ref.tobytes()


def test_ragged_range_without_tiles(G, oracle, ens):
    """A range that is not a whole number of 8x8 tiles (H = 50, odd first/count) takes the linear
    index path; compare a slice of the image with the full render."""
    import ctypes as C

    ens.set("kernel", 1)
    m = G.KerrMetric(1.0, 0.5)
    W, H = 37, 50
    cfg = G.render_configuration(m, X_SMOKE, G.ThinDisc(0.0, 40.0), 200.0, image_width=W, image_height=H,
                                 alpha_lims=(-12, 12), beta_lims=(-12, 12), ensemble=ens)
    full = G.render_into_image(cfg, pf=G.ConstPointFunctions.shadow())
    L = G._lib
    c, pl = cfg.abi_config(), cfg.abi_plane()
    from gradus_jl_amd.rendering import abi_pointfunction
    s, _ = abi_pointfunction(G.ConstPointFunctions.shadow())
    first, count = 123, 777
    rg = L.gr_range(first, count, count, 1)
    out = np.zeros(count)
    L.check(L.load().gr_render(ens.ctx.handle, C.byref(c), C.byref(pl), C.byref(s), C.byref(rg), out.ctypes.data, None))
    lin = full.T.ravel()[first:first + count]
    np.testing.assert_array_equal(np.isnan(out), np.isnan(lin))
    np.testing.assert_array_equal(out[~np.isnan(out)], lin[~np.isnan(lin)])



def test_more_reference_fingerprints_on_device(G, ens):
    """Morris-Thorne shadow (rendergeodesics.jl:44), Johannsen-Psaltis chart test
    (test/integration/test-charts.jl:5-18) and Kerr-Newman q = 0 (test/unit/metrics.kerr-newman.jl:25)."""
    ens.set("kernel", 1).set("precision", 64)
    _, _, img = G.rendergeodesics(G.MorrisThorneWormhole(1.0), X_SMOKE, 200.0, image_width=20, image_height=20,
                                  alpha_lims=(-9.5, 9.5), beta_lims=(-9.5, 9.5), ensemble=ens)
    assert float(np.nansum(img)) == pytest.approx(402.17907632733284, rel=1e-4)
    u = np.array([0.0, 1000.0, math.pi / 2, 0.0])
    _, _, img = G.rendergeodesics(G.JohannsenPsaltisMetric(1.0, 0.8831, 0.4), u, 2000.0, image_width=100,
                                  image_height=100, alpha_lims=(-8, 8), beta_lims=(-8, 8), ensemble=ens)
    assert float(np.nansum(img)) == pytest.approx(2.9619136946153212e6, rel=1e-6)
    _, _, img = G.rendergeodesics(G.KerrNewmanMetric(1.0, 0.6, 0.6), u, 2000.0, image_width=40, image_height=40,
                                  alpha_lims=(-8, 8), beta_lims=(-8, 8), ensemble=ens)
    assert float(np.nansum(img)) == pytest.approx(428809.9681726607, rel=1e-6)


def test_event_horizon_chart_on_device(G, oracle, ens):
    """PoloidalShapeChart(event_horizon shape × 1.001) for the near-naked-singularity Johannsen-Psaltis
    metric: reference fingerprint (test/integration/test-charts.jl:20-31, rtol 1e-4 there) and oracle
    parity; both kernels; the fp32 build accepts the chart too."""
    u = np.array([0.0, 1000.0, math.pi / 2, 0.0])
    m = G.JohannsenPsaltisMetric(1.0, 0.8831, 0.4)
    chart = G.event_horizon_chart(m, closest_approach=1.001, resolution=100)
    assert chart.table.size == 100 and np.all(np.isfinite(chart.table))
    assert chart.table.min() < m.inner_radius() < chart.table.max() * 1.01     # the horizon is not a sphere here
    kw = dict(image_width=100, image_height=100, alpha_lims=(-8, 8), beta_lims=(-8, 8), ensemble=ens, chart=chart)
    for kernel in (0, 1):
        ens.set("kernel", kernel)
        _, _, img = G.rendergeodesics(m, u, 2000.0, **kw)
        assert float(np.nansum(img)) == pytest.approx(2.9540649115176247e6, rel=1e-6)
    _, _, cache = G.prerendergeodesics(m, u, 2000.0, **kw)
    got = np.ascontiguousarray(cache.points.T).ravel()
    ocfg = oracle.make_config("johannsen-psaltis", (1.0, 0.8831, 0.4), lambda_max=2000.0, chart_table=chart.table,
                              chart_theta=(chart.θ_first, chart.θ_last))
    ref = oracle.trace(ocfg, u, oracle.render_velocities(ocfg, u, (-8, 8), (-8, 8), 100, 100))
    _compare_points(G, oracle, got, ref)
    # captured rays stop on the tabulated surface, not on a sphere
    inner = got["status"] == oracle.WITHIN_INNER_BOUNDARY
    assert inner.sum() > 100
    assert np.all(got["x"][inner, 1] <= chart.shapefunc(got["x"][inner, 2]) * (1 + 1e-12))
    with pytest.raises(G.GradusMI355XError):
        bad = G.PoloidalShapeChart(chart.table, 1.0, 1.0)
        G.rendergeodesics(m, u, 2000.0, **{**kw, "chart": bad})


def test_charged_test_particles_in_kerr_newman(G, oracle, ens):
    """rendergeodesics(KerrNewmanMetric(1, 0.6, 0.6), ...; q = ±1): Lorentz force q F^μ_ν v^ν on the
    device against the reference's fingerprints (test/unit/metrics.kerr-newman.jl:26-27, rtol 1e-3
    there) and, end point by end point, against the oracle."""
    u = np.array([0.0, 1000.0, math.pi / 2, 0.0])
    m = G.KerrNewmanMetric(1.0, 0.6, 0.6)
    kw = dict(image_width=40, image_height=40, alpha_lims=(-8, 8), beta_lims=(-8, 8), ensemble=ens)
    for kernel in (0, 1):
        ens.set("kernel", kernel)
        for q, gold in ((1.0, 253280.6794972752), (-1.0, 619335.5670363897)):
            _, _, img = G.rendergeodesics(m, u, 2000.0, q=q, **kw)
            assert float(np.nansum(img)) == pytest.approx(gold, rel=1e-6)
    _, _, cache = G.prerendergeodesics(m, u, 2000.0, q=1.0, **kw)
    got = np.ascontiguousarray(cache.points.T).ravel()
    ocfg = oracle.make_config("kerr-newman", (1.0, 0.6, 0.6), lambda_max=2000.0, q=1.0)
    ref = oracle.trace(ocfg, u, oracle.render_velocities(ocfg, u, (-8, 8), (-8, 8), 40, 40))
    _compare_points(G, oracle, got, ref)
    # the charge is ignored by metrics without an electromagnetic field (as in the reference)
    mk = G.KerrMetric(1.0, 0.6)
    a = G.rendergeodesics(mk, u, 2000.0, q=1.0, **kw)[2]
    b = G.rendergeodesics(mk, u, 2000.0, **kw)[2]
    assert a.tobytes() == b.tobytes()


def test_against_committed_golden_fixtures(G, ens):
    """tests/golden/*.npz (oracle output committed after the oracle was pinned on the reference's
    golden values; generator: tests/golden/make_fixtures.py)."""
    import os

    ens.set("kernel", 1).set("precision", 64)
    gdir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    f = np.load(os.path.join(gdir, "kerr_a0_thindisc_20x20_endpoints.npz"))
    m = G.KerrMetric(*f["params"])
    _, _, cache = G.prerendergeodesics(m, f["x_obs"], G.ThinDisc(*f["disc"]), float(f["lambda_max"]),
                                       image_width=int(f["W"]), image_height=int(f["H"]), alpha_lims=tuple(f["alims"]),
                                       beta_lims=tuple(f["blims"]), ensemble=ens)
    got = np.ascontiguousarray(cache.points.T).ravel()
    ref = f["points"]
    assert (got["status"] != ref["status"]).sum() <= 1
    ok = (got["status"] == ref["status"]) & (ref["status"] != 1)
    np.testing.assert_allclose(got["lambda_max"][ok], ref["lambda_max"][ok], rtol=RTOL)
    np.testing.assert_allclose(got["x"][ok], ref["x"][ok], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(got["v"][ok], ref["v"][ok], rtol=RTOL, atol=1e-9)

    f = np.load(os.path.join(gdir, "kerr_a0998_c1_64x64_redshift.npz"))
    m = G.KerrMetric(*f["params"])
    pf = G.ConstPointFunctions.redshift(m, f["x_obs"]) @ G.ConstPointFunctions.filter_intersected()
    _, _, img = G.rendergeodesics(m, f["x_obs"], G.ThinDisc(*f["disc"]), float(f["lambda_max"]), image_width=int(f["W"]),
                                  image_height=int(f["H"]), alpha_lims=tuple(f["alims"]), beta_lims=tuple(f["blims"]),
                                  pf=pf, ensemble=ens)
    ref = f["image"]
    assert (np.isnan(img) != np.isnan(ref)).sum() <= 4
    both = ~np.isnan(img) & ~np.isnan(ref)
    assert both.sum() > 2000
    np.testing.assert_allclose(img[both], ref[both], rtol=RTOL)


def test_shakura_sunyaev_disc_matches_oracle(G, oracle, ens):
    ens.set("kernel", 1).set("precision", 64)
    m = G.KerrMetric(1.0, 0.9)
    d = G.ShakuraSunyaev.for_metric(m)
    ocfg0 = oracle.make_config("kerr", (1.0, 0.9))
    ss = oracle.shakura_sunyaev(ocfg0)
    W = H = 96
    _, _, cache = G.prerendergeodesics(m, X_SMOKE, d, 200.0, image_width=W, image_height=H, alpha_lims=(-30, 30),
                                       beta_lims=(-20, 20), ensemble=ens)
    got = np.ascontiguousarray(cache.points.T).ravel()
    ocfg = oracle.make_config("kerr", (1.0, 0.9), disc=ss, lambda_max=200.0)
    ref = oracle.trace(ocfg, X_SMOKE, oracle.render_velocities(ocfg, X_SMOKE, (-30, 30), (-20, 20), W, H))
    _compare_points(G, oracle, got, ref)
    assert (ref["status"] == 2).sum() > 1000


def test_device_resident_entry_points_and_streams(G, ens):
    """gr_render_device / gr_render_endpoints_device with torch-owned HBM on a non-default stream
    agree bit for bit with the host-buffer entry points; block sizes 64..256 agree too."""
    import torch
    from gradus_jl_amd import device as gdev

    ens.set("kernel", 1).set("precision", 64)
    m = G.KerrMetric(1.0, 0.7)
    d = G.ThinDisc(m.isco(), 30.0)
    pf = G.ConstPointFunctions.redshift(m, X_SMOKE) @ G.ConstPointFunctions.filter_intersected()
    W, H = 96, 64
    cfg = G.render_configuration(m, X_SMOKE, d, 200.0, image_width=W, image_height=H, alpha_lims=(-25, 25),
                                 beta_lims=(-15, 15), ensemble=ens)
    ref_img = G.render_into_image(cfg, pf=pf)
    ref_pts = G.ensemble_solve_tracing_problem(ens, cfg)
    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        img = torch.empty(W * H, dtype=torch.float64, device=dev)
        st = gdev.new_stats(dev)
        gdev.render_device(cfg, pf, img, stats=st)
        raw = torch.empty(W * H * 152, dtype=torch.uint8, device=dev)
        gdev.render_endpoints_device(cfg, raw)
    side.synchronize()
    assert img.view(W, H).t().cpu().numpy().tobytes() == ref_img.tobytes()
    assert gdev.points_from_tensor(raw, W * H).tobytes() == ref_pts.tobytes()
    assert gdev.stats_dict(st)["rays"] == W * H
    for block in (64, 128, 256):
        ens.set("block", block)
        assert G.render_into_image(cfg, pf=pf).tobytes() == ref_img.tobytes()
    ens.set("block", 256)
    with pytest.raises(G.GradusMI355XError):
        ens.set("block", 512)


def test_lds_staging_is_transparent(G, ens):
    """The LDS copies of the plunging table and of the line-profile histogram change nothing:
    Johannsen redshift image bit-identical with lds on / off; line profile equal up to the order
    of the fp64 atomic sums."""
    ens.set("kernel", 1).set("precision", 64)
    m = G.JohannsenMetric(*JOH)
    x = np.array([0.0, 1000.0, math.radians(70), 0.0])
    pf = G.ConstPointFunctions.redshift(m, x, ensemble=ens) @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=64, image_height=64, alpha_lims=(-20, 20), beta_lims=(-12, 12), pf=pf, ensemble=ens)
    imgs = []
    for lds in (1, 0):
        ens.set("lds", lds)
        imgs.append(G.rendergeodesics(m, x, G.ThinDisc(2.0, 50.0), 2000.0, **kw)[2])
    assert imgs[0].tobytes() == imgs[1].tobytes() and np.isfinite(imgs[0]).sum() > 200
    mk = G.KerrMetric(1.0, 0.6)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    plane = G.PolarPlane(G.GeometricGrid(), Nr=64, Nθ=128)
    bins = np.linspace(0.1, 1.3, 100)
    ys = []
    for lds in (1, 0):
        ens.set("lds", lds)
        ys.append(G.lineprofile(bins, G.PowerLawEmissivity(3), mk, u, G.ThinDisc(mk.isco(), 250.0), G.BinningMethod(), plane=plane,
                                ensemble=ens)[1])
    ens.set("lds", 1)
    np.testing.assert_allclose(ys[0], ys[1], rtol=1e-12, atol=1e-18)


def test_thick_disc_sampled_closure_on_device(G, oracle, ens):
    ens.set("kernel", 1).set("precision", 64)

    def torus(ρ):
        return -1.0 if (ρ < 9.0 or ρ > 11.0) else math.sqrt(1.0 - (ρ - 10.0) ** 2)

    m = G.KerrMetric(1.0, 0.9)
    d = G.ThickDisc(torus, ρ_range=(8.5, 11.5), samples=8192)
    W = H = 96
    _, _, cache = G.prerendergeodesics(m, X_SMOKE, d, 200.0, image_width=W, image_height=H, alpha_lims=(-14, 14),
                                       beta_lims=(-8, 8), ensemble=ens)
    got = np.ascontiguousarray(cache.points.T).ravel()
    ocfg = oracle.make_config("kerr", (1.0, 0.9), disc={"table": d.table, "range": d.ρ_range}, lambda_max=200.0)
    ref = oracle.trace(ocfg, X_SMOKE, oracle.render_velocities(ocfg, X_SMOKE, (-14, 14), (-8, 8), W, H))
    _compare_points(G, oracle, got, ref)
    assert (ref["status"] == 2).sum() > 500


def test_in_process_multi_device_render(G, ens):
    """gr_render_multi: several contexts driven from one host thread, block-cyclic columns, strided
    D2H straight into the image.  Exercised with 1, 2 and 4 contexts on the single device of the test
    box; must equal the single-context render bit for bit."""
    ens.set("kernel", 1).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    d = G.ThinDisc(m.isco(), 50.0)
    pf = G.ConstPointFunctions.redshift(m, X_FAR) @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=192, image_height=160, alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf)
    _, _, ref = G.rendergeodesics(m, X_FAR, d, 2000.0, ensemble=ens, **kw)
    for devs in ([0, 0], [0, 0, 0, 0]):
        multi = G.EnsembleMI355X(devices=devs)
        _, _, img, st = G.rendergeodesics(m, X_FAR, d, 2000.0, ensemble=multi, stats=True, **kw)
        assert st["rays"] == 192 * 160
        assert img.tobytes() == ref.tobytes()
    # 20 columns over 3 contexts: the library deals whole columns and refuses; the host side takes the largest leading
    # subset of the contexts whose number divides the width (EnsembleMI355X.contexts_for_width) -- here 2 of the 3
    three = G.EnsembleMI355X(devices=[0, 0, 0])
    assert len(three.contexts_for_width(20)) == 2 and len(three.contexts_for_width(21)) == 3
    kw20 = dict(image_width=20, image_height=20, alpha_lims=ALIMS, beta_lims=BLIMS, pf=pf)
    _, _, img = G.rendergeodesics(m, X_FAR, d, 2000.0, ensemble=three, **kw20)
    _, _, ref = G.rendergeodesics(m, X_FAR, d, 2000.0, ensemble=ens, **kw20)
    assert img.tobytes() == ref.tobytes()


# ---------------- §8 f-4: corona -> disc tracing and emissivity profiles ----------------
def test_corona_rays_match_oracle_and_sampler_matrix(G, oracle, ens):
    """tracegeodesics(m, model, d, λ; n_samples, sampler) for every sampler / generator / domain
    combination of test/smoke-tests/tracegeodesics.jl:44-66, end points against the oracle."""
    ens.set("kernel", 2).set("precision", 64)
    K = G.corona
    m = G.KerrMetric(M=1.0, a=0.0)
    d = G.ThinDisc(m.isco(), 50.0)
    model = G.LampPostModel(h=10.0, θ=math.radians(0.001))
    ocfg = oracle.make_config("kerr", (1.0, 0.0), disc=(m.isco(), 50.0), lambda_max=200.0)
    for Sampler in (G.EvenSampler, G.WeierstrassSampler):
        for gen in (G.GoldenSpiralGenerator(), G.RandomGenerator(seed=7)):
            for dom in (G.LowerHemisphere(), G.BothHemispheres()):
                s = Sampler(domain=dom, generator=gen)
                got = K.tracegeodesics(m, model, d, (0.0, 200.0), n_samples=32, sampler=s, ensemble=ens)
                assert got.size == 32 and np.all((got["status"] >= 0) & (got["status"] <= 3))
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    xs, vs, _ = K.sample_position_direction_velocity(m, model, s, 2048)
    got = G.tracegeodesics(m, xs, vs, d, (0.0, 200.0), ensemble=ens)
    ref = oracle.trace(ocfg, xs, vs)
    _compare_points(G, oracle, got, ref)
    assert (ref["status"] == 2).sum() > 200


def test_emissivity_profiles_reproduce_reference_goldens_on_device(G, ens):
    """test/unit/emissivity.jl on the device: the angular point-source method (atol 1e-5 there) and
    the Monte-Carlo profile with deterministic golden-spiral sampling (rtol 1e-2 there)."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    model = G.LampPostModel(h=10.0)
    d = G.ThinDisc(0.0, 500.0)
    prof = G.emissivity_profile(m, d, model, n_samples=20, ensemble=ens)
    gold = np.array([0.0029464479567890534, 0.0014052519492578114, 0.0008963679521766861, 0.0005749351642563003,
                     0.0003386885861792927, 0.0001703542742784169, 6.482839568020104e-5, 1.3029008103481133e-5,
                     3.432060732289487e-6])
    np.testing.assert_allclose(prof.ε, gold, atol=1e-5)
    np.testing.assert_allclose(prof.ε[:4], gold[:4], rtol=4e-3)
    prof = G.emissivity_profile(m, d, model, n_samples=1000, N=10, ensemble=ens,
                                sampler=G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator()))
    gold = np.array([1.4346387869787864, 3.0822515234888774, 1.7923604648828981, 0.6016959946033558,
                     0.11910008907351012, 0.017392602799041507, 0.0023309504405384547, 0.0003139154565507922,
                     3.665392374360994e-5, 1.2069687133228597e-6])
    np.testing.assert_allclose(prof.ε, gold, rtol=1e-2)
    np.testing.assert_allclose(prof.ε[1:], gold[1:], rtol=1e-6)
    # test/smoke-tests/coronal-spectra.jl: the default spectrum is PowerLawSpectrum(2)
    kw = dict(n_samples=10_000, sampler=G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator()), ensemble=ens)
    d100 = G.ThinDisc(0.0, 100.0)
    p1 = G.emissivity_profile(m, d100, model, **kw)
    p2 = G.emissivity_profile(m, d100, model, G.PowerLawSpectrum(2.0), **kw)
    np.testing.assert_array_equal(p1.ε, p2.ε)
    p3 = G.emissivity_profile(m, d100, model, G.PowerLawSpectrum(3.0), **kw)
    ok = np.isfinite(p1.ε) & np.isfinite(p3.ε) & (p1.ε > 0)
    assert ok.sum() > 50 and not np.allclose(p1.ε[ok], p3.ε[ok])
    # test/disc-profiles/test-beamedpointsource.jl: β = 0 is the lamp post
    e0 = G.emissivity_profile(m, d100, model, n_samples=100, ensemble=ens)
    e1 = G.emissivity_profile(m, d100, G.BeamedPointSource(10.0, 0.0), n_samples=100, ensemble=ens)
    radii = np.linspace(2, 100, 10)
    np.testing.assert_allclose(e0.emissivity_at(radii), e1.emissivity_at(radii), rtol=1e-1)


# ---------------- §8 f-4: datum plane and Cunningham transfer functions ----------------
def test_datum_plane_on_device_matches_oracle(G, oracle, ens):
    """DatumPlane(h): signed distance r cosθ - h, hit from above only (datum-plane.jl:1-10)."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    for h, name, params in ((0.0, "kerr", (1.0, 0.998)), (1.5, "kerr", (1.0, 0.998)), (0.0, "johannsen-psaltis", (1.0, 0.6, 2.0))):
        mm = _metric(G, name, params)
        _, _, cache = G.prerendergeodesics(mm, X_FAR, G.DatumPlane(h), 2000.0, image_width=64, image_height=64,
                                           alpha_lims=(-30, 30), beta_lims=(-20, 20), ensemble=ens)
        got = np.ascontiguousarray(cache.points.T).ravel()
        ocfg = oracle.make_config(name, params, disc={"datum": h}, lambda_max=2000.0)
        ref = oracle.trace(ocfg, X_FAR, oracle.render_velocities(ocfg, X_FAR, (-30, 30), (-20, 20), 64, 64))
        # every ray ends on the plane, i.e. at the event root: the two root finders stop within 1e-13 / 1e-12
        # of a step of each other, which is the median here (2e-11) instead of pure rounding
        _compare_points(G, oracle, got, ref, median=1e-10)
        hit = got["status"] == 2
        assert hit.sum() > 3000
        z = got["x"][hit, 1] * np.cos(got["x"][hit, 2])
        assert np.all(z >= h - 1e-12) and np.all(z < h + 1e-7)


def test_cunningham_transfer_functions_on_device(G, oracle, ens):
    """test/smoke-tests/cunningham-transfer-functions.jl:25-39 through the device tracer: six emission
    radii solved in one batch, recorded values to the reference's own tolerance; and a 60-radius table
    costs about as many launches as one radius."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 100_000.0, math.radians(30), 0.0])
    d = G.ThinDisc(0.0, float("inf"))
    chart = G.chart_for_metric(m, 2 * x[1], closest_approach=1.005)
    radii = [7.0, 10.0, 15.0, 300.0, 800.0, 1000.0]
    gold = [0.12205125501900763, 0.1265019201038228, 0.12875961522283233, 0.13378948600255888,
            0.13470290875241375, 0.13319637850028626]
    out = G.cunningham_transfer_functions(m, x, d, radii, N=80, chart=chart, ensemble=ens)
    for c, g, r in zip(out, gold, radii):
        meas = float(np.sum(c.f * c.g_star) / c.f.size)
        assert c.f.size == 114 and np.all(np.isfinite(c.f))
        assert meas == pytest.approx(g, abs=(2e-3 if r < 10 else 1e-3) if r < 100 else 1e-2 * g)
    # same host logic on oracle-traced rays: the two tracers agree far below the statistic's tolerance
    ocfg = oracle.make_config("kerr", (1.0, 0.998), disc={"datum": 0.0}, lambda_max=2 * x[1], closest_approach=1.005,
                              outer_radius=2 * x[1])

    def otrace(al, be):
        # few threads: hundreds of tiny batches, thread start-up would dominate on a many-core host
        pts = oracle.trace(ocfg, x, oracle.map_impact_parameters(ocfg, x, np.asarray(al), np.asarray(be)), nthreads=8)
        return pts, oracle.apply_pf(ocfg, pts, 2 * x[1], pf_id=oracle.PF_REDSHIFT, filter_id=oracle.FILTER_NONE,
                                    r_isco=m.isco(), nthreads=8)

    ref = G.cunningham_transfer_functions(m, x, d, [10.0, 300.0], N=80, tracer=otrace)
    for c, r in zip(ref, (out[1], out[3])):
        assert c.gmin == pytest.approx(r.gmin, rel=1e-7) and c.gmax == pytest.approx(r.gmax, rel=1e-7)
        assert float(np.sum(c.f * c.g_star) / c.f.size) == pytest.approx(float(np.sum(r.f * r.g_star) / r.f.size), abs=3e-4)
    import time

    t0 = time.perf_counter()
    G.cunningham_transfer_functions(m, x, d, [10.0], N=80, chart=chart, ensemble=ens)
    t1 = time.perf_counter()
    table = G.cunningham_transfer_functions(m, x, d, np.linspace(3.0, 50.0, 60), N=80, chart=chart, ensemble=ens)
    t2 = time.perf_counter()
    assert len(table) == 60 and all(np.all(np.isfinite(c.f)) for c in table)
    assert (t2 - t1) < 8 * (t1 - t0)           # 60x the work, far less than 60x the time
    print(f"transfer functions: 1 radius {t1 - t0:.2f} s, 60 radii {t2 - t1:.2f} s")


def test_transfer_function_line_profiles_on_device(G, ens):
    """test/line-profiles/test-cunningham.jl on the device (Kerr a = 0.6 and Johannsen-Psaltis ϵ3 = 2 at
    60°), and the two independent routes to a line profile -- image-plane binning (C5 path) and
    integrated transfer functions -- against each other."""
    ens.set("kernel", 2).set("precision", 64)
    u = np.array([0.0, 1000.0, math.radians(60), 0.0])
    d = G.ThinDisc(0.0, 250.0)
    bins = np.linspace(0.1, 1.3, 100)
    eps = lambda r: r ** -3.0
    for m, lo in ((G.KerrMetric(M=1.0, a=0.6), 0.355), (G.JohannsenPsaltisMetric(1.0, 0.6, 2.0), 0.27)):
        x, y = G.lineprofile(bins, eps, m, u, d, G.TransferFunctionMethod(), N=40, numrₑ=30, ensemble=ens)
        g_low = x[np.argmax(y > 0)]
        g_high = x[len(y) - 1 - np.argmax(y[::-1] > 0) - 1]
        assert g_low == pytest.approx(lo, abs=0.05)
        assert g_high == pytest.approx(1.2, abs=0.05)
        assert y.sum() == pytest.approx(1.0)
    m = G.KerrMetric(M=1.0, a=0.6)
    x, y_tf = G.lineprofile(bins, eps, m, u, d, G.TransferFunctionMethod(), N=80, numrₑ=60, ensemble=ens)
    plane = G.PolarPlane(G.GeometricGrid(), Nr=700, Nθ=1500, r_max=250.0)
    _, y_bin = G.lineprofile(bins, G.PowerLawEmissivity(3), m, u, G.ThinDisc(0.0, 250.0), G.BinningMethod(), plane=plane,
                             minrₑ=m.isco() + 1e-2, maxrₑ=50.0, ensemble=ens)
    # same weighting in the end: the transfer function carries one power of g, the integrand g³ and
    # `_normalize!` takes one back (÷ (g_i + g_{i+1})): ∝ ε g³ dα dβ, as the binning method sums
    l1 = float(np.abs(y_tf - y_bin).sum())
    print(f"line profile, transfer functions vs image-plane binning: L1 = {l1:.4f}, Linf = {np.abs(y_tf - y_bin).max():.5f}")
    assert l1 < 0.05


def test_lagtransfer_on_device(G, ens):
    """test/transfer-functions/test-2d.jl:4-33 on the device: exact intersection counts of both ray sets
    and the binned flux sum."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(M=1.0, a=0.998)
    x = np.array([0.0, 1e6, math.radians(30), 0.0])
    d = G.ThinDisc(m.isco(), 500.0)
    model = G.LampPostModel(h=10.0, θ=math.radians(0.0001))
    tf = G.lagtransfer(m, x, d, model, plane=G.PolarPlane(G.GeometricGrid(), Nr=20, Nθ=20), n_samples=100,
                       sampler=G.EvenSampler(domain=G.BothHemispheres(), generator=G.GoldenSpiralGenerator()), ensemble=ens)
    assert tf.observer_to_disc.size == 337
    assert tf.coronal_geodesics.geodesic_points.size == 58
    t, E, f = G.binflux(tf, N_t=100, N_E=100, ensemble=ens)
    assert float(np.nansum(f)) == pytest.approx(3.9126785201177956, abs=1e-2)
    assert float(np.nansum(f)) == pytest.approx(3.9126785201177956, rel=1e-5)


def test_semi_analytic_lag_transfer_on_device(G, ens):
    """test/transfer-functions/test-2d.jl:35-79 end to end on the device: emissivity profile from 5000 corona
    rays, transfer functions at 5 radii (from the ISCO itself), (g, t) integration; then the
    lag-frequency spectrum."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(M=1.0, a=0.998)
    x = np.array([0.0, 1e6, math.radians(30), 0.0])
    model = G.LampPostModel(h=10.0, θ=math.radians(0.0001))
    prof = G.emissivity_profile(m, G.ThinDisc(m.isco(), 500.0), model, n_samples=5000, ensemble=ens,
                                sampler=G.EvenSampler(domain=G.BothHemispheres(), generator=G.GoldenSpiralGenerator()))
    radii = G.InverseGrid()(m.isco(), 100.0, 5)
    itb = G.transferfunctions(m, x, G.ThinDisc(0.0, 500.0), radii=radii, ensemble=ens)
    bins, tbins = np.linspace(0.0, 1.5, 100), np.linspace(0.0, 150.0, 100)
    flux = G.integrate_lagtransfer(prof, itb, bins, tbins, t0=x[1], n_radii=1000, rmin=min(radii), rmax=max(radii))
    assert float(flux.sum()) == pytest.approx(1.0, abs=1e-2)
    assert float(flux[39, :].sum()) == pytest.approx(0.021759503160585468, abs=1e-4)
    freq, tau = G.lag_frequency(tbins, flux)
    assert np.all(np.isfinite(tau[1:50])) and tau[1] > 0


def test_randomised_scenes_on_device_vs_oracle(G, oracle, ens):
    """84 random scenes through the C ABI against the oracle: metric family and parameters (incl. charged
    test particles in Kerr-Newman), observer radius / inclination, thin disc or datum plane, gtol >= 0.005,
    tolerance, upper-hemisphere callback, window, both kernels.  Same acceptance as the host-compiled
    kernel-logic test (tests/test_kernel_logic_host.py): disc hits and full-λ rays to 1e3·tol, captured /
    out-of-domain rays by status."""
    rng = np.random.default_rng(20251002)
    fam = [
        ("kerr", lambda: (1.0, float(rng.uniform(-0.998, 0.998))), G.KerrMetric),
        ("johannsen", lambda: (1.0, float(rng.uniform(0, 0.9)), float(rng.uniform(-1, 2)), float(rng.uniform(-1, 1)),
                               float(rng.uniform(-1, 1)), float(rng.uniform(-1, 2))), G.JohannsenMetric),
        ("bumblebee", lambda: (1.0, float(rng.uniform(0, 0.29)), float(rng.uniform(-0.5, 1))), G.BumblebeeMetric),
        ("kerr-newman", lambda: (lambda a: (1.0, a, float(rng.uniform(0, math.sqrt(1 - a * a) * 0.95))))(
            float(rng.uniform(0, 0.9))), G.KerrNewmanMetric),
        ("johannsen-psaltis", lambda: (1.0, float(rng.uniform(0, 0.8)), float(rng.uniform(-0.5, 1))),
         G.JohannsenPsaltisMetric),
        ("morris-thorne", lambda: (float(rng.uniform(0.5, 3)),), G.MorrisThorneWormhole),
        ("dilaton-axion", lambda: (1.0, float(rng.uniform(0.1, 0.8)), float(rng.uniform(-0.3, 0.3)), float(rng.uniform(0.3, 1.5))),
         G.DilatonAxion),
    ]
    total_mismatch = total = 0
    for case in range(84):
        name, gen, cls = fam[case % 7] if case >= 24 else fam[0]
        params = gen()
        r_obs = float(10 ** rng.uniform(1.3, 3.2))
        th = float(np.radians(rng.uniform(5, 175)))
        datum = bool(rng.integers(0, 4) == 0) and th < math.pi / 2 - 0.1
        rin = float(rng.uniform(0, 8))
        rout = float(rin + 10 ** rng.uniform(0, 2.3))
        gtol = float(10 ** rng.uniform(-2.3, -1))
        tol = float(rng.choice([1e-9, 1e-7, 1e-5]))
        hemi = bool(rng.integers(0, 2))
        q = float(rng.uniform(-1, 1)) if (name == "kerr-newman" and rng.integers(0, 2)) else 0.0
        lam = float(rng.uniform(1.2, 3) * r_obs)
        lim = float(rng.uniform(5, 60))
        W = H = 16
        m = cls(*params)
        x = np.array([0.0, r_obs, th, 0.0])
        ens.set("kernel", int(rng.integers(0, 2)))
        d = G.DatumPlane(0.0) if datum else G.ThinDisc(rin, rout)
        _, _, cache = G.prerendergeodesics(m, x, d, lam, image_width=W, image_height=H, alpha_lims=(-lim, lim),
                                           beta_lims=(-lim, lim), gtol=gtol, abstol=tol, reltol=tol, q=q, ensemble=ens,
                                           callback=G.domain_upper_hemisphere() if hemi else None)
        got = np.ascontiguousarray(cache.points.T).ravel()
        ocfg = oracle.make_config(name, params, disc=({"datum": 0.0} if datum else (rin, rout)), lambda_max=lam, gtol=gtol,
                                  abstol=tol, reltol=tol, upper_hemisphere=hemi, q=q)
        ref = oracle.trace(ocfg, x, oracle.render_velocities(ocfg, x, (-lim, lim), (-lim, lim), W, H), nthreads=16)
        mism = int((got["status"] != ref["status"]).sum())
        total_mismatch += mism
        total += got.size
        assert mism <= 8, (case, name, params, mism)
        ok = (got["status"] == ref["status"]) & (ref["status"] >= 2) & (ref["flags"] == 0) & (got["flags"] == 0)
        if ok.any():
            scale = np.maximum(np.abs(ref["x"][ok]), 1.0)
            err = np.abs(got["x"][ok] - ref["x"][ok]) / scale
            assert np.sort(err.max(axis=1))[-2 if err.shape[0] > 1 else -1] < max(1e3 * tol, 1e-6), (case, name, params)
    assert total_mismatch <= 0.003 * total
    ens.set("kernel", 2)


def test_polish_doughnut_on_device(G, oracle, ens):
    """A PolishDoughnut as device geometry (its isobar sampled like a ThickDisc) against the oracle tracing the
    same table; the torus casts a shadow band across the image."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.2)
    d = G.PolishDoughnut(m, rₖ=12.0, n=0.21)
    x = np.array([0.0, 1000.0, math.radians(70), 0.0])
    W = H = 96
    _, _, cache = G.prerendergeodesics(m, x, d, 2000.0, image_width=W, image_height=H, alpha_lims=(-25, 25),
                                       beta_lims=(-15, 15), ensemble=ens)
    got = np.ascontiguousarray(cache.points.T).ravel()
    td = d.thick_disc()
    ocfg = oracle.make_config("kerr", (1.0, 0.2), disc={"table": td.table, "range": td.ρ_range}, lambda_max=2000.0)
    ref = oracle.trace(ocfg, x, oracle.render_velocities(ocfg, x, (-25, 25), (-15, 15), W, H))
    _compare_points(G, oracle, got, ref)
    hit = got["status"] == 2
    assert 1000 < hit.sum() < 7000
    ρ = got["x"][hit, 1] * np.abs(np.sin(got["x"][hit, 2]))
    assert ρ.min() >= d.inner_radius - 1e-6 and ρ.max() <= d.outer_radius + 1e-6


def test_reverberation_chain_on_device(G, ens):
    """test/smoke-tests/reverberation.jl:1-45 end to end on the device (continuum_time, emissivity_profile,
    transfer functions with β₀ = 2, integrate_lagtransfer, lag_frequency)."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 10_000.0, math.radians(45), 0.0])
    d = G.ThinDisc(0.0, float("inf"))
    model = G.LampPostModel()
    t0 = G.continuum_time(m, x, model, ensemble=ens)
    assert 10_000.0 < t0 < 10_030.0
    prof = G.emissivity_profile(m, d, model, n_samples=500, ensemble=ens)
    radii = G.InverseGrid()(m.isco(), 100.0, 10)
    itb = G.transferfunctions(m, x, d, radii=radii, β0=2.0, ensemble=ens)
    bins, tbins = np.linspace(0.0, 1.5, 100), np.linspace(0.0, 100.0, 100)
    flux = G.integrate_lagtransfer(prof, itb, bins, tbins, t0=t0, n_radii=100, h=1e-8, rmin=min(radii), rmax=max(radii))
    flux[flux == 0] = np.nan
    freq, tau = G.lag_frequency(tbins, flux)
    assert float(freq.sum()) == pytest.approx(2449.8787687490535, rel=1e-2)
    assert float(tau[131]) == pytest.approx(9.322742661315855, rel=1e-2)


def test_ring_corona_traces_and_illuminates_the_disc(G, ens):
    """An off-axis, co-rotating source through the generic Monte-Carlo route (per-sample source tetrad):
    rays reach the disc on both sides of the ring's footprint and the emissivity peaks under the ring."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.9)
    d = G.ThinDisc(m.isco(), 200.0)
    model = G.RingCorona(G.SourceVelocities.co_rotating, 8.0, 3.0)
    s = G.EvenSampler(G.BothHemispheres(), G.GoldenSpiralGenerator())
    cg = G.tracecorona(m, d, model, n_samples=600, sampler=s, ensemble=ens)
    rho = cg.geodesic_points["x"][:, 1] * np.abs(np.sin(cg.geodesic_points["x"][:, 2]))
    assert 200 < rho.size < 500 and rho.min() < 6.0 and rho.max() > 30.0
    prof = G.emissivity_profile(m, d, model, sampler=s, n_samples=600, N=12, ensemble=ens)
    ok = np.isfinite(prof.ε) & (prof.ε > 0)
    assert ok.sum() >= 8
    peak = prof.radii[ok][np.argmax(prof.ε[ok])]
    assert 3.0 < peak < 16.0                                   # brightest under the ring (r = 8)


def test_batched_saved_paths(G, oracle, ens):
    """tracegeodesics(m, xs, vs, ...) with every step saved (gr_trace_paths): ray by ray identical to the
    single-geodesic entry point, last row == end-point record, step counts == the oracle's."""
    m = G.KerrMetric(1.0, 0.7)
    x = np.array([0.0, 300.0, math.radians(65), 0.0])
    al, be = np.meshgrid(np.linspace(-9, 9, 12), np.linspace(-9, 9, 11))
    vs = G.map_impact_parameters(m, x, al.ravel(), be.ravel())
    d = G.ThinDisc(m.isco(), 30.0)
    paths = G.tracegeodesic_paths(m, x, vs, d, 600.0, cap=64, ensemble=ens)        # cap too small on purpose: retried
    assert len(paths) == vs.shape[0]
    ends = G.tracegeodesics(m, x, vs, d, 600.0, ensemble=ens)
    ocfg = oracle.make_config("kerr", (1.0, 0.7), disc=(m.isco(), 30.0), lambda_max=600.0)
    for j in (0, 17, 66, 131):
        one = G.tracegeodesic_path(m, x, vs[j], d, 600.0, ensemble=ens)
        assert one.λ.tobytes() == paths[j].λ.tobytes() and one.x.tobytes() == paths[j].x.tobytes()
        assert one.v.tobytes() == paths[j].v.tobytes()
        ref_pt, st = oracle.trace(ocfg, x, vs[j:j + 1], stats=True)
        assert abs((paths[j].λ.size - 2) - int(st["accepted"][0])) <= 2
    for j, p in enumerate(paths):
        assert p.point["status"] == ends["status"][j]
        # the path kernel and the end-point kernel are two instantiations of one source, built with -ffp-contract=on: the
        # same roundings, bit for bit
        np.testing.assert_array_equal(p.x[-1], ends["x"][j])
        np.testing.assert_array_equal(p.v[-1], ends["v"][j])
        assert p.λ[0] == 0.0 and np.all(np.diff(p.λ) > 0) and p.λ[-1] == ends["lambda_max"][j]
    # per-ray start positions
    xs = np.tile(x, (vs.shape[0], 1))
    xs[:, 1] += np.arange(vs.shape[0]) * 0.01
    p2 = G.tracegeodesic_paths(m, xs, vs, d, 600.0, ensemble=ens)
    assert p2[5].x[0, 1] == xs[5, 1] and p2[0].x.tobytes() == paths[0].x.tobytes()


def test_ray_summary_equals_endpoints_plus_redshift(G, ens):
    """gr_ray_summary (one launch, 32 B per ray) against the two-launch route it replaces in the
    transfer-function solver: end-point records + gr_apply_pointfunction."""
    from gradus_jl_amd.rendering import apply_pointfunction
    from gradus_jl_amd.transfer_functions import device_tracer

    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.9)
    x = np.array([0.0, 1000.0, math.radians(50), 0.0])
    pf = G.ConstPointFunctions.redshift(m, x)
    rng = np.random.default_rng(5)
    al, be = rng.uniform(-25, 25, 4000), rng.uniform(-25, 25, 4000)
    trace = device_tracer(m, x, 2000.0, G.chart_for_metric(m, 2000.0), pf, ens)
    pts, g = trace(al, be)
    v = G.map_impact_parameters(m, x, al, be)
    cfg = G.tracing_configuration(m, x, v, G.DatumPlane(0.0), 2000.0, chart=G.chart_for_metric(m, 2000.0), ensemble=ens)
    ref = G.ensemble_solve_tracing_problem(ens, cfg)
    gref = apply_pointfunction(ens, cfg, pf, ref, 2000.0)
    np.testing.assert_array_equal(pts["status"], ref["status"])
    hit = ref["status"] == 2
    assert 3000 < hit.sum() < 4000
    # (the summary path maps impact parameters to velocities on the device, the other one on the host:
    # initial conditions agree to rounding, results to the integration tolerance)
    np.testing.assert_allclose(pts["x"][hit, 0], ref["x"][hit, 0], rtol=1e-8)
    np.testing.assert_allclose(pts["x"][hit, 1], ref["x"][hit, 1] * np.abs(np.sin(ref["x"][hit, 2])), rtol=1e-7)
    np.testing.assert_allclose(g[hit], gref[hit], rtol=1e-7)
    assert np.all(np.isnan(g[~hit]))


def test_rayset_endpoints_with_one_datum_plane_per_ray(G, oracle, ens):
    """gr_rayset_endpoints with gr_rayset.height: every ray against its own DatumPlane (datumplane(d, rₑ),
    datum-plane.jl:14-17) in one launch == the oracle run once per plane; without heights == the
    plane of the configuration; and the summaries of the same rays carry the same (ρ, t, status)."""
    from gradus_jl_amd.transfer_functions import device_tracer

    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, 0.9)
    x = np.array([0.0, 1000.0, math.radians(70), 0.0])
    rng = np.random.default_rng(5)
    n = 600
    al, be = rng.uniform(-25, 25, n), rng.uniform(-20, 20, n)
    hs = rng.choice([0.0, 0.4, 1.1, 2.5], n)
    pf = G.ConstPointFunctions.redshift(m, x)
    trace = device_tracer(m, x, 2000.0, G.chart_for_metric(m, 2000.0), pf, ens)
    got = trace.endpoints(al, be, hs)
    for h in np.unique(hs):
        I = hs == h
        ocfg = oracle.make_config("kerr", (1.0, 0.9), disc={"datum": float(h)}, lambda_max=2000.0, outer_radius=2000.0)
        ref = oracle.trace(ocfg, x, oracle.map_impact_parameters(ocfg, x, al[I], be[I]))
        _compare_points(G, oracle, got[I], ref, median=1e-10)
        hit = got["status"][I] == 2
        z = got["x"][I][hit, 1] * np.cos(got["x"][I][hit, 2])
        assert hit.sum() > 50 and np.all(np.abs(z - h) < 1e-7)
    plain = trace.endpoints(al, be)
    zero = trace.endpoints(al, be, np.zeros(n))
    assert plain.tobytes() == zero.tobytes()
    summ, g = trace(al, be, hs)
    np.testing.assert_array_equal(summ["status"], got["status"])
    np.testing.assert_allclose(summ["x"][:, 0], got["x"][:, 0], rtol=1e-13)
    np.testing.assert_allclose(summ["x"][:, 1], got["x"][:, 1] * np.abs(np.sin(got["x"][:, 2])), rtol=1e-13)
    # empty set
    assert trace.endpoints(np.zeros(0), np.zeros(0), np.zeros(0)).size == 0


THICK_GOLD = [(0.998, 75, 3.0, 0.3, 14.64279128586961, 1e-4), (0.2, 20, 5.469668466100368, 0.2, 21.581370829241525, 1e-2)]


@pytest.mark.parametrize("a,angle,r_e,edd,gold,atol", [
    pytest.param(*THICK_GOLD[0], id="a0.998-75deg", marks=pytest.mark.xfail(
        strict=True, reason="recorded thick-disc sum 14.64279 not reproduced at the reference's atol 1e-4: 14.64494 here (f-4)")),
    pytest.param(*THICK_GOLD[1], id="a0.2-20deg", marks=pytest.mark.xfail(
        strict=True, reason="recorded thick-disc sum 21.5814 not reproduced at the reference's atol 1e-2: 21.4028 here (f-4)")),
])
def test_thick_disc_recorded_sums_at_the_reference_tolerance(G, ens, a, angle, r_e, edd, gold, atol):
    """test/transfer-functions/test-thick-disc.jl:9-11,17-19 at the tolerances the REFERENCE asserts.  Not met (2.2e-3 and 0.18
    absolute: sums over 114 samples, i.e. a sample-set question like the three thin-disc records of tests/test_gpu_tangent.py):
    strict xfails, so that every GPU test record shows 5 unmet of the 13 recorded transfer-function statistics.  The looser
    bounds of the next test are this build's own band around the record, not parity."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(1.0, a)
    x = np.array([0.0, 10_000.0, math.radians(angle), 0.0])
    d = G.ShakuraSunyaev.for_metric(m, eddington_ratio=edd)
    tf = G.cunningham_transfer_function(m, x, d, r_e, β0=2.0, ensemble=ens)
    total = float(np.nansum(tf.f))
    print(f"  thick disc a={a} {angle}°: {total:.6f} vs {gold:.6f} ({total - gold:+.2e}, atol {atol:g})")
    assert total == pytest.approx(gold, abs=atol)


def test_thick_disc_transfer_functions_on_device(G, oracle, ens):
    """test/transfer-functions/test-thick-disc.jl through the device tracer: the datum-plane offsets
    (one plane per emission radius), the visibility re-trace against the ShakuraSunyaev surface and the
    thick-surface Jacobians, several radii in the same launches.  The recorded sums are NOT met at the reference's tolerances
    (strict xfails above); here they are held to this build's own band (5e-4 / 1e-2 relative) so that a regression shows,
    and the well-conditioned samples agree with the same host logic driven by oracle-traced rays."""
    import sys, os

    sys.path.insert(0, os.path.dirname(__file__))
    from test_transfer_functions_host import thick_oracle_tracers

    ens.set("kernel", 2).set("precision", 64)
    # with dual-number Jacobians the device gives 14.64494 / 21.4028 at EVERY tolerance from 1e-9 to 1e-12
    # (scripts/thick_tf_sums.py): 1.5e-4 and -8.3e-3 relative to the record (the second: one sample's worth of 114)
    for a, angle, r_e, edd, gold, tol in ((0.998, 75, 3.0, 0.3, 14.64279128586961, 5e-4),
                                          (0.2, 20, 5.469668466100368, 0.2, 21.581370829241525, 1e-2)):
        m = G.KerrMetric(1.0, a)
        x = np.array([0.0, 10_000.0, math.radians(angle), 0.0])
        d = G.ShakuraSunyaev.for_metric(m, eddington_ratio=edd)
        tf = G.cunningham_transfer_function(m, x, d, r_e, β0=2.0, ensemble=ens)
        assert float(np.nansum(tf.f)) == pytest.approx(gold, rel=tol)
        datum, thick, jac = thick_oracle_tracers(G, oracle, m, a, x, d, 2 * x[1])
        ref = G.transfer_functions.cunningham_transfer_functions(m, x, d, [r_e], tracer=datum, thick_tracers=(thick, jac), β0=2.0)[0]
        assert tf.gmin == pytest.approx(ref.gmin, rel=1e-7) and tf.gmax == pytest.approx(ref.gmax, rel=1e-7)
        core = np.isfinite(ref.f) & (ref.g_star > 1e-3) & (ref.g_star < 1 - 1e-3)
        assert core.sum() > 70 and np.array_equal(np.isfinite(tf.f)[core], np.isfinite(ref.f)[core])
        np.testing.assert_allclose(tf.g_star[core], ref.g_star[core], atol=1e-6)
        # the device's Jacobians are dual numbers through the integrator (gr_ray_tangent); the oracle-driven ones are
        # difference quotients, whose truncation error on the curved ShakuraSunyaev surface reaches 8 % where the
        # determinant nearly cancels (host check: AD at 1e-9 and at 1e-12 agree to 1e-5 on every core sample, the
        # difference quotients stay 8 % off at both; AD against small-step differences at 1e-12: det ratio 1.00000)
        rel = np.abs(tf.f[core] - ref.f[core]) / ref.f[core]
        assert np.median(rel) < 2e-4 and np.percentile(rel, 85) < 2e-3 and rel.max() < 0.1
    # a table of radii in one batch: same values as one at a time
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 10_000.0, math.radians(60), 0.0])
    d = G.ShakuraSunyaev.for_metric(m)
    radii = [3.0, 6.0, 12.0, 40.0]
    table = G.cunningham_transfer_functions(m, x, d, radii, β0=1.0, ensemble=ens)
    single = G.cunningham_transfer_function(m, x, d, 12.0, β0=1.0, ensemble=ens)
    core = np.isfinite(single.f) & (single.g_star > 1e-3) & (single.g_star < 1 - 1e-3)
    np.testing.assert_allclose(table[2].f[core], single.f[core], rtol=1e-6)
    assert all(np.isfinite(c.f).sum() > 60 for c in table)


def test_precision_solvers_on_device(G, ens):
    """src/tracing/precision-solvers.jl on the device tracer: the recorded impact parameters of
    test/integration/test-precision.jl (rtol 1e-3 there), rings of impact parameters for a radius, and
    the obscured variant for a thick disc."""
    ens.set("kernel", 2).set("precision", 64)
    m = G.KerrMetric(M=1.0, a=1.0)
    u = np.array([0.0, 1000.0, math.pi / 2, 0.0])
    α, β, acc = G.impact_parameters_for_target(np.array([10.0, 0.005, 0.0]), m, u, ensemble=ens)
    # the reference's Nelder-Mead stops 4.6e-3 away from this target (its recorded `accuracy`), so its
    # (α, β) are only good to about that: |α| itself is 4e-3.  The ray found here passes through the target.
    assert acc < 1e-6
    assert α == pytest.approx(-0.004013630261097743, abs=4.6e-3) and β == pytest.approx(10.969606493445841, rel=2e-3)
    target = np.array([10.0, math.radians(40), -math.pi / 4])
    α, β, gp, acc = G.optimize_for_target(target, m, u, ensemble=ens)
    assert α == pytest.approx(4.848373364532467, rel=1e-3) and β == pytest.approx(8.02066263349774, rel=1e-3)
    assert gp["x"][0] == pytest.approx(1005.2700874611182, rel=1e-3) and acc < 1e-6
    np.testing.assert_allclose(gp["x"][1:3], target[:2], rtol=1e-6)
    assert math.remainder(gp["x"][3] - target[2], 2 * math.pi) == pytest.approx(0.0, abs=1e-6)

    # a ring on a thin disc: every ray lands on the radius asked for
    m = G.KerrMetric(1.0, 0.998)
    x = np.array([0.0, 1000.0, math.radians(70), 0.0])
    d = G.ThinDisc(0.0, float("inf"))
    a, b = G.impact_parameters_for_radius(m, x, d, 8.0, N=64, ensemble=ens)
    assert np.all(np.isfinite(a)) and a[0] == pytest.approx(a[-1], abs=1e-9)
    pts = G.tracegeodesics(m, x, G.map_impact_parameters(m, x, a, b), G.DatumPlane(0.0), 2000.0, ensemble=ens)
    assert np.all(pts["status"] == 2)
    np.testing.assert_allclose(pts["x"][:, 1] * np.sin(pts["x"][:, 2]), 8.0, atol=2e-7)
    assert G.find_offset_for_radius(m, x, d, 8.0, 0.3, ensemble=ens) == pytest.approx(float(np.hypot(a[3], b[3])), rel=0.2)
    J = G.jacobian_αβ_gr(m, x, d, a[5], b[5], ensemble=ens)
    assert np.isfinite(J) and J > 0

    # thick disc: the far side of an inner ring is hidden behind the disc at high inclination
    ss = G.ShakuraSunyaev.for_metric(m, eddington_ratio=0.3)
    x = np.array([0.0, 1000.0, math.radians(80), 0.0])
    # (the ring's image is lifted off the image-plane origin by the plane's height: centre the angles on β₀ as
    # the reference's thick-disc tests do)
    a0, b0 = G.impact_parameters_for_radius(m, x, ss, 4.0, N=90, β0=2.0, ensemble=ens)
    a1, b1 = G.impact_parameters_for_radius_obscured(m, x, ss, 4.0, N=90, β0=2.0, ensemble=ens)
    vis = np.isfinite(a1)
    print("ring:", np.isfinite(a0).sum(), "offsets,", vis.sum(), "visible")
    assert np.isfinite(a0).sum() > 80 and 10 < vis.sum() < np.isfinite(a0).sum()
    np.testing.assert_array_equal(a1[vis], a0[vis])
    # the visible rays end on the disc surface at the ring's radius and height
    pts = G.tracegeodesics(m, x, G.map_impact_parameters(m, x, a1[vis], b1[vis]), ss, 2000.0, ensemble=ens)
    ρ = pts["x"][:, 1] * np.sin(pts["x"][:, 2])
    np.testing.assert_allclose(ρ, 4.0, atol=1e-5)
    np.testing.assert_allclose(pts["x"][:, 1] * np.cos(pts["x"][:, 2]), ss.cross_section(4.0), atol=1e-5)


@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("name,params,cls", [
    ("spherical", (), "SphericalMetric"),
    ("kerr-dark-matter", (1.0, 0.6, 2.0, 20.0, 10.0), "KerrDarkMatter"),
    ("kerr-refractive", (1.0, 0.6, 1.2, 20.0), "KerrRefractive"),
    ("noz", (1.0, 0.7, 0.5), "NoZMetric"),
])
def test_remaining_metrics_on_device_vs_oracle(G, oracle, ens, kernel, name, params, cls):
    """SphericalMetric, KerrDarkMatter, KerrRefractive, NoZMetric through the C ABI (generic dual-number
    functor) against the oracle, end points of a 48 x 48 plane with a thin disc; the Kerr limits of
    the three deformations reproduce the Kerr kernel's own end points."""
    ens.set("kernel", kernel).set("precision", 64)
    m = getattr(G, cls)(*params)
    x = np.array([0.0, 200.0, math.radians(70), 0.0])
    W = H = 48
    disc = (3.0, 60.0)
    _, _, cache = G.prerendergeodesics(m, x, G.ThinDisc(*disc), 500.0, image_width=W, image_height=H, alpha_lims=(-40, 40),
                                       beta_lims=(-30, 30), ensemble=ens)
    got = np.ascontiguousarray(cache.points.T).ravel()
    ocfg = oracle.make_config(name, params, disc=disc, lambda_max=500.0)
    ref = oracle.trace(ocfg, x, oracle.render_velocities(ocfg, x, (-40, 40), (-30, 30), W, H))
    mism = got["status"] != ref["status"]
    # flat space is integrated in a handful of huge steps: whether one of the reference's 8 samples per step
    # lands inside the disc's thin wedge is then decided by the last bits of the step sizes
    # kerr-refractive: the 2.5e-4-wide index step at the corona radius is not resolved at tolerance 1e-9 by anyone: the
    # oracle flips 2-3 of these 2304 rays against itself when its tolerance is nudged by 10 %, the device 6-8 against it
    assert mism.sum() <= {"spherical": 24, "kerr-refractive": 12}.get(name, 6)
    ok = ~mism & (ref["status"] != oracle.WITHIN_INNER_BOUNDARY)
    assert (ref["status"][ok] == 2).sum() > 200
    # kerr-refractive: see tests/test_kernel_logic_host.py; kerr-dark-matter: the enclosed mass is only C¹ at rₛ and
    # rₛ + Δr, so the step that straddles either radius is not controlled to tolerance (worst ray 1.6e-6)
    tol = {"kerr-refractive": 2e-4, "kerr-dark-matter": 1e-5}.get(name, RTOL)
    np.testing.assert_allclose(got["lambda_max"][ok], ref["lambda_max"][ok], rtol=tol)
    scale = np.maximum(np.abs(ref["x"][ok]), 1.0)
    assert np.max(np.abs(got["x"][ok] - ref["x"][ok]) / scale) < tol
    if name != "spherical" and kernel == 0:
        lim = {"kerr-dark-matter": (1.0, 0.6, 0.0, 20.0, 10.0), "kerr-refractive": (1.0, 0.6, 1.0, 20.0), "noz": (1.0, 0.6, 0.0)}[name]
        kw = dict(image_width=32, image_height=32, alpha_lims=(-40, 40), beta_lims=(-30, 30), ensemble=ens)
        _, _, a = G.prerendergeodesics(getattr(G, cls)(*lim), x, G.ThinDisc(*disc), 500.0, **kw)
        _, _, b = G.prerendergeodesics(G.KerrMetric(1.0, 0.6), x, G.ThinDisc(*disc), 500.0, **kw)
        a, b = a.points.ravel(), b.points.ravel()
        same = (a["status"] == b["status"])
        assert (~same).sum() <= 2
        keep = same & (b["status"] != 1)
        np.testing.assert_allclose(a["x"][keep], b["x"][keep], rtol=1e-6, atol=1e-8)
    ens.set("kernel", 2)


@pytest.mark.parametrize("kernel", [0, 1])
def test_elliptical_and_precessing_discs_on_device(G, oracle, ens, kernel):
    """EllipticalDisc and PrecessingDisc(ThinDisc, β, γ) (src/geometry/discs.jl:57-96) through the C ABI against the
    oracle, two metric functors each; the untilted precessing disc is the thin disc bit for bit in status and to
    rounding in position."""
    ens.set("kernel", kernel).set("precision", 64)
    x = np.array([0.0, 300.0, math.radians(65), 0.0])
    W = H = 64
    kw = dict(image_width=W, image_height=H, alpha_lims=(-45, 45), beta_lims=(-35, 35), ensemble=ens)
    cases = [(G.EllipticalDisc(2.0, 30.0, 4.0), {"ellipse": (2.0, 30.0, 4.0)}),
             (G.PrecessingDisc(G.ThinDisc(3.0, 40.0), 0.35, 0.8), {"precessing": (3.0, 40.0, 0.35, 0.8)})]
    for name, params in (("kerr", (1.0, 0.9)), ("johannsen-psaltis", (1.0, 0.5, 0.8))):
        m = _metric(G, name, params)
        for d, od in cases:
            _, _, cache = G.prerendergeodesics(m, x, d, 700.0, **kw)
            got = np.ascontiguousarray(cache.points.T).ravel()
            ocfg = oracle.make_config(name, params, disc=od, lambda_max=700.0)
            ref = oracle.trace(ocfg, x, oracle.render_velocities(ocfg, x, (-45, 45), (-35, 35), W, H))
            _compare_points(G, oracle, got, ref, median=1e-10)
            assert (got["status"] == 2).sum() > 600
    m = G.KerrMetric(1.0, 0.9)
    _, _, a = G.prerendergeodesics(m, x, G.PrecessingDisc(G.ThinDisc(3.0, 40.0), 0.0, 0.0), 700.0, **kw)
    _, _, b = G.prerendergeodesics(m, x, G.ThinDisc(3.0, 40.0), 700.0, **kw)
    np.testing.assert_array_equal(a.points["status"], b.points["status"])
    np.testing.assert_allclose(a.points["x"], b.points["x"], rtol=1e-9, atol=1e-12)
    # a redshift image of the tilted disc renders through the fused path as well
    pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
    _, _, img = G.rendergeodesics(m, x, G.PrecessingDisc(G.ThinDisc(m.isco(), 40.0), 0.35, 0.8), 700.0, pf=pf, **kw)
    assert np.isfinite(img).sum() > 600 and 0.2 < np.nanmin(img) and np.nanmax(img) < 1.6
    ens.set("kernel", 2)


def test_contexts_release_their_device_memory(G):
    """gr_ctx_create / gr_ctx_destroy in a loop with calls of growing size in between: the context's staging
    buffers (scratch, inputs, tables, cold blocks) are freed with it -- free HBM returns to where it was."""
    import torch

    m = G.KerrMetric(1.0, 0.9)
    d = G.ThinDisc(m.isco(), 30.0)

    def cycle(size):
        ens = G.EnsembleMI355X(0)
        G.prerendergeodesics(m, X_SMOKE, d, 200.0, image_width=size, image_height=size, alpha_lims=(-25, 25),
                             beta_lims=(-15, 15), ensemble=ens)
        G.rendergeodesics(G.JohannsenMetric(1.0, 0.5, 0.3, 0.0, 0.0, 0.0), X_SMOKE, d, 200.0, image_width=size,
                          image_height=size, alpha_lims=(-25, 25), beta_lims=(-15, 15), ensemble=ens)
        for c in ens.contexts:
            c.close()

    cycle(64)
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info(0)
    for k in range(40):
        cycle(64 + 16 * (k % 8))
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info(0)
    assert free0 - free1 < 64 << 20, f"leaked {(free0 - free1) >> 20} MiB over 40 contexts"


def test_warped_thin_disc_on_device(G, oracle, ens):
    """WarpedThinDisc(f) through the C ABI (tabulated geometry with the signed-height flag) against the oracle."""
    ens.set("kernel", 2).set("precision", 64)
    d = G.WarpedThinDisc(lambda ρ: 0.8 * math.sin(ρ / 6.0), inner_radius=3.0, outer_radius=45.0, samples=4096)
    x = np.array([0.0, 300.0, math.radians(65), 0.0])
    for name, params in (("kerr", (1.0, 0.9)), ("johannsen", (1.0, 0.6, 0.5, 0.0, 0.0, 0.3))):
        m = _metric(G, name, params)
        for kernel in (0, 1):
            ens.set("kernel", kernel)
            _, _, cache = G.prerendergeodesics(m, x, d, 700.0, image_width=64, image_height=64, alpha_lims=(-50, 50),
                                               beta_lims=(-35, 35), ensemble=ens)
            got = np.ascontiguousarray(cache.points.T).ravel()
            ocfg = oracle.make_config(name, params, disc={"table": d.table, "range": d.ρ_range, "warped": True}, lambda_max=700.0)
            ref = oracle.trace(ocfg, x, oracle.render_velocities(ocfg, x, (-50, 50), (-35, 35), 64, 64))
            _compare_points(G, oracle, got, ref, median=1e-10)
            assert (got["status"] == 2).sum() > 1200
    ens.set("kernel", 2)


def test_circular_orbits_found_by_tracing(G, ens):
    """test/smoke-tests/circular-orbits.jl:5-22: Σ vϕ over r = 6:0.5:10 of the orbits found by the OPTIMISER
    (golden section on the radial excursion of device-traced μ = 1 geodesics, all radii in lock-step) against the
    recorded values (atol 1e-6 there), and against CircularOrbits' closed form per radius."""
    ens.set("kernel", 2).set("precision", 64)
    rs = np.arange(6.0, 10.0 + 1e-9, 0.5)
    for m, expected in ((G.KerrMetric(M=1.0, a=0.0), 0.5432533297869712), (G.KerrMetric(M=1.0, a=1.0), 0.5016710246454921),
                        (G.KerrMetric(M=1.0, a=-1.0), 0.5993458160081419),
                        (G.JohannsenMetric(M=1.0, a=1.0, alpha22=1.0), 0.4980454719932759)):
        vϕ = G.solve_equatorial_circular_orbit(m, rs, ensemble=ens)
        assert float(np.sum(vϕ)) == pytest.approx(expected, abs=2e-6)
        np.testing.assert_allclose(vϕ, G.CircularOrbits.fourvelocity(m, rs)[:, 3], atol=1e-6)
    path = G.trace_equatorial_circular_orbit(G.KerrMetric(1.0, 0.5), 7.0, ensemble=ens)
    assert np.ptp(path.x[:, 1]) < 1e-4 and path.x[-1, 3] > 2 * math.pi


def test_transfer_function_table_on_device(G, ens):
    """make_transfer_function_table (cunningham-transfer-functions.jl:503-530): a 2 x 2 lattice in (spin, inclination)
    of 24-radius grids, each lattice point one batch on the device; the table interpolates between them."""
    import time

    ens.set("kernel", 2).set("precision", 64)
    t0 = time.perf_counter()
    table = G.make_transfer_function_table(G.KerrMetric, G.ThinDisc(0.0, float("inf")), [0.5, 0.9], [30.0, 60.0], r_max=200.0,
                                           n_radii=24, ensemble=ens)
    dt = time.perf_counter() - t0
    assert table.grids.shape == (2, 2)
    for g in table.grids.ravel():
        assert g.lower_f.shape == (20, 24) and np.all(np.isfinite(g.lower_f)) and np.all(np.isfinite(g.upper_time))
        assert np.all(np.diff(g.r_grid) > 0) and np.all(g.g_min < g.g_max)
    # higher inclination -> broader range of redshifts at every radius
    assert np.all(table.grids[1, 1].g_max - table.grids[1, 1].g_min > table.grids[1, 0].g_max - table.grids[1, 0].g_min)
    mid = table(0.7, 45.0)
    assert mid.lower_f.shape == (20, 24) and np.all(np.isfinite(mid.upper_f))
    lo, hi = np.minimum(table.grids[0, 0].g_max, table.grids[1, 1].g_max), np.maximum(table.grids[0, 1].g_max, table.grids[1, 0].g_max)
    assert np.all(mid.g_max >= np.minimum(lo, table.grids[0, 0].g_max) - 1e-12) and np.all(mid.g_max <= hi + 0.2)
    print(f"transfer-function table: 4 lattice points x 24 radii in {dt:.2f} s")


@pytest.mark.parametrize("kernel", [0, 1])
def test_trace_windings_on_device(G, oracle, ens, kernel):
    """TraceWindings through the C ABI: the winding number of every ray (bits 16..31 of gr_point.flags and the fused
    GR_PF_WINDING image) against the oracle; a plain trace leaves those bits zero; `apply` of the winding point
    function on an end-point cache equals the fused image."""
    ens.set("kernel", kernel).set("precision", 64)
    m = G.KerrMetric(1.0, 0.9)
    x = np.array([0.0, 200.0, math.radians(80), 0.0])
    W = H = 96
    kw = dict(image_width=W, image_height=H, alpha_lims=(-8, 8), beta_lims=(-8, 8), ensemble=ens)
    _, _, cache = G.prerendergeodesics(m, x, 400.0, trace=G.TraceWindings(), **kw)
    got = np.ascontiguousarray(cache.points.T).ravel()
    ocfg = oracle.make_config("kerr", (1.0, 0.9), lambda_max=400.0, winding_plane=math.pi / 2)
    ref = oracle.trace(ocfg, x, oracle.render_velocities(ocfg, x, (-8, 8), (-8, 8), W, H))
    same = got["status"] == ref["status"]
    assert (~same).sum() <= 4
    wg, wr = G.winding_number(got), G.winding_number(ref)
    assert (wg[same] != wr[same]).sum() <= 6 and wg.max() >= 3
    _, _, img = G.rendergeodesics(m, x, 400.0, pf=G.ConstPointFunctions.winding(), trace=G.TraceWindings(), **kw)
    np.testing.assert_array_equal(img.T.ravel(), wg.astype(float))
    np.testing.assert_array_equal(G.apply(G.ConstPointFunctions.winding(), cache).T.ravel(), wg.astype(float))
    _, _, plain = G.prerendergeodesics(m, x, 400.0, **kw)
    assert np.all(plain.points["flags"] == 0)
    ens.set("kernel", 2)


def test_plain_c_client_renders_the_reference_fingerprint(G, tmp_path):
    """tests/c/c_abi_smoke.c: a C11 program that dlopen's the library and renders the reference's 20 x 20 shadow
    through gr_render -- the boundary is usable without Python or torch."""
    import subprocess
    import sys, os

    sys.path.insert(0, os.path.dirname(__file__))
    from test_host_api import _build_c_client

    exe = _build_c_client(tmp_path)
    rc = subprocess.run([exe, G._lib.LIB_PATH], capture_output=True, text=True)
    print(rc.stdout)
    assert rc.returncode == 0, rc.stdout + rc.stderr
    assert "fingerprint 9009.4" in rc.stdout          # the program itself checks it to 1e-6


def test_bench_collective_path_on_one_gpu():
    """bench.py under torch.distributed.run with one rank pushed through the RCCL gather (GRADUS_FORCE_COLLECTIVE=1) and
    two renders in flight -- the stream hand-over between the trace kernels, the collective and the assembly that
    the multi-GPU runs rely on -- produces a valid line and the same throughput class as the plain run."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRADUS_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29561", os.path.join(root, "bench.py"), "--gpus", "1", "--size", "1024", "--steps", "6", "--warmup", "2",
           "--streams", "2", "--no-cpu-baseline"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["renders_in_flight"] == 2 and line["value"] > 5e7
    # launches overlapped pairwise: a launch's own span is longer than the device's busy span per launch (by 1.9x when
    # the Kerr kernel held 2 waves per SIMD; at 3 waves per SIMD a single launch leaves less room beside it: 1.28x)
    assert line["roofline"]["launch_ms"] > 1.15 * line["roofline"]["kernel_ms"]


def test_sharded_lineprofile_script_through_rccl_on_one_gpu():
    """scripts/lineprofile_sharded.py under torch.distributed.run with one rank pushed through the RCCL all-reduce of the
    histogram (GRADUS_FORCE_COLLECTIVE=1): the launch line an 8-GPU node would use, at 1024² rays."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, GRADUS_FORCE_COLLECTIVE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr", "127.0.0.1",
           "--master-port", "29571", os.path.join(root, "scripts", "lineprofile_sharded.py"), "--size", "1024", "--steps", "4",
           "--warmup", "1"]
    out = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    line = json.loads([l for l in out.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["rccl_ranks"] == 1 and line["profile_sum"] == pytest.approx(1.0, abs=1e-12)
    assert line["rays_per_s"] > 5e7


def test_c_abi_rejects_bad_input_without_touching_the_device(G, ens):
    """Error behaviour of the boundary: configuration mistakes come back as negative codes with a
    message -- unknown metric / geometry / point-function ids, null pointers, negative counts, unsorted limits --
    and nothing throws or aborts; a valid call afterwards still works."""
    import ctypes as C

    L = G._lib.load()
    h = ens.ctx.handle
    m = G.KerrMetric(1.0, 0.5)
    cfgo = G.render_configuration(m, X_SMOKE, G.ThinDisc(2.0, 30.0), 200.0, image_width=16, image_height=16, alpha_lims=(-9, 9),
                                  beta_lims=(-9, 9), ensemble=ens)
    cfg, pl = cfgo.abi_config(), cfgo.abi_plane()
    rg = G._lib.gr_range(0, 256, 256, 1)
    img = np.zeros(256)
    pf = G._lib.gr_pointfunction()
    pf.pf_id, pf.filter_id, pf.fill = 0, 0, float("nan")

    def call(cfg_=cfg, pl_=pl, pf_=pf, rg_=rg, img_=img):
        return L.gr_render(h, C.byref(cfg_), C.byref(pl_), C.byref(pf_), C.byref(rg_), img_.ctypes.data if img_ is not None else None, None)

    assert call() == 0
    for field, value, code in (("metric_id", 12, -2), ("metric_id", 11, -1), ("metric_id", -1, -2), ("disc_id", 9, -2), ("disc_id", 8, -1), ("disc_id", 7, -1), ("abstol", -1.0, -1)):
        bad = type(cfg).from_buffer_copy(cfg)
        setattr(bad, field, value)
        assert call(cfg_=bad) == code, field
        assert len(L.gr_last_error()) > 0
    badpf = type(pf).from_buffer_copy(pf)
    badpf.pf_id = 9
    assert call(pf_=badpf) == -2
    badpl = type(pl).from_buffer_copy(pl)
    badpl.alpha0, badpl.alpha1 = 5.0, -5.0
    assert call(pl_=badpl) == -1
    assert call(img_=None) == -1
    rs = G._lib.gr_rayset()
    rs.n = -3
    assert L.gr_rayset_endpoints(h, C.byref(cfg), C.byref(rs), None, None) == -1
    rs.n = 4                                    # alpha / beta missing
    pts = np.zeros(4, dtype=G.POINT_DTYPE)
    assert L.gr_rayset_endpoints(h, C.byref(cfg), C.byref(rs), pts.ctypes.data, None) == -1
    assert L.gr_render(None, C.byref(cfg), C.byref(pl), C.byref(pf), C.byref(rg), img.ctypes.data, None) == -1
    # separable ray sets (ABI 4): missing tables, rays past the end of the set, a stride below the block, heights
    r, cs, sn, hs = np.linspace(2.0, 9.0, 8), np.cos(np.linspace(0, 6, 8)), np.sin(np.linspace(0, 6, 8)), np.zeros(64)
    pts64 = np.zeros(64, dtype=G.POINT_DTYPE)

    def sep(**kw):
        q = G._lib.gr_rayset()
        q.sep_r, q.sep_cos, q.sep_sin, q.sep_nr, q.sep_nt, q.sep_tiled, q.n = r.ctypes.data, cs.ctypes.data, sn.ctypes.data, 8, 8, 1, 64
        for k, v in kw.items():
            setattr(q, k, v)
        return L.gr_rayset_endpoints(h, C.byref(cfg), C.byref(q), pts64.ctypes.data, None)

    assert sep() == 0
    assert sep(sep_cos=None) == -1 and sep(sep_nt=0) == -1
    assert sep(n=65) == -1 and sep(sep_first=1) == -1 and sep(sep_first=-1) == -1
    assert sep(sep_block=16, sep_stride=8) == -1
    assert sep(sep_first=0, sep_block=16, sep_stride=32, n=32) == 0          # blocks 0 and 2 of four
    assert sep(sep_first=16, sep_block=16, sep_stride=32, n=48) == -1        # a third block would start at ray 80
    assert sep(height=hs.ctypes.data) == -1
    assert call() == 0 and np.isfinite(img).sum() > 0
