"""MeshAccretionGeometry(mesh) (src/geometry/meshes.jl; jsf_algorithm and intersects_geometry, src/geometry/intersections.jl) --
the last geometry of the reference's catalogue, a DiscreteCallback on the Cartesian line element of every accepted step.  The
reference holds no test and no recorded value for it, so parity is anchored on:

  * the Jiménez-Segura-Feito predicate against an independent segment/triangle test (Möller-Trumbore) and hand-made cases;
  * properties of the oracle that tie the mesh to the pinned mesh-less path (a mesh no ray comes near changes nothing, bit for
    bit; a hit ends AT an accepted step's end of the mesh-less trajectory, below the surface it crossed);
  * oracle ⇄ the kernel logic compiled for the host ⇄ (GPU) the device through the C ABI, on three scenes and two metrics.
"""
import math

import numpy as np
import pytest

import harness as Hh

X_OBS = np.array([0.0, 1000.0, math.radians(72), 0.0])


def _rays(G, m, n=24, half=(14.0, 10.0)):
    a = np.linspace(-half[0], half[0], n)
    b = np.linspace(-half[1], half[1], n)
    aa, bb = np.meshgrid(a, b)
    return np.stack([G.map_impact_parameters(m, X_OBS, al, be) for al, be in zip(aa.ravel(), bb.ravel())])


def annulus(r_in, r_out, n_r, n_phi, z=0.0, up=True):
    """A flat ring in the plane z: 2 n_r n_phi triangles, front faces (the side jsf_algorithm sees) up or down."""
    rr = np.linspace(r_in, r_out, n_r + 1)
    ph = np.linspace(0.0, 2.0 * math.pi, n_phi + 1)
    tri = []
    for i in range(n_r):
        for j in range(n_phi):
            p00 = (rr[i] * math.cos(ph[j]), rr[i] * math.sin(ph[j]), z)
            p10 = (rr[i + 1] * math.cos(ph[j]), rr[i + 1] * math.sin(ph[j]), z)
            p01 = (rr[i] * math.cos(ph[j + 1]), rr[i] * math.sin(ph[j + 1]), z)
            p11 = (rr[i + 1] * math.cos(ph[j + 1]), rr[i + 1] * math.sin(ph[j + 1]), z)
            # (V1, V2, V3) with (V1 - V3) x (V2 - V3) along +z
            a, b = (p10, p11, p00), (p11, p01, p00)
            tri += [a, b] if up else [(a[1], a[0], a[2]), (b[1], b[0], b[2])]
    return np.array(tri)


def box(centre, size):
    """A closed cube, front faces outwards (12 triangles)."""
    c, h = np.asarray(centre, dtype=float), 0.5 * size
    v = np.array([[x, y, z] for x in (-h, h) for y in (-h, h) for z in (-h, h)]) + c      # index = 4 ix + 2 iy + iz
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]   # each seen from outside, counter-clockwise
    tri = []
    for q in quads:
        p = v[list(q)]
        n = np.cross(p[1] - p[0], p[2] - p[0])
        if np.dot(n, p[0] - c) < 0:
            p = p[::-1]
        # (V1, V2, V3): (V1 - V3) x (V2 - V3) outwards
        tri += [(p[1], p[2], p[0]), (p[2], p[3], p[0])]
    return np.array(tri)


def slab(r_in, r_out, n_r, n_phi, half):
    """A ring-shaped slab: a top face at z = +half looking up and a bottom face at z = -half looking down (no side walls).
    The reference tests a step only when it ENDS strictly inside the bounding box (in_nearby_region, meshes.jl:46-51), so a
    mesh needs a volume to be seen at all -- a flat one has an empty box."""
    return np.concatenate([annulus(r_in, r_out, n_r, n_phi, z=half, up=True), annulus(r_in, r_out, n_r, n_phi, z=-half, up=False)])


def octahedron(centre, radius):
    c = np.asarray(centre, dtype=float)
    ax = np.eye(3) * radius
    tri = []
    for sx in (1, -1):
        for sy in (1, -1):
            for sz in (1, -1):
                p = [c + sx * ax[0], c + sy * ax[1], c + sz * ax[2]]
                n = np.cross(p[0] - p[2], p[1] - p[2])
                if np.dot(n, p[0] + p[1] + p[2] - 3 * c) < 0:
                    p = [p[1], p[0], p[2]]
                tri.append(p)
    return np.array(tri)


def shards(n=900, radius=14.0, far=5000.0, seed=3):
    """Small triangles scattered through a ball around the hole, every orientation, plus two far away that stretch the
    bounding box to 10^4 per axis: the device's grid over the first vertices has to coarsen its cells (gr_mesh_grid.hpp)."""
    rng = np.random.default_rng(seed)
    c = rng.normal(size=(n, 3))
    c *= (radius * rng.random(n) ** (1 / 3) / np.linalg.norm(c, axis=1))[:, None]
    tri = c[:, None, :] + rng.normal(size=(n, 3, 3)) * 0.9
    extra = np.array([[[far, far, far], [far + 1, far, far], [far, far + 1, far]], [[-far, -far, -far], [-far - 1, -far, -far], [-far, -far - 1, -far]]])
    return np.concatenate([tri, extra])


SCENES = {
    "ring slab": lambda: slab(2.5, 9.0, 5, 24, 1.5),
    "scattered shards": shards,
    "cube beside the hole": lambda: box((0.0, 6.0, 2.0), 4.0),
    "octahedron in front": lambda: octahedron((5.0, -3.0, 1.0), 3.0),
}


# ---------------------------------------------------------------- the predicate
def _moller_trumbore(V1, V2, V3, Q1, Q2):
    e1, e2 = V2 - V1, V3 - V1
    d = Q2 - Q1
    h = np.cross(d, e2)
    a = np.dot(e1, h)
    if abs(a) < 1e-14:
        return False, 0.0, 0.0, 0.0
    s = Q1 - V1
    u = np.dot(s, h) / a
    q = np.cross(s, e1)
    v = np.dot(d, q) / a
    t = np.dot(e2, q) / a
    return (u >= 0 and v >= 0 and u + v <= 1 and 0 <= t <= 1), u, v, t


def test_jsf_algorithm_hand_made_cases(oracle):
    V1, V2, V3 = np.array([1.0, 0, 0]), np.array([0.0, 1, 0]), np.array([0.0, 0, 0])      # front side: +z
    hit = lambda q1, q2: oracle.jsf_algorithm(V1, V2, V3, np.array(q1, dtype=float), np.array(q2, dtype=float))[0]
    assert hit([0.2, 0.2, 1], [0.2, 0.2, -1])            # through the interior, front to back
    assert not hit([0.2, 0.2, -1], [0.2, 0.2, 1])        # back to front: culled (w < -ϵ)
    assert not hit([0.2, 0.2, 1], [0.2, 0.2, 0.5])       # stops short of the plane (s > ϵ)
    assert not hit([2.0, 2.0, 1], [2.0, 2.0, -1])        # crosses the plane outside the triangle
    assert not hit([0.2, 0.2, 1], [1.2, 0.2, 1])         # parallel to the plane
    assert hit([0.2, 0.2, 1], [0.2, 0.2, 0.0])           # ends in the plane (s = 0 <= ϵ)
    assert hit([0.0, 0.0, 1], [0.0, 0.0, -1])            # through the vertex V3 (t = u = 0 >= -ϵ)
    assert hit([0.5, 0.5, 1], [0.5, 0.5, -1])            # through the edge V1 V2
    assert not hit([0.51, 0.51, 1], [0.51, 0.51, -1])    # just outside that edge
    # starts IN the plane (|w| <= ϵ): the second branch of the algorithm.  As written in the reference (intersections.jl:84-97)
    # it asks for t, u <= ϵ and -s <= t + u with s < -ϵ, which no segment through the interior satisfies: such a step is
    # not a hit there, so it is not one here
    assert not hit([0.2, 0.2, 0.0], [0.2, 0.2, -1])
    assert not hit([0.2, 0.2, 0.0], [0.2, 0.2, 1])
    assert not hit([0.2, 0.2, 0.0], [0.7, 0.1, 0.0])


def test_jsf_algorithm_agrees_with_moller_trumbore(oracle):
    rng = np.random.default_rng(5)
    n_hit = 0
    for _ in range(4000):
        V = rng.normal(size=(3, 3)) * 2.0
        Q1, Q2 = rng.normal(size=3) * 3.0, rng.normal(size=3) * 3.0
        ref, u, v, t = _moller_trumbore(V[0], V[1], V[2], Q1, Q2)
        # away from the edges of the decision (ϵ = 1e-8 on un-normalised products there, exact comparisons here)
        if min(abs(u), abs(v), abs(1 - u - v), abs(t), abs(1 - t)) < 1e-6:
            continue
        front = np.dot(Q1 - V[2], np.cross(V[0] - V[2], V[1] - V[2])) > 0
        got, _ = oracle.jsf_algorithm(V[0], V[1], V[2], Q1, Q2)
        assert got == (ref and front)
        n_hit += got
    assert n_hit > 100


# ---------------------------------------------------------------- the oracle's trace
def _orc(oracle, metric, params, mesh, v):
    cfg = oracle.make_config(metric, params, disc=None if mesh is None else {"mesh": mesh}, lambda_max=2000.0)
    return oracle.trace(cfg, X_OBS, v)


def test_oracle_mesh_hits_end_at_a_step_of_the_free_ray(G, oracle):
    m = G.KerrMetric(1.0, 0.9)
    v = _rays(G, m)
    free = _orc(oracle, "kerr", (1.0, 0.9), None, v)
    # a mesh whose bounding box no ray enters: nothing changes, bit for bit
    far = _orc(oracle, "kerr", (1.0, 0.9), box((0.0, 0.0, 400.0), 2.0), v)
    for f in ("status", "lambda_max", "x", "v"):
        assert np.array_equal(far[f], free[f]), f
    ring = slab(2.5, 9.0, 5, 24, 1.5)
    got = _orc(oracle, "kerr", (1.0, 0.9), ring, v)
    hit = got["status"] == 2
    assert hit.sum() > 60
    assert np.array_equal(got["status"][~hit], free["status"][~hit])
    # a DiscreteCallback: the ray stops at an accepted step's end strictly inside the bounding box, having come in through a
    # front face -- the path up to there is the free ray's (compare λ: the hit ray ends earlier)
    x = got["x"][hit]
    z = x[:, 1] * np.cos(x[:, 2])
    rho = x[:, 1] * np.sin(x[:, 2])
    assert (np.abs(z) < 1.5).all()
    assert (rho < 9.0 * math.sqrt(2.0)).all()
    assert (got["lambda_max"][hit] < free["lambda_max"][hit]).all()
    # the same steps: a hit ray's end state is the free ray's state at that affine time (a free trace cut at λ_hit takes the
    # same steps and a last one that differs by the rounding of λ_hit - λ_prev)
    for i in np.flatnonzero(hit)[::7]:
        cut = oracle.trace(oracle.make_config("kerr", (1.0, 0.9), lambda_max=float(got["lambda_max"][i])), X_OBS, v[i:i + 1])
        assert cut["status"][0] == 3
        np.testing.assert_allclose(cut["x"][0], got["x"][i], rtol=1e-11, atol=1e-11)
        np.testing.assert_allclose(cut["v"][0], got["v"][i], rtol=1e-11, atol=1e-11)


METRICS = [("kerr", (1.0, 0.9), lambda G: G.KerrMetric(1.0, 0.9)),
           ("johannsen-psaltis", (1.0, 0.6, 1.0), lambda G: G.JohannsenPsaltisMetric(1.0, 0.6, 1.0))]


def _compare(oracle, name, params, mesh, v, got, ref, rtol=1e-6, every=5):
    """A mesh hit ends at a STEP END (DiscreteCallback, no root finding), and two implementations of the controller do not
    share their step grids to the last bit (fp32 controller arithmetic on the device): the affine time of the
    last step is implementation-dependent the way the chart's end points are.  What parity means here: the same decision for
    every ray (flips where a step ends within rounding of a face are counted like disc-rim flips); stopping
    within a step or two of the oracle's stop, strictly inside the bounding box; and the stopping STATE on the oracle's
    trajectory -- the oracle traced mesh-less to the device's stopping time agrees to rtol."""
    # (both hit, but another triangle a few steps apart: the same kind of flip -- a step that ends within rounding of a face
    # or of the 3.0 sphere is seen by one step grid and not by the other)
    mism = (got["status"] != ref["status"]) | ((ref["status"] == 2) & (np.abs(got["lambda_max"] - ref["lambda_max"]) >= 4.0))
    assert mism.sum() <= max(2, got.size // 100), f"{mism.sum()} class mismatches of {got.size}"
    # (a ray that stalls at the horizon of a metric whose chart ends inside it -- dilaton-axion: step below dtmin, flagged,
    # NoStatus -- has no end state to compare: where it gives up depends on the last bits)
    free = ~mism & (ref["status"] == 3) & ((ref["flags"] & 0xFFFF) == 0) & ((got["flags"] & 0xFFFF) == 0)
    np.testing.assert_allclose(got["lambda_max"][free], ref["lambda_max"][free], rtol=rtol)
    for f in ("x", "v"):
        # (rays that wind around the photon sphere amplify the last bits of every step: the same handful of ill-conditioned
        # rays every parity test of the free path sets aside)
        err = (np.abs(got[f][free] - ref[f][free]) / np.maximum(np.abs(ref[f][free]), 1.0)).max(axis=1)
        assert (err < rtol).mean() > 0.985 and np.median(err) < 1e-8, (f, err.max(), np.median(err), (err >= rtol).sum())
    hit = ~mism & (ref["status"] == 2)
    assert np.abs(got["lambda_max"][hit] - ref["lambda_max"][hit]).max() < 4.0
    box6 = oracle.mesh_table_header(mesh)
    x = got["x"][hit]
    cart = np.stack([x[:, 1] * np.sin(x[:, 2]) * np.cos(x[:, 3]), x[:, 1] * np.sin(x[:, 2]) * np.sin(x[:, 3]), x[:, 1] * np.cos(x[:, 2])], axis=1)
    assert ((cart > box6[0::2]) & (cart < box6[1::2])).all()
    for i in np.flatnonzero(hit)[::every]:
        cut = oracle.trace(oracle.make_config(name, params, lambda_max=float(got["lambda_max"][i])), X_OBS, v[i:i + 1])
        assert cut["status"][0] == 3
        for f in ("x", "v"):
            scale = np.maximum(np.abs(cut[f][0]), 1.0)
            assert (np.abs(got[f][i] - cut[f][0]) / scale).max() < rtol, (i, f)


@pytest.mark.parametrize("metric", METRICS, ids=[t[0] for t in METRICS])
@pytest.mark.parametrize("scene", list(SCENES), ids=list(SCENES))
def test_host_kernel_logic_equals_oracle_on_mesh_scenes(G, oracle, scene, metric):
    name, params, mk = metric
    m = mk(G)
    v = _rays(G, m, 32)
    mesh = SCENES[scene]()
    cfg = G.tracing_configuration(m, X_OBS, v, G.MeshAccretionGeometry(mesh), (0.0, 2000.0), ensemble=G.EnsembleMI355X.__new__(G.EnsembleMI355X))
    got = Hh.trace_endpoints(G, cfg)
    ref = _orc(oracle, name, params, mesh, v)
    assert (ref["status"] == 2).sum() > 15
    _compare(oracle, name, params, mesh, v, got, ref)


def _low_lid():
    """A ring whose upper face lies just above the equatorial plane (z = 0.3) over a deep box (down to z = -2): a step of
    the size the integrator takes there comes in through the face and ends below the plane."""
    return np.concatenate([annulus(2.5, 9.0, 5, 24, z=0.3, up=True), annulus(2.5, 9.0, 5, 24, z=-2.0, up=False)])


def test_a_later_callback_of_the_set_overrides_the_mesh(G, oracle):
    """CallbackSet order: the geometry's DiscreteCallback, the user's, the chart's (callbacks.jl:19-38 with bootstrap.jl:11-21);
    every callback whose condition holds applies its affect!, so where a step both enters a front face and drops below
    domain_upper_hemisphere's δ the ray ends there as OutOfDomain, not as IntersectedWithGeometry."""
    m = G.KerrMetric(1.0, 0.9)
    v = _rays(G, m, 32)
    mesh = _low_lid()
    cfg = G.tracing_configuration(m, X_OBS, v, G.MeshAccretionGeometry(mesh), (0.0, 2000.0), callback=G.domain_upper_hemisphere(0.35),
                                  ensemble=G.EnsembleMI355X.__new__(G.EnsembleMI355X))
    got = Hh.trace_endpoints(G, cfg)
    ocfg = oracle.make_config("kerr", (1.0, 0.9), disc={"mesh": mesh}, lambda_max=2000.0, upper_hemisphere=True, hemi_delta=0.35)
    ref = oracle.trace(ocfg, X_OBS, v)
    plain = _orc(oracle, "kerr", (1.0, 0.9), mesh, v)
    # rays the mesh alone ends below the plane z = δ (here just above the lid: a step that comes in through the lid starts
    # above δ and ends below it) are OutOfDomain with the hemisphere callback in the set -- at the same step
    z = plain["x"][:, 1] * np.cos(plain["x"][:, 2])
    both = (plain["status"] == 2) & (z < 0.35)
    assert both.sum() > 20
    # (most of them were below the plane earlier and end there; some end AT the step of the hit: both conditions held)
    assert (ref["status"][both] == 0).all() and (ref["lambda_max"][both] <= plain["lambda_max"][both]).all()
    assert (both & (ref["lambda_max"] == plain["lambda_max"])).sum() >= 3
    assert (got["status"] != ref["status"]).sum() <= 4
    same = got["status"] == ref["status"]
    assert (got["status"][both & same] == 0).all()


def test_grid_walk_reaches_every_triangle_the_reference_would_test(G):
    """gr_mesh_grid.hpp + Ray::mesh_cells against the reference's loop over the whole list (meshes.jl:53-64), point by point:
    every triangle whose first vertex is within 3 of the point is among those the grid walk visits -- for compact meshes
    (cells of 3), for one stretched to 10^4 per axis (coarser cells) and for points outside the grid of first vertices; and the
    walk visits a small part of a large mesh."""
    import ctypes as C

    L = Hh.lib()
    rng = np.random.default_rng(8)
    total = 0
    for mesh, spread in ((shards(1500, 14.0, 20.0, 1), 16.0), (shards(1200, 20.0, 5000.0, 2), 22.0), (slab(2.0, 50.0, 10, 96, 1.0), 52.0),
                         (box((3.0, -2.0, 1.0), 4.0), 6.0)):
        g = G.MeshAccretionGeometry(mesh)
        out = (C.c_int64 * 3)()
        visited = []
        for _ in range(300):
            q = rng.uniform(-spread, spread, 3)
            if rng.random() < 0.5:                   # half of the points next to a first vertex
                q = mesh[rng.integers(len(mesh)), 0] + rng.normal(size=3) * 1.5
            qa = np.ascontiguousarray(q, dtype=np.float64)
            assert L.hh_mesh_candidates(g.table.ctypes.data_as(C.c_void_p), C.c_int64(len(g)), qa.ctypes.data_as(C.c_void_p), out) == 0
            assert out[1] == out[0], (q, out[0], out[1])
            total += out[0]
            visited.append(out[2])
        if len(mesh) == 1502:                        # the compact cloud: cells of 3 (the stretched one walks cells of ~100)
            assert np.mean(visited) < 0.2 * len(mesh)
    assert total > 3000


def test_mesh_constructor_and_abi_table(G):
    mesh = box((1.0, -2.0, 3.0), 2.0)
    g = G.MeshAccretionGeometry(mesh)
    assert len(g) == 12 and g.mesh.shape == (12, 3, 3)
    assert (g.x_extent, g.y_extent, g.z_extent) == ((0.0, 2.0), (-3.0, -1.0), (2.0, 4.0)) == G.bounding_box(mesh)
    assert g.table.shape == (6 + 9 * 12,) and np.array_equal(g.table[:6], [0.0, 2.0, -3.0, -1.0, 2.0, 4.0])
    assert np.array_equal(g.table[6:15], mesh[0].ravel())
    # nested lists as the reference's constructor takes them; extents of one's own
    g2 = G.MeshAccretionGeometry([[list(p) for p in t] for t in mesh], z_extent=(2.5, 3.5))
    assert np.array_equal(g2.mesh, g.mesh) and g2.z_extent == (2.5, 3.5) and g2.x_extent == g.x_extent
    with pytest.raises(ValueError):
        G.MeshAccretionGeometry(np.zeros((0, 3, 3)))
    m = G.KerrMetric(1.0, 0.5)
    c = G.tracing_configuration(m, X_OBS, np.zeros((1, 4)), g, (0.0, 10.0), ensemble=G.EnsembleMI355X.__new__(G.EnsembleMI355X)).abi_config()
    assert c.disc_id == 8 and c.disc_table_n == 12 and c.disc_table == g.table.ctypes.data


# ---------------------------------------------------------------- GPU
@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1])
@pytest.mark.parametrize("metric", METRICS, ids=[t[0] for t in METRICS])
@pytest.mark.parametrize("scene", list(SCENES), ids=list(SCENES))
def test_device_equals_oracle_on_mesh_scenes(G, oracle, ens, scene, metric, kernel):
    ens.set("kernel", kernel)
    name, params, mk = metric
    m = mk(G)
    v = _rays(G, m, 48)
    mesh = SCENES[scene]()
    got = G.tracegeodesics(m, X_OBS, v, G.MeshAccretionGeometry(mesh), (0.0, 2000.0), ensemble=ens)
    ref = _orc(oracle, name, params, mesh, v)
    assert (ref["status"] == 2).sum() > 30
    _compare(oracle, name, params, mesh, v, got, ref, every=11)


@pytest.mark.gpu
def test_device_mesh_far_away_changes_nothing_and_image_path_agrees(G, ens):
    m = G.KerrMetric(1.0, 0.9)
    v = _rays(G, m, 40)
    free = G.tracegeodesics(m, X_OBS, v, (0.0, 2000.0), ensemble=ens)
    far = G.tracegeodesics(m, X_OBS, v, G.MeshAccretionGeometry(box((0.0, 0.0, 400.0), 2.0)), (0.0, 2000.0), ensemble=ens)
    for f in ("status", "lambda_max", "x", "v"):
        assert np.array_equal(far[f], free[f]), f
    # the fused image and the end-point cache + apply see the same mesh
    d = G.MeshAccretionGeometry(SCENES["cube beside the hole"]())
    pf = G.ConstPointFunctions.affine_time() @ G.ConstPointFunctions.filter_intersected()
    kw = dict(image_width=96, image_height=64, alpha_lims=(-14.0, 14.0), beta_lims=(-10.0, 10.0), ensemble=ens)
    _, _, img = G.rendergeodesics(m, X_OBS, d, 2000.0, pf=pf, **kw)
    _, _, cache = G.prerendergeodesics(m, X_OBS, d, 2000.0, **kw)
    img2 = G.apply(pf, cache)
    assert np.isfinite(img).sum() > 100
    assert np.array_equal(np.isnan(img), np.isnan(img2)) and np.array_equal(img[np.isfinite(img)], img2[np.isfinite(img2)])


@pytest.mark.gpu
@pytest.mark.parametrize("kernel", [0, 1])
def test_device_later_callback_overrides_the_mesh(G, oracle, ens, kernel):
    ens.set("kernel", kernel)
    m = G.KerrMetric(1.0, 0.9)
    v = _rays(G, m, 48)
    mesh = _low_lid()
    got = G.tracegeodesics(m, X_OBS, v, G.MeshAccretionGeometry(mesh), (0.0, 2000.0), ensemble=ens, callback=G.domain_upper_hemisphere(0.35))
    ref = oracle.trace(oracle.make_config("kerr", (1.0, 0.9), disc={"mesh": mesh}, lambda_max=2000.0, upper_hemisphere=True, hemi_delta=0.35), X_OBS, v)
    plain = _orc(oracle, "kerr", (1.0, 0.9), mesh, v)
    both = (plain["status"] == 2) & (plain["x"][:, 1] * np.cos(plain["x"][:, 2]) < 0.35)
    assert both.sum() > 50 and (ref["status"][both] == 0).all() and (both & (ref["lambda_max"] == plain["lambda_max"])).sum() >= 10
    assert (got["status"] != ref["status"]).sum() <= max(2, got["status"].size // 200)
    same = got["status"] == ref["status"]
    assert (got["status"][both & same] == 0).all()


@pytest.mark.gpu
def test_saved_paths_end_at_the_mesh_hit(G, ens):
    """gr_trace_paths (save_on = true) against a mesh: the path kernel runs the same DiscreteCallback -- the stored path ends at
    the end point the end-point kernels report, and every earlier row is a step of the mesh-less path."""
    m = G.KerrMetric(1.0, 0.9)
    v = _rays(G, m, 12)
    d = G.MeshAccretionGeometry(SCENES["ring slab"]())
    pts = G.tracegeodesics(m, X_OBS, v, d, (0.0, 2000.0), ensemble=ens)
    paths = G.tracegeodesic_paths(m, X_OBS, v, d, (0.0, 2000.0), ensemble=ens)
    free = G.tracegeodesic_paths(m, X_OBS, v, (0.0, 2000.0), ensemble=ens)
    hits = np.flatnonzero(pts["status"] == 2)
    assert hits.size > 10
    for j in hits:
        p, f = paths[j], free[j]
        assert p.point["status"] == 2 and p.λ[-1] == pts["lambda_max"][j] and np.array_equal(p.x[-1], pts["x"][j])
        k = p.λ.size
        assert k < f.λ.size and np.array_equal(p.λ, f.λ[:k]) and np.array_equal(p.x, f.x[:k])


@pytest.mark.gpu
def test_mesh_is_refused_where_it_is_not_built(G, ens):
    m = G.KerrMetric(1.0, 0.9)
    v = _rays(G, m, 8)
    d = G.MeshAccretionGeometry(SCENES["cube beside the hole"]())
    e32 = G.EnsembleMI355X(0, precision=32)
    with pytest.raises(G.GradusMI355XError, match="fp64"):
        G.tracegeodesics(m, X_OBS, v, d, (0.0, 2000.0), ensemble=e32)
    bad = SCENES["cube beside the hole"]().copy()
    bad[3, 1, 2] = np.inf
    with pytest.raises(G.GradusMI355XError, match="not finite"):
        G.tracegeodesics(m, X_OBS, v, G.MeshAccretionGeometry(bad, x_extent=(-1, 1), y_extent=(-1, 1), z_extent=(-1, 1)), (0.0, 2000.0), ensemble=ens)
    # a valid call on the same context afterwards
    out = G.tracegeodesics(m, X_OBS, v, d, (0.0, 2000.0), ensemble=ens)
    assert out["status"].size == 64


@pytest.mark.gpu
def test_mesh_through_several_contexts(G, ens):
    m = G.KerrMetric(1.0, 0.9)
    v = _rays(G, m, 40)
    d = G.MeshAccretionGeometry(SCENES["ring slab"]())
    one = G.tracegeodesics(m, X_OBS, v, d, (0.0, 2000.0), ensemble=ens)
    multi = G.EnsembleMI355X(devices=[0, 0, 0])
    three = G.tracegeodesics(m, X_OBS, v, d, (0.0, 2000.0), ensemble=multi)
    for f in ("status", "lambda_max", "x", "v"):
        assert np.array_equal(one[f], three[f]), f
