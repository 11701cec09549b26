"""Cunningham transfer functions with the reference's own root finder and dual-number Jacobians:
`find_offsets_for_radius_newton_ad` + the tangent flavour of the integrator, here in its host build (no GPU), against the
values the reference records (test/smoke-tests/cunningham-transfer-functions.jl:25-39, atol 1e-3).

What this pins: with the reference's Newton iteration (dual-number derivative, exit at the first |ρ - rₑ| <= 1e-7,
src/transfer-functions/cunningham-transfer-functions.jl:72-117 via src/tracing/precision-solvers.jl:230-330) and Jacobians
from dual numbers through the integrator (precision-solvers.jl:401-451), the recorded statistics are reproduced to 1e-5
... 1e-7 wherever they are well-conditioned -- two to four digits below the reference's own tolerance.  The rₑ = 4
values at low inclination (3°, 30°, 35°) are the ones still outside 1e-3 (by 1.85e-2, 9.5e-3, 2.1e-3): strict xfails
below; what this build gives there depends neither on the root finder nor on whether the tangents enter the error norm
(round 3: DiffEqBase's norm on Dual state, now the default, moved no statistic by more than 1e-5)."""
import math

import numpy as np
import pytest

import harness as Hh

GOLD = {(3, 4.0): 0.14048899037409682, (35, 4.0): 0.10846177995555085, (74, 4.0): 0.05550300700779827,
        (85, 4.0): 0.03602870590038378, (30, 4.0): 0.11958152396826184, (30, 7.0): 0.12205125501900763,
        (30, 10.0): 0.1265019201038228, (30, 15.0): 0.12875961522283233, (30, 300.0): 0.13378948600255888,
        (30, 800.0): 0.13470290875241375, (30, 1000.0): 0.13319637850028626}


def measure(c):
    return float(np.sum(c.f * c.g_star) / c.f.size)


def ctf(G, angle, radii, **kw):
    x = np.array([0.0, 100_000.0, math.radians(angle), 0.0])
    m, tr = Hh.tangent_tracer(G, 0.998, x, 2 * x[1])
    return G.transfer_functions.cunningham_transfer_functions(m, x, G.ThinDisc(0.0, float("inf")), radii, N=80, tracer=tr, **kw)


def test_well_conditioned_reference_values_to_1e4(G):
    """Measured with this build: (30°, 7) -6.7e-6, (30°, 10) 4.0e-5, (30°, 300) 5e-5, (30°, 800) -3.7e-7,
    (30°, 1000) 2.1e-5 -- bound 2e-4 (another compiler's rounding moves these by up to 1e-4: the statistic amplifies
    relative noise in g by 1e6), five times below the reference's atol."""
    radii = [7.0, 10.0, 300.0, 800.0, 1000.0]
    out = ctf(G, 30, radii, root_finder="reference")
    for c, r in zip(out, radii):
        assert c.f.size == 114 and not np.any(np.isnan(c.f))
        assert measure(c) == pytest.approx(GOLD[(30, r)], abs=2e-4), r


def test_reference_values_within_the_reference_tolerance(G):
    """(74°, 4) 4.6e-6; (30°, 15) 4.1e-4 and (85°, 4) -5.5e-4 inside the reference's 1e-3."""
    assert measure(ctf(G, 74, [4.0], root_finder="reference")[0]) == pytest.approx(GOLD[(74, 4.0)], abs=2e-4)
    assert measure(ctf(G, 30, [15.0], root_finder="reference")[0]) == pytest.approx(GOLD[(30, 15.0)], abs=1e-3)
    assert measure(ctf(G, 85, [4.0], root_finder="reference")[0]) == pytest.approx(GOLD[(85, 4.0)], abs=1e-3)


@pytest.mark.parametrize("angle,bound", [(3, 2.5e-2), (30, 1.5e-2), (35, 4e-3)])
def test_low_inclination_inner_disc_is_stable_here_but_off_the_record(G, angle, bound):
    """rₑ = 4 at low inclination: g_max - g_min is small (0.037 at 3°) and 18 + 18 of the 114 samples sit within 1e-3 of
    g✶ = 0 / 1, where f ∝ sqrt(g✶(1 - g✶)) / |J| is the ratio of two vanishing quantities.  This build gives the SAME
    statistic whichever root finder is used once the Jacobian comes from dual numbers (converged, reference-faithful and
    deliberately loose roots agree to 1e-4), and f tends to a clean limit at the ends (0.250 for every end sample at 3°);
    the recorded values are 1.85e-2 (3°), 9.5e-3 (30°), 2.1e-3 (35°) higher -- not monotonic in the inclination, i.e. carried
    by individual end samples of the reference's run, which cannot be reproduced without running it.  Pinned: our value's
    stability, and the distance to the record."""
    conv = ctf(G, angle, [4.0], root_finder="reference")[0]
    loose = ctf(G, angle, [4.0], root_finder="polished", polish=False)[0]
    tight = ctf(G, angle, [4.0], root_finder="polished", polish=True)[0]
    assert measure(conv) == pytest.approx(measure(loose), abs=2e-4)
    assert measure(conv) == pytest.approx(measure(tight), abs=2e-4)
    assert measure(conv) == pytest.approx(GOLD[(angle, 4.0)], abs=bound)
    if angle == 3:
        assert ((conv.g_star > 0.999) | (conv.g_star < 0.001)).sum() >= 30
        top = conv.f[(conv.g_star > 0.999) & (conv.g_star < 1.0)]     # the extremal sample itself has f = 0
        assert np.ptp(top) < 0.02 * np.mean(top)                     # a clean limit, not noise


UNMET = [(3, 4.0), (30, 4.0), (35, 4.0)]


@pytest.mark.parametrize("angle,re", [pytest.param(a, r, marks=pytest.mark.xfail(strict=True, reason="recorded statistic not reproduced "
                                                   "with the 114 samples the reference's current source stores (f-4, VERDICT r2)"))
                                      for a, r in UNMET])
def test_unmet_reference_values_stay_visible(G, angle, re):
    """The three recorded statistics this build does not reproduce, inside the reference's own tolerance (atol 1e-3):
    strict xfails, so the gap shows in every test record and closing it turns the suite red until the marks go."""
    assert measure(ctf(G, angle, [re], root_finder="reference")[0]) == pytest.approx(GOLD[(angle, re)], abs=1e-3)


@pytest.mark.parametrize("angle,re,k_min,bound", [(3, 4.0, 2, 1e-4), (30, 4.0, 8, 1e-4), (35, 4.0, 15, 5e-4), (74, 4.0, 17, 1e-4)])
def test_the_unmet_records_are_this_sample_set_with_a_shortened_g_min_search(G, angle, re, k_min, bound):
    """A numerical lead, recorded as such (scripts/tf_truncation_scan.py -> profiles/r3_tf_truncation.json).  The three unmet
    records equal THIS build's statistic when only the first k_min stored calls of the g_min golden-section search enter the
    mean -- k_min = 2 at 3°, 8 at 30°, 15 at 35° (agreement 1e-5, 2e-5, 2e-4), 17 = all of them at 74° (4e-6) -- i.e.
    `N = count(data.mask)` samples with N = 99, 105, 112, 114 (cunningham-transfer-functions.jl:303-334 truncates its
    accumulator to the calls Optim's search really made).  Monotonic in the inclination, the g_max search complete in
    every case, all three inside the reference's tolerance: a difference in HOW MANY samples the search stores, not in any
    g or f (the f values satisfy their normalisation identity below).  The current source of the reference and Optim's
    GoldenSection as published give 17 calls per search (`iterations = N`, no convergence exit reachable at these bracket
    widths); what shortened the searches of the recorded run cannot be told without running it.  This build follows the
    source: 114 samples.

    WHAT THIS TEST IS NOT: parity.  It is a one-integer fit per record -- k_min is chosen from 1..17 so that the statistic lands
    on the record, and 17 candidates spanning ~2e-2 make a hit inside 2e-4 close to chance for ONE record (probability ~0.2);
    three records whose fitted k_min rise monotonically with the inclination, plus the control at 74°, are less likely by
    accident (~1e-2) but still a conjecture about a run nobody here can repeat.  The three records stay strict xfails at
    the reference's tolerance (above, and tests/test_gpu_tangent.py); this test only keeps the lead reproducible."""
    raw = []
    c = ctf(G, angle, [re], root_finder="reference", _raw=raw)[0]
    assert c.f.size == 114
    th, g, J, t = raw[0][0][0]                          # columns 80..96 = the g_min search in call order
    keep = np.ones(114, bool)
    keep[80 + k_min:97] = False
    gg, JJ = g[keep], J[keep]
    gs = (gg - gg.min()) / np.ptp(gg)
    f = gg * np.sqrt(gs * (1 - gs)) * np.ptp(gg) * JJ / (math.pi * re)
    assert float(np.mean(f * gs)) == pytest.approx(GOLD[(angle, re)], abs=bound)
    # the calls left out all sit at g✶ ~ 0: zero terms of the sum, only the sample count changes
    full = (g - g.min()) / np.ptp(g)
    assert np.all(full[80 + max(k_min, 3):97] < 6e-3)


def test_problem_cases_with_the_reference_root_finder(G):
    """'ones that have been problematic in the past' (smoke-tests/cunningham-transfer-functions.jl:41-51,
    test/transfer-functions/test-problem-cases.jl:20-35) through the Newton iteration with dual-number derivatives:
    grazing inclinations, retrograde spins, the ISCO itself, a = 1 with the emitter 1 % outside the horizon.  Every root
    find converges, every sample is finite, and the safeguarded difference-quotient route gives the same statistic."""
    cases = [(-0.6, 1e5, 88, 784.8253509875607), (-0.998, 1e5, 88, 953.9915665264327), (0.0, 1e5, 88, 631.1007589946363),
             (0.744, 1e5, 88, 3.1880132176627862), (0.998, 5e5, 88.0, 1.2469706551751847),
             (0.10324137931034483, 5e5, 82.06896551724138, 21.755193176415617), (0.998, 5e5, 88.0, 1.2369706551751847),
             (0.9291724137931034, 5e5, 88.0, 2.1204839212537308), (1.0, 1e5, 88, 1.01)]
    for a, r_obs, angle, r in cases:
        x = np.array([0.0, r_obs, math.radians(angle), 0.0])
        m, tr = Hh.tangent_tracer(G, a, x, 2 * r_obs)
        d = G.ThinDisc(0.0, float("inf"))
        c = G.transfer_functions.cunningham_transfer_function(m, x, d, r, N=80, tracer=tr, root_finder="reference")
        p = G.transfer_functions.cunningham_transfer_function(m, x, d, r, N=80, tracer=tr, root_finder="polished")
        assert c.f.size == 114 and np.all(np.isfinite(c.f)) and 0 < c.gmin < c.gmax < 2.0, (a, angle, r)
        assert measure(c) == pytest.approx(measure(p), rel=3e-3), (a, angle, r)


@pytest.mark.parametrize("angle,re,rel", [(3, 4.0, 2e-4), (30, 4.0, 5e-4), (35, 4.0, 5e-4), (85, 4.0, 6e-3), (30, 10.0, 2e-4)])
def test_transfer_functions_satisfy_their_normalisation_identity(G, angle, re, rel):
    """A reference-independent check of the transfer functions themselves, at the very radii and inclinations whose recorded
    statistics this build does not meet: by construction f = g √(g✶(1-g✶)) (g_max - g_min) |∂(α,β)/∂(rₑ,g)| / (π rₑ), so
    with g✶ = sin²φ the closed integral ∮ (f/g) 2 dφ over both branches equals (1/π rₑ) dA/drₑ, where A(rₑ) is the area the
    image of the ring ρ = rₑ encloses -- a number the root finder alone gives (A = ½∮r(θ)² dθ, differenced in rₑ).  The 114
    samples of cunningham_transfer_function satisfy it to 4e-6 at (3°, 4), 1e-4 at (30°, 4), 4e-3 at (85°, 4): the f values are
    right where the recorded mean(f g✶) is 15 % / 8 % / 1.5 % away, i.e. that difference lies in which samples enter the mean."""
    from gradus_jl_amd import transfer_functions as TF

    x = np.array([0.0, 100_000.0, math.radians(angle), 0.0])
    m, tr = Hh.tangent_tracer(G, 0.998, x, 2 * x[1])
    th = np.linspace(0.0, 2 * math.pi, 1441)[:-1]

    def area(r_e):
        r = TF.find_offsets_for_radius_newton_ad(tr, np.full(th.size, r_e), th, r_min=m.inner_radius())[0]
        return 0.5 * np.sum(r * r) * (th[1] - th[0])

    dA = (area(re + 1e-3) - area(re - 1e-3)) / 2e-3
    c = ctf(G, angle, [re])[0]                                  # samples sorted by θ: a closed curve in (φ, f/g)
    gs = np.clip(c.g_star, 0.0, 1.0)
    y = c.f / (c.gmin + gs * (c.gmax - c.gmin))
    for k in np.flatnonzero(c.f == 0.0):                        # the two extremal samples: f = 0 exactly, its limit is finite
        y[k] = 0.5 * (y[k - 1] + y[(k + 1) % y.size])
    phi = np.arcsin(np.sqrt(gs))
    total = np.sum(0.5 * (y + np.roll(y, -1)) * 2.0 * np.abs(np.roll(phi, -1) - phi))
    assert total == pytest.approx(dA / (math.pi * re), rel=rel)
