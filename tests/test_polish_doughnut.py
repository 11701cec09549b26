"""PolishDoughnut (gradus.jl_amd/polish_doughnut.py): the reference's recorded cross-section
fingerprint (test/discs/test-polish-doughnut.jl) -- which also pins the host restatement of
OrdinaryDiffEq's Tsit5 stepping rules on reference output."""
import math

import numpy as np
import pytest


def test_cross_section_fingerprint(G):
    m = G.KerrMetric(1.0, 0.2)
    d = G.PolishDoughnut(m, rₖ=12.0, n=0.21)
    r = np.linspace(10.0, 15.0, 200)
    h = d.cross_section(r)
    assert float(h.sum()) == pytest.approx(219.97440610254944, abs=1e-5)      # the reference's tolerance
    assert float(h.sum()) == pytest.approx(219.97440610254944, rel=1e-12)     # same step sequence, to rounding
    assert d.inner_radius < 12.0 < d.outer_radius and np.all(np.diff(d.r) > 0)
    assert d.cross_section(d.inner_radius - 1e-6) == 0.0 and d.cross_section(d.outer_radius + 1e-6) == 0.0
    # steps are capped by dtmax = 0.05 after the start-up: arc length between saved points
    ds = np.hypot(np.diff(d.r), np.diff(d.z))
    assert ds.max() < 0.0501 and np.median(ds) > 0.04


def test_tsit5_restatement_on_a_known_problem(G):
    """exponential decay: the adaptive solver lands on exp(-t) within its tolerance and obeys dtmax"""
    from gradus_jl_amd.polish_doughnut import tsit5_solve

    sol = tsit5_solve(lambda u: -u, [1.0], 0.0, 2.0, abstol=1e-9, reltol=1e-9)
    assert sol[-1][0] == pytest.approx(math.exp(-2.0), rel=1e-7)
    sol2 = tsit5_solve(lambda u: -u, [1.0], 0.0, 2.0, dtmax=0.1)
    assert len(sol2) >= 21
    stop = tsit5_solve(lambda u: np.array([1.0]), [0.0], 0.0, 10.0, terminate=lambda u: u[0] > 3.0)
    assert stop[-1][0] > 3.0 and stop[-1][0] == stop[-2][0]          # DiscreteCallback saves before and after


def test_sampled_torus_matches_isobar(G):
    d = G.PolishDoughnut(G.KerrMetric(1.0, 0.2), rₖ=12.0, n=0.21)
    td = d.thick_disc(samples=4096)
    ρ = np.linspace(td.ρ_range[0], td.ρ_range[1], 4096)
    np.testing.assert_allclose(td.table, d.cross_section(ρ), atol=1e-12)
    assert td.table.max() == pytest.approx(float(d.z.max()), rel=2e-3)
