"""Host-side mirror of the reference's AbstractStaticAxisSymmetric metrics.

Only what the render path needs on the host: parameters -> (metric_id, params[8]) for the C
ABI, `metric_components` for the one-off observer set-up (LNRF basis), `inner_radius` and
`isco`.  The per-ray evaluation (with derivatives) happens in the HIP kernels.

Reference: src/metrics/kerr-metric.jl:11-28,62-72,91; src/metrics/johannsen-ad.jl:4-34,49-67;
src/metrics/kerr-metric-first-order.jl:297-337 (Z1, Z2, isco).
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np

GR_METRIC_KERR, GR_METRIC_JOHANNSEN = 0, 1


class AbstractMetric:
    metric_id: int = -1

    def abi_params(self):
        raise NotImplementedError

    def metric_components(self, r, theta):
        """(g_tt, g_rr, g_θθ, g_ϕϕ, g_tϕ) at (r, θ)."""
        raise NotImplementedError(f"Not implemented for {type(self).__name__}.")

    def inner_radius(self):
        raise NotImplementedError(f"Not implemented for {type(self).__name__}.")

    # metric(m, x) = _symmetric_matrix(comps); auto-diff.jl:228-232, utils.jl:60-67
    def metric(self, x):
        r, th = (x[1], x[2]) if len(x) == 4 else (x[0], x[1])
        g = self.metric_components(r, th)
        G = np.zeros((4, 4))
        G[0, 0], G[1, 1], G[2, 2], G[3, 3] = g[0], g[1], g[2], g[3]
        G[0, 3] = G[3, 0] = g[4]
        return G


class AbstractStaticAxisSymmetric(AbstractMetric):
    pass


@dataclass(frozen=True)
class KerrMetric(AbstractStaticAxisSymmetric):
    """KerrMetric(M = 1.0, a = 0.0) -- src/metrics/kerr-metric.jl:62-69."""

    M: float = 1.0
    a: float = 0.0
    metric_id = GR_METRIC_KERR

    def abi_params(self):
        return [self.M, self.a]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        """kerr-metric.jl:11-28; `r` may be a float or a special_radii.Jet."""
        M, a = self.M, self.a
        R = 2.0 * M
        s2 = s * s
        c2 = 1.0 - s2
        Sig = r * r + a * a * c2
        iSig = 1.0 / Sig
        gam = s2 * R * r * a
        tt = -(1.0 - (R * r) * iSig)
        rr = Sig / (r * r + a * a - R * r)
        pp = s2 * (r * r + a * a + (gam * a) * iSig)
        tp = -gam * iSig
        return (tt, rr, Sig, pp, tp)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        return kerr_isco(self.M, self.a)


@dataclass(frozen=True)
class JohannsenMetric(AbstractStaticAxisSymmetric):
    """JohannsenMetric(M, a, α13, α22, α52, ϵ3) -- src/metrics/johannsen-ad.jl:49-63."""

    M: float = 1.0
    a: float = 0.0
    alpha13: float = 0.0
    alpha22: float = 0.0
    alpha52: float = 0.0
    eps3: float = 0.0
    metric_id = GR_METRIC_JOHANNSEN

    def abi_params(self):
        return [self.M, self.a, self.alpha13, self.alpha22, self.alpha52, self.eps3]

    def metric_components(self, r, theta):
        return self._components(r, math.sin(theta), math.cos(theta))

    def _components(self, r, s, c):
        """johannsen-ad.jl:4-34; `r` may be a float or a special_radii.Jet."""
        M, a = self.M, self.a
        Mr = M / r
        A1 = 1.0 + self.alpha13 * (Mr * Mr * Mr)
        A2 = 1.0 + self.alpha22 * (Mr * Mr)
        A5 = 1.0 + self.alpha52 * (Mr * Mr)
        Sig = r * r + a * a * (c * c) + (self.eps3 * M ** 3) / r
        Del = r * r - 2.0 * M * r + a * a
        r2a2 = r * r + a * a
        s2 = s * s
        dn = r2a2 * A1 - (a * a * s2) * A2
        denom = dn * dn
        tt = -(Sig * (Del - (a * a * s2) * (A2 * A2)))
        rr = Sig / (Del * A5)
        pp = (Sig * s2) * ((r2a2 * r2a2) * (A1 * A1) - (a * a * s2) * Del)
        tp = -(a * ((Sig * s2) * (r2a2 * A1 * A2 - Del)))
        return (tt / denom, rr, Sig, pp / denom, tp / denom)

    def inner_radius(self):
        return self.M + math.sqrt(self.M ** 2 - self.a ** 2)

    def isco(self):
        from .special_radii import generic_isco

        return generic_isco(self)


def kerr_isco(M, a):
    """__BoyerLindquistFO.isco -- Bardeen et al. (1972) eq. 2.21; kerr-metric-first-order.jl:297-337."""
    x = a / M
    Z1 = 1.0 + np.cbrt(1.0 - x * x) * (np.cbrt(1.0 + x) + np.cbrt(1.0 - x))
    Z2 = math.sqrt(3.0 * x * x + Z1 * Z1)
    s = math.sqrt((3.0 - Z1) * (3.0 + Z1 + 2.0 * Z2))
    return float(M * (3.0 + Z2 - s) if a > 0.0 else M * (3.0 + Z2 + s))


def inner_radius(m):
    return m.inner_radius()


def isco(m):
    return m.isco()
