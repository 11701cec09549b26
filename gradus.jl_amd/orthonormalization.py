"""LNRF tetrads by Gram-Schmidt -- host-side, once per render.

Follows src/orthonormalization.jl:3-123 step by step (including its `sum(p) > tol` loop test
and the tetrad permutation bookkeeping) so the observer basis is the one the reference builds.
"""
from __future__ import annotations

import numpy as np


def dotproduct(g, v1, v2):
    # dotproduct(g, v1, v2) = _fast_dot(g * v1, v2)   :3-16
    return float(np.dot(g @ v1, v2))


def propernorm(g, v):
    return dotproduct(g, v, v)


def mproject(g, v, u):
    return dotproduct(g, v, u) / propernorm(g, u)


def projectbasis(g, basis, v):
    s = np.zeros(4)
    for e in basis:
        s = s + mproject(g, v, e) * e
    return s


def gramschmidt(v, basis, g, tol=4 * np.finfo(np.float64).eps):
    v = np.array(v, dtype=np.float64)
    p = projectbasis(g, basis, v)
    guard = 0
    while p.sum() > tol and guard < 1000:
        v = v - p
        p = projectbasis(g, basis, v)
        guard += 1
    v = v - p
    return v / np.sqrt(abs(propernorm(g, v)))


def _tetrad_permute(x):
    return (x[0], x[3], x[1], x[2])


def tetradframe(g, v):
    """tetradframe(g, v) :75-103; returns the four vectors ordered (t, r, θ, ϕ)."""
    v = np.asarray(v, dtype=np.float64)
    v1 = v / np.sqrt(abs(propernorm(g, v)))
    state = [bool(c != 0) for c in v1]
    if sum(state) == 1:
        state = [True, False, False, True]
    # permutations = searchsortedfirst(state[2:end], 1)
    permutations = 4
    for i in range(1, 4):
        if state[i]:
            permutations = i
            break
    v2 = gramschmidt(np.array(state, dtype=np.float64), (v1,), g)
    state = [a or b for a, b in zip(state, _tetrad_permute(state))]
    v3 = gramschmidt(np.array(state, dtype=np.float64), (v1, v2), g)
    state = [a or b for a, b in zip(state, _tetrad_permute(state))]
    v4 = gramschmidt(np.array(state, dtype=np.float64), (v1, v2, v3), g)
    ret = (v1, v2, v3, v4)
    for _ in range(2, permutations + 1):
        ret = _tetrad_permute(ret)
    return ret


def lnrframe(g):
    """Tetrad with latin indices down :106-111."""
    om = -g[0, 3] / g[3, 3]
    return tetradframe(g, np.array([1.0, 0.0, 0.0, om]))


def lnrbasis(g):
    """Tetrad with latin indices up :114-122; returns (vt, vr, vθ, vϕ)."""
    om = -g[0, 3] / g[3, 3]
    vphi, vr, vth, vt = tetradframe(np.linalg.inv(g), np.array([-om, 0.0, 0.0, 1.0]))
    return (vt, vr, vth, vphi)


def lnrbasis_matrix(m, x):
    return np.column_stack(lnrbasis(m.metric(x)))


def lnrframe_matrix(m, x):
    return np.column_stack(lnrframe(m.metric(x)))
