"""LNRF tetrads by Gram-Schmidt -- host-side, once per render.

Follows src/orthonormalization.jl:3-123 step by step (including its `sum(p) > tol` loop test
and the tetrad permutation bookkeeping) so the observer basis is the one the reference builds.
"""
from __future__ import annotations

import numpy as np


def dotproduct(g, v1, v2):
    # dotproduct(g, v1, v2) = _fast_dot(g * v1, v2)   :3-16
    # (sums written out in one fixed order -- the order _bdot below uses on arrays: a tetrad built sample by sample and the same
    # tetrad built for many samples at once are then the same bits, whatever BLAS would have made of a 4 x 4 product)
    s = 0.0
    for i in range(4):
        gi = g[i]
        s = s + (((gi[0] * v1[0] + gi[1] * v1[1]) + gi[2] * v1[2]) + gi[3] * v1[3]) * v2[i]
    return float(s)


def propernorm(g, v):
    return dotproduct(g, v, v)


def mproject(g, v, u):
    return dotproduct(g, v, u) / propernorm(g, u)


def projectbasis(g, basis, v):
    s = np.zeros(4)
    for e in basis:
        s = s + mproject(g, v, e) * e
    return s


# The reference's loop (`while sum(p) > tol`, :37-47) tests the SUM of the projection's components: once they are rounding noise
# (±5e-16 against tol = 9e-16) the sign pattern of that noise decides whether it ends, and for some inputs it never does.  A pass
# changes v by an ulp; sixteen are the same vector as a thousand.
kGuard = 16


def gramschmidt(v, basis, g, tol=4 * np.finfo(np.float64).eps):
    v = np.array(v, dtype=np.float64)
    p = projectbasis(g, basis, v)
    guard = 0
    while ((p[0] + p[1]) + p[2]) + p[3] > tol and guard < kGuard:
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


# ---- the same, for MANY (g, v) at once (a corona without one position: a tetrad per sample) ----
# Inside, arrays are COMPONENT-major -- g (4, 4, n), vectors (4, n) -- so that every operand of the sums below is a contiguous vector.
def _bdot(g, a, b):
    # dotproduct per sample, in dotproduct's order of operations
    s = 0.0
    for i in range(4):
        s = s + (((g[i][0] * a[0] + g[i][1] * a[1]) + g[i][2] * a[2]) + g[i][3] * a[3]) * b[i]
    return s


def _bproject(g, basis, v):
    s = np.zeros_like(v)
    for e in basis:
        s = s + (_bdot(g, v, e) / _bdot(g, e, e)) * e
    return s


def _bgramschmidt(v, basis, g, tol=4 * np.finfo(np.float64).eps):
    """gramschmidt() per sample: the `sum(p) > tol` loop runs for the samples that still ask for it."""
    v = np.array(v, dtype=np.float64)
    p = _bproject(g, basis, v)
    psum = lambda q: ((q[0] + q[1]) + q[2]) + q[3]
    active = psum(p) > tol
    guard = 0
    while active.any() and guard < kGuard:
        idx = np.nonzero(active)[0]
        v[:, idx] = v[:, idx] - p[:, idx]
        p[:, idx] = _bproject(g[:, :, idx], [e[:, idx] for e in basis], v[:, idx])
        active[idx] = psum(p[:, idx]) > tol
        guard += 1
    v = v - p
    return v / np.sqrt(np.abs(_bdot(g, v, v)))


def tetradframe_batch(g, v):
    """tetradframe(g[k], v[k]) for every k: g (n, 4, 4), v (n, 4) -> four arrays (n, 4) ordered (t, r, θ, ϕ).  Samples are grouped by
    which components of v vanish (the branch structure of :75-103 depends on nothing else)."""
    g = np.ascontiguousarray(np.asarray(g, dtype=np.float64).transpose(1, 2, 0))
    v = np.ascontiguousarray(np.asarray(v, dtype=np.float64).T)
    n = v.shape[1]
    v1 = v / np.sqrt(np.abs(_bdot(g, v, v)))
    out = [np.zeros((4, n)) for _ in range(4)]
    code = np.array([1, 2, 4, 8]) @ (v1 != 0)          # which components vanish, as one integer per sample
    for pc in np.unique(code):
        sel = np.nonzero(code == pc)[0]
        state = [bool(pc & (1 << q)) for q in range(4)]
        if sum(state) == 1:
            state = [True, False, False, True]
        permutations = 4
        for i in range(1, 4):
            if state[i]:
                permutations = i
                break
        whole = sel.size == n
        gs, e1 = (g, v1) if whole else (np.ascontiguousarray(g[:, :, sel]), np.ascontiguousarray(v1[:, sel]))
        start = lambda st: np.repeat(np.array(st, dtype=np.float64)[:, None], sel.size, axis=1)
        e2 = _bgramschmidt(start(state), (e1,), gs)
        state = [a or b for a, b in zip(state, _tetrad_permute(state))]
        e3 = _bgramschmidt(start(state), (e1, e2), gs)
        state = [a or b for a, b in zip(state, _tetrad_permute(state))]
        e4 = _bgramschmidt(start(state), (e1, e2, e3), gs)
        ret = (e1, e2, e3, e4)
        for _ in range(2, permutations + 1):
            ret = _tetrad_permute(ret)
        for q in range(4):
            out[q][:, sel] = ret[q]
    return tuple(np.ascontiguousarray(o.T) for o in out)
