"""Host-side set-up for non-Kerr redshift: generic ISCO and the plunging-velocity table.

Reference: src/special-radii.jl:14-60 (isco root find), src/orbits/circular-orbits.jl:11-48
(Ω, u_t, u_ϕ, energy), src/orbits/orbit-solving.jl:99-167 (PlungingInterpolation).
Implemented in round 2 of the build; Kerr needs none of this.
"""
from __future__ import annotations


def generic_isco(m):
    raise NotImplementedError("generic isco(m) root find is scheduled after the Kerr path (SURVEY §8 a18)")


def interpolate_plunging_velocities(m, **kw):
    raise NotImplementedError("PlungingInterpolation is scheduled after the Kerr path (SURVEY §8 a18)")
