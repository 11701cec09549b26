"""Image planes and grids -- src/image-planes/planes.jl:70-184, src/image-planes/grids.jl:11-36."""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np


class Abstract2DGrid:
    pass


class LinearGrid(Abstract2DGrid):
    def __call__(self, lo, hi, N):
        return np.linspace(lo, hi, N)


class GeometricGrid(Abstract2DGrid):
    def __call__(self, lo, hi, N):
        K = (hi / lo) ** (1.0 / (N - 1))
        return np.array([lo * K ** i for i in range(N)])


class InverseGrid(Abstract2DGrid):
    def __call__(self, lo, hi, N):
        return np.array([1.0 / x for x in np.linspace(1.0 / hi, 1.0 / lo, N)][::-1])


class AbstractImagePlane:
    pass


@dataclass(frozen=True)
class PolarPlane(AbstractImagePlane):
    grid: Abstract2DGrid
    Nr: int = 400
    Nθ: int = 100
    r_min: float = 1.0
    r_max: float = 250.0
    θ_min: float = 0.0
    θ_max: float = 2 * math.pi


@dataclass(frozen=True)
class CartesianPlane(AbstractImagePlane):
    grid: Abstract2DGrid
    Nx: int = 150
    Ny: int = 150
    x_min: float = 0.0
    x_max: float = 150.0
    y_min: float = 0.0
    y_max: float = 150.0


def trajectory_count(plane):
    if isinstance(plane, PolarPlane):
        return plane.Nr * plane.Nθ
    return (2 * (plane.Ny // 2) - 1) * (2 * (plane.Nx // 2) - 1)


def image_plane(plane, x=None):
    """Returns (αs, βs) as 2-D arrays indexed like the reference's matrices (Julia column-major)."""
    if isinstance(plane, PolarPlane):
        rs = np.asarray(plane.grid(plane.r_min, plane.r_max, plane.Nr))
        dθ = (plane.θ_max - plane.θ_min) / plane.Nθ
        θs = np.linspace(plane.θ_min, plane.θ_max - dθ, plane.Nθ)
        return rs[:, None] * np.cos(θs)[None, :], rs[:, None] * np.sin(θs)[None, :]
    xs = np.asarray(plane.grid(plane.x_min, plane.x_max, plane.Nx // 2))
    ys = np.asarray(plane.grid(plane.y_min, plane.y_max, plane.Ny // 2))
    X_size = 2 * (plane.Ny // 2) - 1
    Y_size = 2 * (plane.Nx // 2) - 1
    X = np.tile(xs[1:][None, :], (X_size, 1))
    Y = np.tile(ys[1:][:, None], (1, Y_size))
    αs = np.hstack([-X[:, ::-1], np.full((X_size, 1), xs[0]), X])
    βs = np.vstack([-Y[::-1, :], np.full((1, Y_size), ys[0]), Y])
    return αs, βs


def impact_parameters(plane, x=None):
    αs, βs = image_plane(plane, x)
    # vec() of a Julia matrix is column-major
    return αs.ravel(order="F"), βs.ravel(order="F")


def unnormalized_areas(plane):
    if isinstance(plane, PolarPlane):
        rs = np.asarray(plane.grid(plane.r_min, plane.r_max, plane.Nr))
        return np.repeat((rs ** 2)[:, None], plane.Nθ, axis=1)
    if isinstance(plane.grid, LinearGrid):
        X_size = 2 * (plane.Ny // 2) - 1
        Y_size = 2 * (plane.Nx // 2) - 1
        return np.ones((Y_size, X_size))
    raise NotImplementedError
