"""Accretion geometry available on the device.  ThinDisc -- src/geometry/discs/thin-disc.jl:9-26."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np

GR_DISC_NONE, GR_DISC_THIN, GR_DISC_SHAKURA_SUNYAEV, GR_DISC_TABULATED, GR_DISC_DATUM = 0, 1, 2, 3, 4
GR_DISC_ELLIPTICAL, GR_DISC_PRECESSING_THIN, GR_DISC_COMPOSITE, GR_DISC_MESH = 5, 6, 7, 8


class AbstractAccretionGeometry:
    def __matmul__(self, other):
        """d1 @ d2 = the reference's d1 ∘ d2 = CompositeGeometry(d1, d2) (src/geometry/composite.jl:24-25)."""
        if not isinstance(other, AbstractAccretionGeometry):
            return NotImplemented
        mine = self.geometry if isinstance(self, CompositeGeometry) else (self,)
        theirs = other.geometry if isinstance(other, CompositeGeometry) else (other,)
        return CompositeGeometry(*mine, *theirs)


@dataclass(frozen=True)
class ThinDisc(AbstractAccretionGeometry):
    """ThinDisc(inner_radius = 0.0, outer_radius = 500.0)."""

    inner_radius: float = 0.0
    outer_radius: float = 500.0
    disc_id = GR_DISC_THIN


@dataclass(frozen=True)
class DatumPlane(AbstractAccretionGeometry):
    """DatumPlane(height) -- src/geometry/discs/datum-plane.jl:1-10: the surface r cosθ = height, hit
    from above only (signed distance, no gtol, no radial extent)."""

    height: float = 0.0
    disc_id = GR_DISC_DATUM
    inner_radius = 0.0


@dataclass(frozen=True)
class EllipticalDisc(AbstractAccretionGeometry):
    """EllipticalDisc(inner_radius, semi_major, semi_minor) -- src/geometry/discs.jl:57-72: a disc whose
    half-thickness follows the ellipse sqrt((1 - (r/semi_major)²) semi_minor²)."""

    inner_radius: float
    semi_major: float
    semi_minor: float
    disc_id = GR_DISC_ELLIPTICAL


@dataclass(frozen=True)
class PrecessingDisc(AbstractAccretionGeometry):
    """PrecessingDisc(disc, β, γ) -- src/geometry/discs.jl:74-96: `disc` tilted by β about the x axis and turned by γ
    about the spin axis.  The device implements it around a ThinDisc."""

    disc: ThinDisc
    β: float
    γ: float
    disc_id = GR_DISC_PRECESSING_THIN

    def __post_init__(self):
        if not isinstance(self.disc, ThinDisc):
            raise NotImplementedError("PrecessingDisc runs on the device around a ThinDisc")

    @property
    def inner_radius(self):
        return self.disc.inner_radius


@dataclass(frozen=True)
class ShakuraSunyaev(AbstractAccretionGeometry):
    """ShakuraSunyaev(m; eddington_ratio = 0.3, η = nothing) -- src/geometry/discs/shakura-sunyaev.jl:22-57.
    Height 2H with H = (3/2)(1/η)(Ṁ/Ṁ_Edd)(1 - sqrt(r_isco/ρ)); η defaults to 1 - E_isco."""

    Ṁ_Ṁedd: float
    inv_η: float
    inner_radius: float
    disc_id = GR_DISC_SHAKURA_SUNYAEV

    @staticmethod
    def for_metric(m, eddington_ratio=0.3, η=None):
        from .special_radii import _energy_jet

        r_isco = m.isco()
        if η is None:
            η = 1.0 - _energy_jet(m, r_isco)[0]
        return ShakuraSunyaev(float(eddington_ratio), 1.0 / η, r_isco)

    def cross_section(self, ρ):
        if ρ < self.inner_radius:
            return -0.0
        return 3.0 * self.inv_η * self.Ṁ_Ṁedd * (1.0 - (self.inner_radius / ρ) ** 0.5)


class ThickDisc(AbstractAccretionGeometry):
    """ThickDisc(f; inner_radius = 0, outer_radius = Inf) -- src/geometry/discs/thick-disc.jl:30-58.

    `f(ρ)` is the height cross-section (<= 0 where there is no disc).  A Python closure cannot run on
    the device, so it is sampled on a uniform grid over `ρ_range` (`samples` points, linear
    interpolation on the device); choose the range to cover the region where f > 0."""

    disc_id = GR_DISC_TABULATED

    def __init__(self, f, *, inner_radius=0.0, outer_radius=float("inf"), ρ_range=(0.0, 100.0), samples=16384):
        if isinstance(f, (int, float)):
            raise TypeError("Invalid constructor (you probably meant ThinDisc, not ThickDisc).")
        self.f = f
        self.inner_radius, self.outer_radius = float(inner_radius), float(outer_radius)
        self.ρ_range = (float(ρ_range[0]), float(ρ_range[1]))
        ρ = np.linspace(self.ρ_range[0], self.ρ_range[1], int(samples))
        self.table = np.ascontiguousarray([float(f(r)) for r in ρ], dtype=np.float64)

    def cross_section(self, ρ):
        return self.f(ρ)


class WarpedThinDisc(AbstractAccretionGeometry):
    """WarpedThinDisc(f; inner_radius = 0, outer_radius = 500) -- src/geometry/discs/thin-disc.jl:28-66: a thin sheet
    at the signed height f(ρ) above the equatorial plane.  Like ThickDisc(f), the closure is sampled on a uniform ρ
    grid by the host (here over [inner_radius, outer_radius]) and interpolated on the device."""

    disc_id = GR_DISC_TABULATED

    def __init__(self, f, *, inner_radius=0.0, outer_radius=500.0, samples=16384):
        self.f = f
        self.inner_radius, self.outer_radius = float(inner_radius), float(outer_radius)
        self.ρ_range = (self.inner_radius, self.outer_radius)
        ρ = np.linspace(self.ρ_range[0], self.ρ_range[1], int(samples))
        self.table = np.ascontiguousarray([float(f(r)) for r in ρ], dtype=np.float64)


class CompositeGeometry(AbstractAccretionGeometry):
    """CompositeGeometry(d1, d2, ...) = d1 ∘ d2 ∘ ... -- src/geometry/composite.jl:1-26.  A ray ends at the earliest
    intersection with any of its geometries (the VectorContinuousCallback of geometry/bootstrap.jl:76-110).  On the device:
    2 to 4 components, each a ThinDisc, ShakuraSunyaev, EllipticalDisc or DatumPlane."""

    disc_id = GR_DISC_COMPOSITE

    def __init__(self, *geometry):
        if not geometry:
            raise ValueError("Must provide at least one disc as argument to constructor")      # composite.jl:1-3
        self.geometry = tuple(geometry)

    def __len__(self):
        return len(self.geometry)

    def __iter__(self):
        return iter(self.geometry)

    @property
    def inner_radius(self):
        return min(getattr(g, "inner_radius", 0.0) for g in self.geometry)


def bounding_box(mesh):
    """bounding_box(mesh) -- src/geometry/meshes.jl:24-44: ((x_min, x_max), (y_min, y_max), (z_min, z_max)) over every vertex."""
    pts = np.asarray(mesh, dtype=np.float64).reshape(-1, 3)
    lo, hi = pts.min(axis=0), pts.max(axis=0)
    return (float(lo[0]), float(hi[0])), (float(lo[1]), float(hi[1])), (float(lo[2]), float(hi[2]))


class MeshAccretionGeometry(AbstractAccretionGeometry):
    """MeshAccretionGeometry(mesh) -- src/geometry/meshes.jl:1-80.  `mesh`: triangles as an (n, 3, 3) array (or any nesting
    of n triangles x 3 vertices x (x, y, z)) in the Cartesian coordinates (r sinθ cosϕ, r sinθ sinϕ, r cosθ) of geometry.jl:13-16.
    Fields as in the reference: `mesh`, `x_extent`, `y_extent`, `z_extent` (the bounding box of the constructor; pass your own
    extents to narrow the region in which the triangles are tested at all).  A DiscreteCallback: after every accepted step the
    segment from the previous position to the new one is tested with the Jiménez-Segura-Feito algorithm (intersections.jl:58-101)
    against the triangles whose first vertex is within 3 of the new position; the ray ends at the step's end, front faces only.
    On the device the triangle list is walked once per wave and accepted step inside the box (fp64 kernels only)."""

    disc_id = GR_DISC_MESH

    def __init__(self, mesh, x_extent=None, y_extent=None, z_extent=None):
        tri = np.ascontiguousarray(mesh, dtype=np.float64).reshape(-1, 3, 3)
        if tri.shape[0] < 1:
            raise ValueError("a mesh needs at least one triangle")
        self.mesh = tri
        bx, by, bz = bounding_box(tri)
        self.x_extent = tuple(map(float, x_extent)) if x_extent is not None else bx
        self.y_extent = tuple(map(float, y_extent)) if y_extent is not None else by
        self.z_extent = tuple(map(float, z_extent)) if z_extent is not None else bz
        # what crosses the ABI (gr_config.disc_table): the six extents, then 9 doubles per triangle
        self.table = np.ascontiguousarray(np.concatenate([[*self.x_extent, *self.y_extent, *self.z_extent], tri.ravel()]), dtype=np.float64)

    def __len__(self):
        return self.mesh.shape[0]
