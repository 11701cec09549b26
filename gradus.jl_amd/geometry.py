"""Accretion geometry available on the device.  ThinDisc -- src/geometry/discs/thin-disc.jl:9-26."""
from __future__ import annotations

from dataclasses import dataclass

GR_DISC_NONE, GR_DISC_THIN, GR_DISC_SHAKURA_SUNYAEV = 0, 1, 2


class AbstractAccretionGeometry:
    pass


@dataclass(frozen=True)
class ThinDisc(AbstractAccretionGeometry):
    """ThinDisc(inner_radius = 0.0, outer_radius = 500.0)."""

    inner_radius: float = 0.0
    outer_radius: float = 500.0
    disc_id = GR_DISC_THIN


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
