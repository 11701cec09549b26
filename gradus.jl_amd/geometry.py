"""Accretion geometry available on the device.  ThinDisc -- src/geometry/discs/thin-disc.jl:9-26."""
from __future__ import annotations

from dataclasses import dataclass

GR_DISC_NONE, GR_DISC_THIN = 0, 1


class AbstractAccretionGeometry:
    pass


@dataclass(frozen=True)
class ThinDisc(AbstractAccretionGeometry):
    """ThinDisc(inner_radius = 0.0, outer_radius = 500.0)."""

    inner_radius: float = 0.0
    outer_radius: float = 500.0
    disc_id = GR_DISC_THIN
