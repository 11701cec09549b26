"""PointFunction / FilterPointFunction and the built-ins of ConstPointFunctions.

Reference: src/point-functions.jl:44-47,74-79,81-129; src/const-point-functions.jl:26-79;
src/redshift.jl:225-276.  Built-ins carry a device tag so `rendergeodesics` can fuse them into
the trace kernel; arbitrary Python callables are applied on the host to device-traced endpoints
(the reference's generic `apply_to_image!`, rendering.jl:103-107).
"""
from __future__ import annotations

from typing import Callable, Optional

from .metrics import AbstractMetric, KerrMetric
from .status import StatusCodes

GR_PF_AFFINE_TIME, GR_PF_REDSHIFT, GR_PF_STATUS, GR_PF_RADIUS, GR_PF_WINDING = 0, 1, 2, 3, 4
GR_FILTER_NONE, GR_FILTER_EARLY_TERM, GR_FILTER_INTERSECTED = 0, 1, 2


class AbstractPointFunction:
    f: Callable
    device_pf: Optional[int] = None       # GR_PF_* when the value is a built-in
    device_filter: Optional[int] = None   # GR_FILTER_* when the filter is a built-in
    fill: float = float("nan")
    extra: Optional[dict] = None          # r_isco / plunging table for redshift

    def __call__(self, m, gp, max_time, **kw):
        return float(self.f(m, gp, max_time, **kw))

    # Julia's `pf1 ∘ pf2`
    def __matmul__(self, other):
        return compose(self, other)

    def compose(self, other):
        return compose(self, other)

    @property
    def fusable(self):
        return self.device_pf is not None


class PointFunction(AbstractPointFunction):
    def __init__(self, f, device_pf=None, device_filter=None, fill=float("nan"), extra=None):
        self.f = f
        self.device_pf = device_pf
        self.device_filter = device_filter
        self.fill = fill
        self.extra = extra


class FilterPointFunction(AbstractPointFunction):
    def __init__(self, f, default=float("nan"), device_filter=None):
        self.f = f
        self.default = default
        self.device_filter = device_filter
        self.device_pf = None


def compose(pf1: AbstractPointFunction, pf2: AbstractPointFunction) -> PointFunction:
    """point-functions.jl:103-127"""
    if isinstance(pf2, FilterPointFunction):
        def _f(m, gp, max_time, **kw):
            if pf2.f(m, gp, max_time, **kw):
                return pf1.f(m, gp, max_time, **kw)
            return pf2.default

        fus = pf1.device_pf is not None and pf1.device_filter is None and pf2.device_filter is not None
        return PointFunction(
            _f,
            device_pf=pf1.device_pf if fus else None,
            device_filter=pf2.device_filter if fus else None,
            fill=pf2.default,
            extra=pf1.extra,
        )

    def _g(m, gp, max_time, **kw):
        return pf1.f(m, gp, max_time, value=pf2.f(m, gp, max_time, **kw))

    return PointFunction(_g)


def FilterStatusCode(code, default=float("nan")):
    return FilterPointFunction(lambda m, gp, λ, **kw: gp["status"] == int(code), default)


class ConstPointFunctions:
    """const-point-functions.jl"""

    @staticmethod
    def filter_early_term():
        return FilterPointFunction(lambda m, gp, max_time, **kw: gp["lambda_max"] < max_time, float("nan"),
                                   device_filter=GR_FILTER_EARLY_TERM)

    @staticmethod
    def filter_intersected():
        return FilterPointFunction(
            lambda m, gp, max_time, **kw: gp["status"] == int(StatusCodes.IntersectedWithGeometry),
            float("nan"), device_filter=GR_FILTER_INTERSECTED)

    @staticmethod
    def affine_time():
        return PointFunction(lambda m, gp, max_time, **kw: gp["lambda_max"], device_pf=GR_PF_AFFINE_TIME)

    @staticmethod
    def winding():
        """PointFunction((m, gp, t) -> gp.aux.winding) for a TraceWindings render (photon-ring order)."""
        return PointFunction(lambda m, gp, max_time, **kw: float((int(gp["flags"]) & 0xFFFFFFFF) >> 16), device_pf=GR_PF_WINDING)

    @staticmethod
    def shadow():
        return ConstPointFunctions.affine_time() @ ConstPointFunctions.filter_early_term()

    @staticmethod
    def interpolate_redshift(plunging, u=None):
        """interpolate_redshift(plunging_interpolation, u) (redshift.jl:246-276): Keplerian outside the ISCO,
        the tabulated plunge inside, for any metric (Kerr included)."""
        m = plunging.m
        extra = {"r_isco": m.isco(), "plunge": tuple(plunging)}

        def _host(_m, gp, max_time, **_kw):
            raise NotImplementedError("redshift is evaluated on the device (gr_apply_pointfunction)")

        return PointFunction(_host, device_pf=GR_PF_REDSHIFT, extra=extra)

    @staticmethod
    def redshift(m: AbstractMetric, u=None, **kw):
        """redshift(::KerrMetric, _) = analytic; other metrics: interpolate_redshift(m, u)."""
        if isinstance(m, KerrMetric):
            extra = {"r_isco": m.isco(), "plunge": None}
        else:
            from .special_radii import interpolate_plunging_velocities

            extra = {"r_isco": m.isco(), "plunge": tuple(interpolate_plunging_velocities(m, **kw))}

        def _host(_m, gp, max_time, **_kw):
            raise NotImplementedError("redshift is evaluated on the device (gr_apply_pointfunction)")

        return PointFunction(_host, device_pf=GR_PF_REDSHIFT, extra=extra)
