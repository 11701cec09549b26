"""rendergeodesics / prerendergeodesics / apply -- src/rendering/{rendering,utility,cache}.jl."""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .pointfunctions import (GR_FILTER_NONE, AbstractPointFunction, ConstPointFunctions)
from .tracing import (EnsembleMI355X, RenderVelocity, TracingConfiguration, ensemble_solve_tracing_problem,
                      tracing_configuration)


def impact_axes(width, height, αlims, βlims):
    """rendering/utility.jl:43-47"""
    return np.linspace(αlims[0], αlims[1], width), np.linspace(βlims[0], βlims[1], height)


def _split_args(args):
    if len(args) == 2:
        return args[0], args[1]
    if len(args) == 1:
        return None, args[0]
    raise TypeError("expected ([disc], λ_max)")


def _unicode_kwargs(kw):
    for uni, asc in (("αlims", "alpha_lims"), ("βlims", "beta_lims")):
        if uni in kw:
            kw[asc] = kw.pop(uni)
    return kw


def render_configuration(m, position, *args, image_width, image_height, alpha_lims, beta_lims, **kwargs):
    """rendering.jl:1-26"""
    if not (alpha_lims[0] <= alpha_lims[1]):
        raise AssertionError("α limits must be sorted")
    if not (beta_lims[0] <= beta_lims[1]):
        raise AssertionError("β limits must be sorted")
    geometry, λmax = _split_args(args)
    vel = RenderVelocity(tuple(alpha_lims), tuple(beta_lims), int(image_width), int(image_height))
    return tracing_configuration(m, position, vel, geometry, λmax, trajectories=image_width * image_height, **kwargs)


def abi_pointfunction(pf: AbstractPointFunction):
    """Flatten a fusable built-in into gr_pointfunction; returns (struct, keepalive)."""
    s = _lib.gr_pointfunction()
    s.pf_id = pf.device_pf
    s.filter_id = pf.device_filter if pf.device_filter is not None else GR_FILTER_NONE
    s.fill = float(pf.fill)
    keep = []
    if pf.extra:
        s.r_isco = float(pf.extra.get("r_isco") or 0.0)
        tab = pf.extra.get("plunge")
        if tab is not None:
            arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in tab]
            keep = arrs
            s.n_plunge = arrs[0].size
            dp = C.POINTER(C.c_double)
            s.plunge_r, s.plunge_vt, s.plunge_vr, s.plunge_vphi = (a.ctypes.data_as(dp) for a in arrs)
    return s, keep


def _apply_host(pf, m, points, max_time):
    out = np.empty(points.shape[0])
    for i in range(points.shape[0]):
        out[i] = pf(m, points[i], max_time)
    return out


def apply_pointfunction(ensemble: EnsembleMI355X, config_or_cache, pf, points, max_time):
    """`apply_to_image!` (rendering.jl:103-107): device kernel for built-ins, host loop otherwise."""
    if pf.fusable:
        cfg = config_or_cache.abi_config()
        s, keep = abi_pointfunction(pf)
        pts = np.ascontiguousarray(points.ravel())
        out = np.zeros(pts.shape[0])
        _lib.check(_lib.load().gr_apply_pointfunction(ensemble.ctx.handle, C.byref(cfg), C.byref(s), pts.ctypes.data,
                                                      pts.shape[0], float(max_time), out.ctypes.data))
        return out
    return _apply_host(pf, config_or_cache.metric, points.ravel(), max_time)


def render_into_image(config: TracingConfiguration, pf=None, stats=False):
    """render_into_image! (rendering.jl:89-101).  Fused on the device when `pf` is a built-in."""
    if pf is None:
        pf = ConstPointFunctions.shadow()   # default pf of rendering.jl:93-94
    ens = config.ensemble
    rv = config.velocity
    n = rv.image_width * rv.image_height
    st = _lib.gr_stats()
    if pf.fusable:
        cfg, pl = config.abi_config(), config.abi_plane()
        s, keep = abi_pointfunction(pf)
        multi = ens.multi
        img = _lib.result_image(ens.ctx, n)      # >= 8 MiB: written by the kernel(s) across the link, each pixel at its place
        if multi:
            ctxs = ens.contexts_for_width(pl.width)
            arr, sts = _lib.ctx_array(ctxs)
            _lib.check(_lib.load().gr_render_multi(arr, len(ctxs), C.byref(cfg), C.byref(pl), C.byref(s), 0,
                                                   img.ctypes.data, sts))
            st = _lib.merge_stats(sts)
        else:
            rg = _lib.gr_range(0, n, max(n, 1), 1)
            _lib.check(_lib.load().gr_render(ens.ctx.handle, C.byref(cfg), C.byref(pl), C.byref(s), C.byref(rg),
                                             img.ctypes.data, C.byref(st)))
    else:
        pts = ensemble_solve_tracing_problem(ens, config)
        img = _apply_host(pf, config.metric, pts, config.λ_domain[1])
    # Julia image is (H, W) column-major with linear index i = x*H + y
    image = img.reshape(rv.image_width, rv.image_height).T
    return (image, st.asdict()) if stats else image


def rendergeodesics(m, position, *args, image_width=375, image_height=250, alpha_lims=(-60, 60), beta_lims=(-40, 40),
                    pf=None, ensemble=None, stats=False, **kwargs):
    """rendergeodesics(m, x, [d], λmax; image_width, image_height, αlims, βlims, pf, ensemble)
    -> (α, β, image) -- src/rendering/rendering.jl:28-54."""
    kwargs = _unicode_kwargs(kwargs)
    alpha_lims = kwargs.pop("alpha_lims", alpha_lims)
    beta_lims = kwargs.pop("beta_lims", beta_lims)
    config = render_configuration(m, position, *args, image_width=image_width, image_height=image_height,
                                  alpha_lims=alpha_lims, beta_lims=beta_lims, ensemble=ensemble, **kwargs)
    res = render_into_image(config, pf=pf, stats=stats)
    α, β = impact_axes(image_width, image_height, alpha_lims, beta_lims)
    if stats:
        return α, β, res[0], res[1]
    return α, β, res


class EndpointRenderCache:
    """rendering/cache.jl:40-52.  `points`: (height, width) GeodesicPoint records on the host.  With
    `prerendergeodesics(..., keep_on_device=True)` the records stay in HBM as well (`device_points`: a uint8 CUDA tensor of
    152 B per ray in image order) -- `apply` of a built-in point function then runs on them where they are, and the host
    copy is made only when `points` is first read."""

    def __init__(self, config, max_time, height, width, points=None, device_points=None):
        self.config, self.max_time, self.height, self.width = config, max_time, height, width
        self._points, self.device_points = points, device_points

    @property
    def points(self):
        if self._points is None:
            import torch

            n = self.height * self.width
            host = np.empty(n, dtype=_lib.POINT_DTYPE)
            torch.from_numpy(host.view(np.uint8)).copy_(self.device_points[: n * _lib.POINT_DTYPE.itemsize])
            self._points = host.reshape(self.width, self.height).T
        return self._points

    @property
    def m(self):
        return self.config.metric


def prerendergeodesics(m, position, *args, image_width=375, image_height=250, alpha_lims=(-60, 60),
                       beta_lims=(-40, 40), ensemble=None, keep_on_device=False, **kwargs):
    """prerendergeodesics -> (α, β, EndpointRenderCache) -- rendering.jl:56-87,121-138.  `keep_on_device`: leave the end
    points in HBM (152 B per ray) for `apply` of built-in point functions; the host copy is made on first use."""
    kwargs = _unicode_kwargs(kwargs)
    alpha_lims = kwargs.pop("alpha_lims", alpha_lims)
    beta_lims = kwargs.pop("beta_lims", beta_lims)
    config = render_configuration(m, position, *args, image_width=image_width, image_height=image_height,
                                  alpha_lims=alpha_lims, beta_lims=beta_lims, ensemble=ensemble, **kwargs)
    if keep_on_device:
        import torch

        from .device import render_endpoints_device

        ens = config.ensemble
        dev = torch.device("cuda", ens.device)
        raw = torch.empty(image_width * image_height * _lib.POINT_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        render_endpoints_device(config, raw)
        torch.cuda.synchronize(dev)
        cache = EndpointRenderCache(config, config.λ_domain[1], image_height, image_width, None, raw)
        α, β = impact_axes(image_width, image_height, alpha_lims, beta_lims)
        return α, β, cache
    pts = ensemble_solve_tracing_problem(config.ensemble, config)
    cache = EndpointRenderCache(config, config.λ_domain[1], image_height, image_width,
                                pts.reshape(image_width, image_height).T)
    α, β = impact_axes(image_width, image_height, alpha_lims, beta_lims)
    return α, β, cache


def apply(pf, rc: EndpointRenderCache, **kw):
    """apply(pf, cache) -- point-functions.jl:98-101"""
    if pf.fusable and rc.device_points is not None:
        # the records are in HBM already: k_apply_pf reads them there (0.1 ms at 2048² instead of uploading 637 MB)
        import torch

        from .device import _stream_handle

        n = rc.width * rc.height
        cfg = rc.config.abi_config()
        s, keep = abi_pointfunction(pf)
        out = torch.empty(n, dtype=torch.float64, device=rc.device_points.device)
        _lib.check(_lib.load().gr_apply_pointfunction_device(
            rc.config.ensemble.ctx.handle, C.byref(cfg), C.byref(s), C.c_void_p(rc.device_points.data_ptr()), n, float(rc.max_time),
            C.c_void_p(out.data_ptr()), _stream_handle()))
        return out.cpu().numpy().reshape(rc.width, rc.height).T
    pts = np.ascontiguousarray(rc.points.T).ravel()
    out = apply_pointfunction(rc.config.ensemble, rc.config, pf, pts, rc.max_time)
    return out.reshape(rc.width, rc.height).T
