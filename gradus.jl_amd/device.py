"""Device-resident entry points: results stay in HBM (torch tensors), launches are asynchronous
on torch's current HIP stream.  PyTorch is plumbing here (allocation, streams, RCCL); all
compute is in libgradus_mi355x.so."""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import _lib
from .rendering import abi_pointfunction
from .tracing import TracingConfiguration


def _stream_handle():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def new_stats(device):
    import torch

    return torch.zeros(11, dtype=torch.int64, device=device)   # gr_stats: 9 counters + kernel_ms + call_ms slots


def stats_dict(t):
    h = t.cpu().numpy()
    return {
        "rays": int(h[0]), "accepted_steps": int(h[1]), "rejected_steps": int(h[2]), "rhs_evals": int(h[3]),
        "flagged_rays": int(h[4]), "status_count": [int(x) for x in h[5:9]],
    }


def render_device(config: TracingConfiguration, pf, out, ray_range=None, stats=None):
    """gr_render_device: fused PointFunction image of `ray_range` into the CUDA tensor `out`
    (float64, at least range.count elements).  Asynchronous on the current stream."""
    import torch

    assert out.is_cuda and out.dtype == torch.float64 and out.is_contiguous()
    if not pf.fusable:
        raise NotImplementedError("only built-in point functions are fused on the device")
    cfg, pl = config.abi_config(), config.abi_plane()
    n = pl.width * pl.height
    rg = ray_range if ray_range is not None else _lib.gr_range(0, n, max(n, 1), 1)
    assert out.numel() >= rg.count
    s, keep = abi_pointfunction(pf)
    ens = config.ensemble
    _lib.check(_lib.load().gr_render_device(
        ens.ctx.handle, C.byref(cfg), C.byref(pl), C.byref(s), C.byref(rg), C.c_void_p(out.data_ptr()),
        C.c_void_p(stats.data_ptr()) if stats is not None else None, _stream_handle()))
    return out


def render_endpoints_device(config: TracingConfiguration, out, ray_range=None, stats=None):
    """gr_render_endpoints_device into a uint8 CUDA tensor of 152 * count bytes."""
    import torch

    assert out.is_cuda and out.dtype == torch.uint8 and out.is_contiguous()
    cfg, pl = config.abi_config(), config.abi_plane()
    n = pl.width * pl.height
    rg = ray_range if ray_range is not None else _lib.gr_range(0, n, max(n, 1), 1)
    assert out.numel() >= rg.count * _lib.POINT_DTYPE.itemsize
    ens = config.ensemble
    _lib.check(_lib.load().gr_render_endpoints_device(
        ens.ctx.handle, C.byref(cfg), C.byref(pl), C.byref(rg), C.c_void_p(out.data_ptr()),
        C.c_void_p(stats.data_ptr()) if stats is not None else None, _stream_handle()))
    return out


def points_from_tensor(t, count):
    return np.frombuffer(t.cpu().numpy().tobytes(), dtype=_lib.POINT_DTYPE, count=count)


def lineprofile_device(bins, ε, m, u, d, plane, *, shard=None, maxrₑ=50.0, minrₑ=None, λ_max=None, redshift_pf=None,
                       callback="default", ensemble=None, flux=None, stats=None, **solver_args):
    """gr_lineprofile_device: the un-normalised BinningMethod histogram of (a shard of) a PolarPlane's rays, accumulated
    into a float64 CUDA tensor of bins.size entries (`flux` if given, else a new one; the launch zeroes it first).  `shard`: a distributed.RayShard (block-cyclic sub-range of the plane's rays) or None = all of them.
    Asynchronous on the current stream; the fused route: power laws and emissivity profiles (tables)."""
    import torch

    from .lineprofiles import PowerLawEmissivity, _emissivity_table
    from .planes import PolarPlane
    from .pointfunctions import ConstPointFunctions
    from .tracing import domain_upper_hemisphere, separable_rayset, tracing_configuration

    table = None if isinstance(ε, PowerLawEmissivity) else _emissivity_table(ε)
    if not (isinstance(ε, PowerLawEmissivity) or table is not None) or not isinstance(plane, PolarPlane):
        raise NotImplementedError("the device-resident line profile is the fused route: a PowerLawEmissivity or an emissivity "
                                  "profile (RadialDiscProfile) on a PolarPlane")
    u = np.asarray(u, dtype=np.float64)
    bins = np.ascontiguousarray(bins, dtype=np.float64)
    λ_max = 2.0 * u[1] if λ_max is None else λ_max
    minrₑ = m.isco() if minrₑ is None else minrₑ
    if callback == "default":
        callback = domain_upper_hemisphere()
    if redshift_pf is None:
        redshift_pf = ConstPointFunctions.redshift(m, u, **({"ensemble": ensemble} if m.metric_id != 0 else {}))
    config = tracing_configuration(m, u, np.zeros((1, 4)), d, (0.0, λ_max), callback=callback, ensemble=ensemble, **solver_args)
    cfg = config.abi_config()
    ens = config.ensemble
    dev = torch.device("cuda", ens.device)
    # always the separable form here (three device-resident tables), whatever GRADUS_MI355X_SEPARABLE_RAYS says for
    # the host route: the pointers set below are only meaningful for it
    tiled = os.environ.get("GRADUS_MI355X_TILE_RAYS", "1") != "0" and plane.Nr >= 8 and plane.Nθ >= 8
    rs, keep = separable_rayset(config.metric, config.position, plane, tiled)
    rs._tiled = tiled
    # the three tables and the bin edges live in HBM for the launch (tiny: Nr + 2 Nθ + n_bins doubles)
    extra = [] if table is None else [table[0], table[1]]
    tabs = torch.from_numpy(np.concatenate([keep[0], keep[1], keep[2], bins] + extra)).to(dev)
    nr, nt = plane.Nr, plane.Nθ
    base = tabs.data_ptr()
    rs.sep_r, rs.sep_cos, rs.sep_sin = base, base + 8 * nr, base + 8 * (nr + nt)
    if shard is not None:
        rs.sep_first, rs.sep_block, rs.sep_stride, rs.n = shard.first, shard.block, shard.stride, shard.count
    if flux is None:
        flux = torch.zeros(bins.size, dtype=torch.float64, device=dev)
    assert flux.is_cuda and flux.dtype == torch.float64 and flux.numel() == bins.size
    b = _lib.gr_binning(float(minrₑ), float(maxrₑ), ε.q if table is None else 0.0, bins.size, base + 8 * (nr + 2 * nt))
    if table is not None:
        off = base + 8 * (nr + 2 * nt + bins.size)
        b.eps_r, b.eps_v, b.eps_n = off, off + 8 * table[0].size, table[0].size
    pf, keep_pf = abi_pointfunction(redshift_pf)
    lane = rs._tiled and ens.knobs.get("kernel", 2) == 2 and max(config.abstol, config.reltol) <= 1e-6
    if lane:
        ens.ctx.set("kernel", 0)
    try:
        if rs.n > 0:
            _lib.check(_lib.load().gr_lineprofile_device(
                ens.ctx.handle, C.byref(cfg), C.byref(rs), C.byref(pf), C.byref(b), C.c_void_p(flux.data_ptr()),
                C.c_void_p(stats.data_ptr()) if stats is not None else None, _stream_handle()))
    finally:
        if lane:
            ens.ctx.set("kernel", 2)
    flux._gradus_keep = (tabs, keep_pf)          # the launch reads them asynchronously
    return flux
