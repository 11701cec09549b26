"""Device-resident entry points: results stay in HBM (torch tensors), launches are asynchronous
on torch's current HIP stream.  PyTorch is plumbing here (allocation, streams, RCCL); all
compute is in libgradus_mi355x.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from .rendering import abi_pointfunction
from .tracing import TracingConfiguration


def _stream_handle():
    import torch

    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def new_stats(device):
    import torch

    return torch.zeros(10, dtype=torch.int64, device=device)   # gr_stats: 9 counters + kernel_ms slot


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
