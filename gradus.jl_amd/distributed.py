"""Sharding the image plane over the GPUs of one node (one process per GPU).

Rays are independent (the reference fans them out over threads, src/tracing/tracing.jl:186), so
each rank traces its own rays with no exchange; one RCCL gather over xGMI at the end assembles
the H x W image on rank 0.  Rays are dealt block-cyclically in groups of `block_cols` image
columns so every rank gets a similar mix of short (captured / disc-hit) and long (escaping) rays.
"""
from __future__ import annotations

import os
from dataclasses import dataclass

from . import _lib


@dataclass(frozen=True)
class ShardPlan:
    width: int
    height: int
    world: int
    rank: int
    block_cols: int

    @property
    def block(self) -> int:          # rays per block
        return self.block_cols * self.height

    @property
    def n_blocks(self) -> int:       # blocks per rank
        return self.width // (self.block_cols * self.world)

    @property
    def count(self) -> int:          # rays of this rank
        return self.n_blocks * self.block

    def ray_range(self) -> _lib.gr_range:
        return _lib.gr_range(self.rank * self.block, self.count, self.block, self.world)

    def global_index(self, j):
        """image ray index of local ray j (same formula as gr_range in the C ABI)."""
        b = j // self.block
        return self.rank * self.block + b * self.world * self.block + (j - b * self.block)


def shard_plan(width: int, height: int, world: int, rank: int, block_cols: int = 8) -> ShardPlan:
    bc = block_cols
    while bc > 1 and width % (bc * world) != 0:
        bc //= 2
    if width % (bc * world) != 0:
        raise ValueError(f"image width {width} cannot be dealt in column blocks over {world} ranks")
    return ShardPlan(width, height, world, rank, bc)


def gather_image(local, plan: ShardPlan, group=None, dst: int = 0):
    """One collective: gather every rank's compact slab on `dst` and undo the block-cyclic deal.
    `local` is a 1-D tensor of plan.count values in local ray order.  Returns the (H, W) image on
    `dst` (None elsewhere).  Works with RCCL ("nccl") on GPUs and gloo on CPU tensors."""
    import torch
    import torch.distributed as dist

    if plan.world == 1:
        full = local
    else:
        bufs = [torch.empty_like(local) for _ in range(plan.world)] if plan.rank == dst else None
        dist.gather(local, bufs, dst=dst, group=group)
        if plan.rank != dst:
            return None
        full = torch.stack(bufs)                                   # [world, n_blocks * block]
        full = full.view(plan.world, plan.n_blocks, plan.block).permute(1, 0, 2).reshape(-1)
    # linear index i = x*H + y  ->  Julia's (H, W) column-major matrix
    return full.view(plan.width, plan.height).t()


def gather_points(local, plan: ShardPlan, group=None, dst: int = 0):
    """The end points of a sharded plane (prerendergeodesics / the generic boundary with one process per GPU): every rank
    holds its `plan.count` GeodesicPoint records in local ray order as a uint8 tensor of 152 bytes per ray; ONE gather brings
    them to `dst`, one permute undoes the block-cyclic deal.  Returns a uint8 tensor of 152 * W * H bytes in image order
    (ray i = x H + y, the order of gr_render_endpoints) on `dst`, None elsewhere.  RCCL on GPU tensors, gloo on CPU ones.

    This is the one exchange of the whole path whose size matters: 637 MB for a 2048² plane, 7/8 of it crossing xGMI into
    rank 0 over seven point-to-point links at ≈153 GB/s each: ≈ 0.5 ms if the seven senders arrive together, against ≈ 18 / 8 ms
    of tracing per rank."""
    import torch
    import torch.distributed as dist

    rec = _lib.POINT_DTYPE.itemsize
    assert local.dtype == torch.uint8 and local.numel() == plan.count * rec
    if plan.world == 1:
        return local
    bufs = [torch.empty_like(local) for _ in range(plan.world)] if plan.rank == dst else None
    dist.gather(local, bufs, dst=dst, group=group)
    if plan.rank != dst:
        return None
    full = torch.stack(bufs)                                   # [world, n_blocks * block * 152]
    return full.view(plan.world, plan.n_blocks, plan.block * rec).permute(1, 0, 2).reshape(-1)


class PendingGather:
    """Handle of an in-flight gather started by `gather_image_async`.  `result()` waits for the
    collective (making the current stream wait, for RCCL) and returns the (H, W) image on `dst`."""

    def __init__(self, work, bufs, local, plan, dst):
        self.work, self.bufs, self.local, self.plan, self.dst = work, bufs, local, plan, dst

    def result(self):
        import torch

        if self.work is not None:
            self.work.wait()
        plan = self.plan
        if plan.world == 1 and self.bufs is None:
            full = self.local
        elif plan.rank != self.dst:
            return None
        else:
            # receive buffers that are consecutive rows of one [world, count] tensor (see
            # `gather_buffers`) need no stacking copy: one permute kernel undoes the deal
            base = getattr(self.bufs[0], "_base", None)
            if (base is not None and base.dim() == 2 and base.shape[0] == plan.world
                    and all(getattr(b, "_base", None) is base for b in self.bufs)):
                stacked = base
            else:
                stacked = torch.stack(self.bufs)
            full = stacked.view(plan.world, plan.n_blocks, plan.block).permute(1, 0, 2).reshape(-1)
        return full.view(plan.width, plan.height).t()


def gather_buffers(plan: ShardPlan, like):
    """Receive buffers for `gather_image_async` on the destination rank: `world` rows of ONE
    [world, count] tensor, so the assembled image is a single permute away."""
    import torch

    base = torch.empty((plan.world, plan.count), dtype=like.dtype, device=like.device)
    return [base[i] for i in range(plan.world)]


def gather_image_async(local, plan: ShardPlan, group=None, dst: int = 0, recv_bufs=None) -> PendingGather:
    """Start the single end-of-render gather without blocking the stream that traces the next
    image: RCCL runs it on its own stream, so it overlaps with the next render's kernel.  The caller
    must not overwrite `local` (nor `recv_bufs`) before `result()` of this handle has been called."""
    import os

    import torch
    import torch.distributed as dist

    # GRADUS_FORCE_COLLECTIVE=1 sends a single rank through the collective as well (exercises the RCCL call, its
    # stream hand-over and the receive-buffer assembly on a one-GPU box)
    force = plan.world == 1 and os.environ.get("GRADUS_FORCE_COLLECTIVE") == "1" and dist.is_initialized()
    if plan.world == 1 and not force:
        return PendingGather(None, None, local, plan, dst)
    bufs = None
    if plan.rank == dst:
        bufs = recv_bufs if recv_bufs is not None else [torch.empty_like(local) for _ in range(plan.world)]
    work = dist.gather(local, bufs, dst=dst, group=group, async_op=True)
    return PendingGather(work, bufs, local, plan, dst)


# ------------------------------------------------------------------------------------------
# BinningMethod line profiles (BASELINE config 5) over several GPUs: the rays of one PolarPlane are dealt to the ranks,
# every rank bins its own, ONE all-reduce of the histogram (n_bins doubles) is the path's only exchange step.
# ------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class RayShard:
    """Block-cyclic deal of the rays 0 .. n-1 of a separable ray set (gr_rayset.sep_first / sep_block / sep_stride):
    rank r takes blocks r, r + world, r + 2 world, ... of `block` consecutive rays; the last block may be short."""

    n: int
    world: int
    rank: int
    block: int

    @property
    def first(self) -> int:
        return self.rank * self.block

    @property
    def stride(self) -> int:
        return self.world * self.block

    @property
    def count(self) -> int:
        nb = -(-self.n // self.block)                       # blocks in the set
        mine = len(range(self.rank, nb, self.world))
        if mine == 0:
            return 0
        last = self.rank + (mine - 1) * self.world
        return (mine - 1) * self.block + min(self.block, self.n - last * self.block)

    def global_index(self, j):
        b = j // self.block
        return self.first + b * self.stride + (j - b * self.block)


def ray_shard(plane, world: int, rank: int) -> RayShard:
    """One block = one strip of 8 x 8 tiles down the plane (all radii of 8 neighbouring angles): every rank gets every
    radius and an even sample of the angles, i.e. the same mix of short and long rays."""
    n = plane.Nr * plane.Nθ
    block = 8 * ((plane.Nr // 8) * 8) if plane.Nr >= 8 and plane.Nθ >= 8 else max(64, -(-n // (world * 8)))
    return RayShard(n, world, rank, block)


def lineprofile_sharded(bins, ε, m, u, d, plane, *, maxrₑ=50.0, minrₑ=None, λ_max=None, redshift_pf=None, callback="default",
                        ensemble=None, group=None, shard=None, **solver_args):
    """lineprofile(bins, ε, m, u, d, BinningMethod(); plane) (src/line-profiles.jl:152-198) with the plane's rays dealt
    over the ranks of `group` (one process per GPU).  Each rank traces and bins its rays into an un-normalised histogram
    in HBM (gr_lineprofile_device on a sub-range of the separable ray set), one all-reduce (RCCL) sums the histograms,
    every rank normalises.  Returns (bins, flux) on every rank.  Without an initialised process group: one rank."""
    import numpy as np
    import torch

    from .device import lineprofile_device

    world = rank = None
    try:
        import torch.distributed as dist

        if dist.is_available() and dist.is_initialized():
            world, rank = dist.get_world_size(group), dist.get_rank(group)
    except ImportError:          # pragma: no cover
        dist = None
    if world is None:
        world, rank = 1, 0
    if shard is None:                                        # (a given shard: one rank's share timed alone)
        shard = ray_shard(plane, world, rank)
    flux = lineprofile_device(bins, ε, m, u, d, plane, shard=shard, maxrₑ=maxrₑ, minrₑ=minrₑ, λ_max=λ_max, redshift_pf=redshift_pf,
                              callback=callback, ensemble=ensemble, **solver_args)
    forced = os.environ.get("GRADUS_FORCE_COLLECTIVE") == "1" and dist is not None and dist.is_initialized()
    if world > 1 or forced:                                  # forced: one rank through RCCL (tests on a 1-GPU box)
        dist.all_reduce(flux, op=dist.ReduceOp.SUM, group=group)
    total = flux.sum()
    out = torch.where(total != 0, flux / total, flux)
    return np.asarray(bins, dtype=np.float64), out.cpu().numpy()
