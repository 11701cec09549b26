#!/usr/bin/env python3
"""BASELINE config 5 (Kerr a = 0.998, θ = 60°, ThinDisc(isco, 250), 4096² polar-plane rays, 180 bins) sharded over the GPUs
of a node: one process per GPU, each bins its block-cyclic share of the plane's rays, ONE RCCL all-reduce of the 180-bin
histogram per profile (gradus.jl_amd/distributed.py::lineprofile_sharded).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \\
        scripts/lineprofile_sharded.py [--size 4096] [--steps 10] [--warmup 2] [--tol 1e-9]

Prints one JSON line on rank 0: rays/s over all ranks (max over ranks of the timed region, barrier + synchronize on both
sides) and the profile's distance to the single-GPU one."""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--size", type=int, default=4096)
ap.add_argument("--steps", type=int, default=10)
ap.add_argument("--warmup", type=int, default=2)
ap.add_argument("--tol", type=float, default=1e-9)
ap.add_argument("--emulate-shard", type=str, default=None, help="W:R -- time rank R's share of a W-rank deal on this one GPU")
args = ap.parse_args()
rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
torch.cuda.set_device(local)
dev = torch.device("cuda", local)
if "RANK" in os.environ:
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

import gradus_jl_amd as G
from gradus_jl_amd.distributed import lineprofile_sharded

ens = G.EnsembleMI355X(local)
m = G.KerrMetric(1.0, 0.998)
u = np.array([0.0, 1000.0, math.radians(60), 0.0])
d = G.ThinDisc(m.isco(), 250.0)
plane = G.PolarPlane(G.GeometricGrid(), Nr=args.size, Nθ=args.size, r_min=1.0, r_max=250.0)
bins = np.linspace(0.1, 1.5, 180)
kw = dict(maxrₑ=250.0, ensemble=ens, abstol=args.tol, reltol=args.tol)
if args.emulate_shard:
    from gradus_jl_amd.distributed import ray_shard

    W, R = (int(v) for v in args.emulate_shard.split(":"))
    kw["shard"] = ray_shard(plane, W, R)


def sync():
    if dist.is_initialized():
        dist.barrier()
    torch.cuda.synchronize()


for _ in range(args.warmup):
    _, y = lineprofile_sharded(bins, G.PowerLawEmissivity(3), m, u, d, plane, **kw)
sync()
t0 = time.perf_counter()
for _ in range(args.steps):
    _, y = lineprofile_sharded(bins, G.PowerLawEmissivity(3), m, u, d, plane, **kw)
sync()
t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
if dist.is_initialized():
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
if rank == 0:
    n = args.size * args.size
    line = {"workload": f"C5 line profile, {args.size}^2 polar-plane rays, tol {args.tol:g}", "n_gpus": world, "steps": args.steps,
            "ms_per_profile": float(t.item()) / args.steps * 1e3, "rays_per_s": n * args.steps / float(t.item()),
            "rccl_ranks": dist.get_world_size() if dist.is_initialized() else 1, "profile_sum": float(y.sum()),
            "profile_peak_bin": int(np.argmax(y)), "profile_peak": float(y.max())}
    if args.emulate_shard:
        line["emulated_shard"] = args.emulate_shard
        line["rays_of_this_shard"] = kw["shard"].count
    print(json.dumps(line))
if dist.is_initialized():
    dist.barrier()
    dist.destroy_process_group()
