"""Planes sized for 288 GB of HBM: a fused N x N redshift render kept on the device (8 B/ray), with
the 64-bit index maps checked by re-rendering windows of the same plane through `gr_range` -- including
windows beyond ray 2^31 when N > 46340 -- and comparing bit for bit.   python scripts/big_render.py [N]"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gradus_jl_amd as G
from gradus_jl_amd import _lib, device as gdev

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
dev = torch.device("cuda", 0)
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
pf = G.ConstPointFunctions.redshift(m, x) @ G.ConstPointFunctions.filter_intersected()
cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=N, image_height=N,
                             alpha_lims=(-60, 60), beta_lims=(-35, 35), ensemble=ens)
n = N * N
out = torch.empty(n, dtype=torch.float64, device=dev)
stats = gdev.new_stats(dev)
torch.cuda.synchronize()
t0 = time.perf_counter()
gdev.render_device(cfg, pf, out, None, stats)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
st = gdev.stats_dict(stats)
print(f"N={N}: {n} rays ({n * 8 / 2**30:.1f} GiB image) in {dt:.2f} s = {n / dt:.3e} rays/s, steps/ray "
      f"{st['accepted_steps'] / st['rays']:.1f}, flagged {st['flagged_rays']}, status {st['status_count']}", flush=True)
assert st["rays"] == n and st["flagged_rays"] == 0
win = 1 << 16
firsts = [0, n // 3 // win * win, (n - win) // win * win]
if n > 2**31:
    firsts += [2**31 - win // 2, 2**31 + 5 * win]
sub = torch.empty(win, dtype=torch.float64, device=dev)
for f in firsts:
    gdev.render_device(cfg, pf, sub, _lib.gr_range(int(f), win, win, 1), None)
    torch.cuda.synchronize()
    a, b = out[f:f + win], sub
    same = bool(torch.equal(torch.nan_to_num(a, nan=-1.0), torch.nan_to_num(b, nan=-1.0)))
    print(f"  window at ray {f}: {'identical' if same else 'DIFFERENT'}; hits {int(torch.isfinite(b).sum())}", flush=True)
    assert same
# the far columns hold no disc: check the pixel -> ray map there with a point function that varies from
# pixel to pixel on every ray (projected end radius), against the same pixels of a SMALL plane whose
# window coincides: columns [c0, c0 + 4) of the big plane == a 4-column plane with the matching α limits
from gradus_jl_amd.pointfunctions import GR_PF_RADIUS, PointFunction
pr = PointFunction(lambda *a, **k: None, device_pf=GR_PF_RADIUS)
al = np.linspace(-60.0, 60.0, N)
# (not the central columns: rays that wind around the photon ring amplify the 1e-16 difference between the two ways
# of forming α by ten orders of magnitude)
for c0 in ([N // 2 + N // 16, N - 8] + ([int(2**31 // N) + 1] if n > 2**31 else [])):
    f = c0 * N
    big = torch.empty(4 * N, dtype=torch.float64, device=dev)
    gdev.render_device(cfg, pr, big, _lib.gr_range(int(f), 4 * N, 4 * N, 1), None)
    small_cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=4, image_height=N,
                                       alpha_lims=(float(al[c0]), float(al[c0 + 3])), beta_lims=(-35, 35), ensemble=ens)
    small = torch.empty(4 * N, dtype=torch.float64, device=dev)
    gdev.render_device(small_cfg, pr, small, None, None)
    torch.cuda.synchronize()
    rel = float(((big - small).abs() / small.abs().clamp(min=1.0)).max())
    print(f"  columns {c0}..{c0 + 3} (first ray {f}): max rel diff vs a 4-column plane of the same pixels {rel:.2e}, "
          f"radius range [{float(small.min()):.3f}, {float(small.max()):.3f}]", flush=True)
    assert rel < 1e-6 and float(small.max()) > float(small.min())
hits = int(torch.isfinite(out).sum())
print(f"hits {hits} ({hits / n:.4f} of the plane), g in [{float(torch.nan_to_num(out, nan=9.0).min()):.4f}, "
      f"{float(torch.nan_to_num(out, nan=-9.0).max()):.4f}]")
