"""Device time and achieved HBM bandwidth of k_apply_pf (apply(pf, cache): 152 B read + 8 B written per ray) on the
end points of the bench plane.   python scripts/apply_pf_time.py [size]"""
import ctypes as C, math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gradus_jl_amd as G
from gradus_jl_amd import _lib, device as gdev
from gradus_jl_amd.rendering import abi_pointfunction

N = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
ens = G.EnsembleMI355X(0)
m = G.KerrMetric(1.0, 0.998)
x = np.array([0.0, 1000.0, math.radians(75), 0.0])
cfg = G.render_configuration(m, x, G.ThinDisc(m.isco(), 50.0), 2000.0, image_width=N, image_height=N, alpha_lims=(-60, 60),
                             beta_lims=(-35, 35), ensemble=ens)
dev = torch.device("cuda", 0)
raw = torch.empty(N * N * 152, dtype=torch.uint8, device=dev)
gdev.render_endpoints_device(cfg, raw)
out = torch.empty(N * N, dtype=torch.float64, device=dev)
L = _lib.load()
c = cfg.abi_config()
CPF = G.ConstPointFunctions
for name, pf in (("redshift∘filter_intersected", CPF.redshift(m, x) @ CPF.filter_intersected()), ("affine_time", CPF.affine_time())):
    s, keep = abi_pointfunction(pf)
    ts = []
    for i in range(10):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        _lib.check(L.gr_apply_pointfunction_device(ens.ctx.handle, C.byref(c), C.byref(s), raw.data_ptr(), N * N, 2000.0, out.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream))
        b.record(); torch.cuda.synchronize()
        if i >= 3:
            ts.append(a.elapsed_time(b))
    t = float(np.median(ts))
    print(f"{name:28s} {t:.3f} ms  = {N * N * 160 / t / 1e6:.0f} GB/s of algorithmic traffic ({N * N * 160 / 1e6:.0f} MB)")
