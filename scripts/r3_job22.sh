#!/bin/bash
# round 3, GPU job 22: result blocks on transparent huge pages + hipHostRegister: cost of a block, what the caller sees, GPU suite
R=${GRAFT_REPO_ROOT:-$(pwd)}; cd $R
O=$R/gpurun_out/r3v; mkdir -p $O
python3 -c "
import gradus_jl_amd as G
e = G.EnsembleMI355X(0); e.set('pinned_pool_mib', 0)
" 
GRADUS_POOL0=1 timeout 300 python3 - > $O/host_alloc_time.txt 2>&1 <<'PY'
import ctypes as C, time, sys, os
sys.path.insert(0, os.getcwd())
import numpy as np, gradus_jl_amd as G
from gradus_jl_amd import _lib
ens = G.EnsembleMI355X(0); L = _lib.load()
for pool in (0, 4096):
    ens.set("pinned_pool_mib", pool)
    for mib in (608, 608, 608, 32, 4):
        p = C.c_void_p(); t0 = time.perf_counter()
        _lib.check(L.gr_host_alloc(ens.ctx.handle, mib << 20, C.byref(p))); t1 = time.perf_counter()
        _lib.check(L.gr_host_free(ens.ctx.handle, p)); t2 = time.perf_counter()
        print(f"pool {pool:4d} MiB: gr_host_alloc({mib} MiB) {1e3*(t1-t0):8.2f} ms   gr_host_free {1e3*(t2-t1):8.2f} ms")
PY
cat $O/host_alloc_time.txt
timeout 600 python3 scripts/endpoints_call_time.py 2048 > $O/endpoints_call_time.txt 2>&1; cat $O/endpoints_call_time.txt
timeout 600 python3 scripts/endpoints_pinned_time.py 2048 > $O/endpoints_pinned.log 2>&1; head -6 $O/endpoints_pinned.log
timeout 1500 python3 -m pytest tests -m gpu -q -rxXs > $O/pytest.log 2>&1; tail -3 $O/pytest.log
