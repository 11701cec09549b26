#!/bin/bash
# Build libgradus_mi355x.so of the working tree (or of a git revision that already has the per-metric layout) into
# ab/<name>.so with extra hipcc flags, for interleaved A/B timing (scripts/ab_bench.py).
#   scripts/build_variant.sh <rev|WORK> <name> [extra hipcc flags...]
# METRICS="0 1" limits the kernel objects that are recompiled with the extra flags (the others are taken from the
# in-tree build): A/B of one metric's kernels in seconds.  Revisions from before the per-metric layout (round 1) are
# built by their own recipe: git worktree + that revision's __graft_entry__.build_hip().
set -e
REV=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${AB_DIR:-$ROOT/ab}; mkdir -p $OUT
python3 - "$ROOT" "$REV" "$OUT/$NAME.so" "${METRICS:-}" "$@" <<'PY'
import os, shutil, subprocess, sys, tempfile
root, rev, out, metrics = sys.argv[1:5]
extra = sys.argv[5:]
sys.path.insert(0, root)
tmp = tempfile.mkdtemp()
src = os.path.join(tmp, "a", "b", "csrc")          # the sources include "../../include/gradus_mi355x.h" relative to csrc
os.makedirs(src); os.makedirs(os.path.join(tmp, "a", "include"))
names = ["gr_device.hpp", "gr_kernels.hpp", "gr_tangent.hpp", "gr_mesh_grid.hpp", "gr_tabmetric.hpp", "gradus_mi355x.hip", "kernels_tu.hip", "metric_table.hip"]
if rev == "WORK":
    for f in names:
        shutil.copy(os.path.join(root, "gradus.jl_amd", "csrc", f), src)
    shutil.copy(os.path.join(root, "include", "gradus_mi355x.h"), os.path.join(tmp, "a", "include"))
else:
    for f in names:
        open(os.path.join(src, f), "wb").write(subprocess.check_output(["git", "-C", root, "show", f"{rev}:gradus.jl_amd/csrc/{f}"]))
    open(os.path.join(tmp, "a", "include", "gradus_mi355x.h"), "wb").write(
        subprocess.check_output(["git", "-C", root, "show", f"{rev}:include/gradus_mi355x.h"]))
import __graft_entry__ as g
units = g.hip_units(extra)
if metrics and rev == "WORK":
    # "t0" = only the TANGENT object of metric 0 (kernelstan_m0.o); "0" = that metric's fp64, fp32 and tangent objects
    keep = {"gradus_mi355x.o"}
    for m in metrics.split():
        if m.startswith("t"):
            keep.add(f"kernelstan_m{m[1:]}.o")
        elif m == "11":      # GR_METRIC_TABULATED: its kernels and the host fit share gr_tabmetric.hpp (-DGR_TAB_DEGREE=...)
            keep |= {"kernels_m11.o", "kernels32_m11.o", "kernelstan_m11.o", "kernelstan1_m11.o", "metric_table.o"}
        else:
            keep |= {f"kernels_m{m}.o", f"kernels32_m{m}.o", f"kernelstan_m{m}.o"}
    for o, _, _ in units:
        if o not in keep:
            shutil.copy(os.path.join(root, "gradus.jl_amd", "csrc", o), os.path.join(src, o))
    units_c = [u for u in units if u[0] in keep]
else:
    units_c = units
g.compile_units(units_c, src, src, force=True)
subprocess.check_call([g._hipcc(), "--offload-arch=gfx950", "--offload-compress", "-shared", "-fPIC", "-o", out] + [os.path.join(src, o) for o, _, _ in units])
shutil.rmtree(tmp)
print("built", out)
PY
