#!/bin/bash
# Register / scratch report of one metric's kernels (device-only compile of its fp64 unit).
# usage: scripts/kernel_regs.sh [metric id, default 0 = Kerr] [grep pattern, default k_trace_lane]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm --cuda-device-only -DGR_TU_METRIC=${1:-0} \
      -c "$ROOT/gradus.jl_amd/csrc/kernels_tu.hip" -o $TMP/dev.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/dev.co \
      --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.elf \
  | grep -E "\.name:|\.vgpr_count|\.sgpr_count|spill_count|private_segment_fixed" | paste - - - - - - \
  | sed 's/ \+/ /g' | grep -E "${2:-k_trace_lane}"
rm -rf $TMP
