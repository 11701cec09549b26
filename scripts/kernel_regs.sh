#!/bin/bash
# Register / spill report of the trace kernels (device-only compile of the fp64 unit, ~1 min).
# usage: scripts/kernel_regs.sh [grep pattern, default k_trace_lane]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TMP=$(mktemp -d)
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm --cuda-device-only \
      -c "$ROOT/gradus.jl_amd/csrc/gradus_mi355x.hip" -o $TMP/dev.co
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=$TMP/dev.co \
      --targets=hip-amdgcn-amd-amdhsa--gfx950 --output=$TMP/dev.elf
/opt/rocm/lib/llvm/bin/llvm-readelf --notes $TMP/dev.elf \
  | grep -E "\.name:|\.vgpr_count|\.sgpr_count|spill_count|private_segment_fixed" | paste - - - - - - \
  | sed 's/ \+/ /g' | grep -E "${1:-k_trace_lane}"
rm -rf $TMP
