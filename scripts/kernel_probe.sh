#!/bin/bash
# Compile ONE instantiation of a trace kernel to gfx950 assembly (seconds instead of the ~90 s of the whole library)
# and print its register use and the instruction mix of the step loop's always-executed path.
#   scripts/kernel_probe.sh [Metric=KerrMetric] [DISC=GR_DISC_THIN] [kernel=k_trace_lane] [extra hipcc flags...]
# Output: /tmp/probe/probe.s ; the per-block table comes from scripts/asm_blocks.py.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
M=${1:-KerrMetric}; D=${2:-GR_DISC_THIN}; K=${3:-k_trace_lane}; shift 3 2>/dev/null || true
mkdir -p /tmp/probe
cat > /tmp/probe/probe.hip <<EOT
#include <hip/hip_runtime.h>
#define GR_NS gr
#define GR_NO_LAUNCHER 1
#include "$ROOT/gradus.jl_amd/csrc/gr_kernels.hpp"
using namespace gr;
template __global__ void gr::$K<$M, $D>(const Params);
EOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm -ffp-contract=on -S --cuda-device-only "$@" \
      -o /tmp/probe/probe.s /tmp/probe/probe.hip 2>&1 | grep -v "hip-link" || true
grep -E "amdhsa_next_free_vgpr|amdhsa_next_free_sgpr|amdhsa_private_segment_fixed_size" /tmp/probe/probe.s | tr -s '\t ' ' ' | paste -sd' '
python3 "$ROOT/scripts/asm_blocks.py" /tmp/probe/probe.s
