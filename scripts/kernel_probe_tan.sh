#!/bin/bash
# The TANGENT flavour of one trace kernel to gfx950 assembly (see kernel_probe.sh).
#   scripts/kernel_probe_tan.sh [Metric=KerrMetric] [DISC=GR_DISC_DATUM] [W=2] [min waves per SIMD=1] [extra hipcc flags...]
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
M=${1:-KerrMetric}; D=${2:-GR_DISC_DATUM}; W=${3:-2}; MW=${4:-1}; shift 4 2>/dev/null || true
mkdir -p /tmp/probe
cat > /tmp/probe/probe_tan.hip <<EOT
#include <hip/hip_runtime.h>
#define GR_REAL_IS_TAN2 1
#define GR_TAN_W $W
#define GR_NS grt
#define GR_LANE_ONLY 1
#define GR_LANE_MIN_WAVES $MW
#ifndef GR_PARK_STAGES
#define GR_PARK_STAGES 0
#endif
#define GR_NO_LAUNCHER 1
#include "$ROOT/gradus.jl_amd/csrc/gr_kernels.hpp"
using namespace grt;
template __global__ void grt::k_trace_lane<$M, $D>(const Params);
EOT
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -mllvm -disable-machine-licm -ffp-contract=on -S --cuda-device-only "$@" \
      -o /tmp/probe/probe_tan.s /tmp/probe/probe_tan.hip 2>&1 | grep -v "hip-link" || true
grep -E "amdhsa_next_free_vgpr|amdhsa_next_free_sgpr|amdhsa_private_segment_fixed_size" /tmp/probe/probe_tan.s | tr -s '\t ' ' ' | paste -sd' '
python3 "$ROOT/scripts/asm_blocks.py" /tmp/probe/probe_tan.s
