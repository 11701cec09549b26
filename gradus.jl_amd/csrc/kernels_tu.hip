// kernels_tu.hip -- the device code of ONE metric: compiled once per metric id and per precision,
//
//     hipcc ... -DGR_TU_METRIC=<GR_METRIC_* id> -c kernels_tu.hip -o kernels_m<id>.o                       (fp64)
//     hipcc ... -DGR_TU_METRIC=<id> -DGR_TU_F32 -Xclang -cl-single-precision-constant ... -o kernels32_m<id>.o   (fp32)
//     hipcc ... -DGR_TU_METRIC=<id> -DGR_TU_TAN -c kernels_tu.hip -o kernelstan_m<id>.o                     (tangent, one lane per ray)
//     hipcc ... -DGR_TU_METRIC=<id> -DGR_TU_TAN1 -c kernels_tu.hip -o kernelstan1_m<id>.o                   (tangent, a pair of lanes per ray)
//
// so that every metric's kernels hold only that metric's component function (gr_device.hpp, GenericMetricT) and the
// library builds on all cores at once (__graft_entry__.build_hip).  The fp32 objects are the same source with
// real = float (every floating literal of gr_device.hpp / gr_kernels.hpp is single precision there); they carry the
// trace kernels only: gr_ctx_set(ctx, "precision", 32) selects them for tolerance sweeps.  Inputs, outputs and tables
// stay double in both.
//
// Exports (plain signatures: the host unit, gradus_mi355x.hip, passes its gr::Params by address -- gr32::Params has the
// same layout, its fields are double / integer / pointers only):
//     gr64_launch_trace_m<ID>, gr64_launch_path_m<ID>, gr64_launch_apply_m<ID>      gr32_launch_trace_m<ID>      grt_launch_trace_m<ID>      grt1_launch_trace_m<ID>
#ifndef GR_TU_METRIC
#error "compile with -DGR_TU_METRIC=<metric id>"
#endif
#if defined(GR_TU_TAN) || defined(GR_TU_TAN1)
// third flavour: real = value + tangents (gr_tangent.hpp), the one-ray-per-lane kernel only, one wave per SIMD, in TWO shapes
// (round 4; measured with scripts/wave_timeline.py, scripts/lineprofile_tf_time.py; profiles/r4_tangent_*):
//   GR_TU_TAN   namespace grt,  GR_TAN_W = 2: one lane carries a ray with both directions of the Jacobian (455 registers, no
//               scratch).  The THROUGHPUT shape: 1024² rays in 20.9 ms.
//   GR_TU_TAN1  namespace grt1, GR_TAN_W = 1: a PAIR of lanes per ray, one direction each (299 registers, no scratch).  A
//               wave's step is a long dependent chain and a lane with one direction has 1750 instead of 2900 vector
//               instructions in it: 8.8 instead of 10.3 µs per step.  The LATENCY shape: the default line profile of the
//               reference (TransferFunctionMethod: 909 launches of ~126 rays, each as long as its longest ray) takes 1.16 s
//               with it against 1.55 s -- and 24.5 ms against 20.9 for the 1024² launch, where the value part computed
//               twice costs more than the shorter chain saves.
// The host unit picks per launch (gr_ctx_set "tangent_pairs": launches that cannot fill the SIMDs anyway take the pairs).
// A 256-register variant of the pair shape (two waves per SIMD, stage accelerations parked in LDS: -DGR_TAN_MIN_WAVES=2
// -DGR_PARK_STAGES=4) was measured and lost to both (24.5 ms; 1.31 s).
#define GR_REAL_IS_TAN2 1
#if defined(GR_TU_TAN1)
#define GR_TAN_W 1
#define GR_NS grt1
#define GR_TU_PREFIX grt1
#else
#define GR_TAN_W 2
#define GR_NS grt
#define GR_TU_PREFIX grt
#endif
#define GR_LANE_ONLY 1
#ifndef GR_TAN_MIN_WAVES
#define GR_TAN_MIN_WAVES 1  // waves per SIMD the tangent kernels are compiled for
#endif
#define GR_LANE_MIN_WAVES GR_TAN_MIN_WAVES
#ifndef GR_PARK_STAGES
#define GR_PARK_STAGES 0    // stage accelerations parked in LDS while a right-hand side runs (ParkA, gr_device.hpp)
#endif
#elif defined(GR_TU_F32)
#define GR_REAL_IS_FLOAT 1
#define GR_NS gr32
#define GR_TU_PREFIX gr32
#else
#define GR_NS gr
#define GR_TU_PREFIX gr64
#endif
#include <hip/hip_runtime.h>

#include "gr_kernels.hpp"

#define GR_CAT3_(a, b, c) a##b##c
#define GR_CAT3(a, b, c) GR_CAT3_(a, b, c)
#define GR_TU_NAME(what) GR_CAT3(GR_TU_PREFIX, what, GR_TU_METRIC)

namespace {
typedef GR_NS::MetricOf<GR_TU_METRIC>::type TuMetric;
}

hipError_t GR_TU_NAME(_launch_trace_m)(int kernel, int block, int n_cu, int waves_per_simd, unsigned long long* queue,
                                       const void* params, hipStream_t stream)
{
    GR_NS::Params p = *reinterpret_cast<const GR_NS::Params*>(params);
    GR_NS::LaunchKnobs k{ kernel, block, n_cu, waves_per_simd, queue };
    return GR_NS::launch_metric<TuMetric>(k, p, stream);
}

#if !defined(GR_TU_F32) && !defined(GR_TU_TAN) && !defined(GR_TU_TAN1)
hipError_t GR_TU_NAME(_launch_path_m)(const void* params, double* d_path, int64_t cap, unsigned long long* d_n, hipStream_t stream)
{
    return GR_NS::launch_path_metric<TuMetric>(*reinterpret_cast<const GR_NS::Params*>(params), d_path, cap, d_n, stream);
}

hipError_t GR_TU_NAME(_launch_apply_m)(const void* params, const gr_point* pts, double max_time, double* out, hipStream_t stream)
{
    return GR_NS::launch_apply_metric<TuMetric>(*reinterpret_cast<const GR_NS::Params*>(params), pts, max_time, out, stream);
}
#endif
