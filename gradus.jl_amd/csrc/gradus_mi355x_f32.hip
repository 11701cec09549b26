// gradus_mi355x_f32.hip -- the same trace kernels instantiated with real = float.
//
// Compiled with `-Xclang -cl-single-precision-constant` so every floating literal in
// gr_device.hpp / gr_kernels.hpp is single precision here.  Used for the fp32-vs-fp64 tolerance
// sweep of the line-profile configuration (BASELINE.json configs[4]); selected at run time with
// gr_ctx_set(ctx, "precision", 32).  Inputs, outputs and tables stay double.
#define GR_REAL_IS_FLOAT 1
#define GR_NS gr32
#include <hip/hip_runtime.h>

#include "gr_kernels.hpp"

hipError_t gr32_launch(int kernel, int block, int n_cu, int waves_per_simd, unsigned long long* queue,
                       const void* params, hipStream_t stream)
{
    // gr32::Params has the layout of gr::Params (its fields are double / integer / pointers only)
    gr32::Params p = *reinterpret_cast<const gr32::Params*>(params);
    gr32::LaunchKnobs k{ kernel, block, n_cu, waves_per_simd, queue };
    return gr32::launch_by_config(k, p, stream);
}
