// valu_calib.hip -- calibration streams for the utilisation numbers of profiles/*_summary.json.
//
// One NAMED kernel per instruction class, each a pure stream of that instruction on 8 independent registers, three
// waves per SIMD on every SIMD of the chip, long enough (>= 20 ms) for the clocks to settle.  Run twice:
//   (1) bare:               ./valu_calib            -> wave-instructions/s per class from HIP events
//   (2) under the counters: rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES -- ./valu_calib
// (2) gives, per class: the sustained shader clock (GRBM_GUI_ACTIVE / 8 XCDs / duration), SIMD cycles per instruction
// (clock x duration x 1024 SIMDs / SQ_INSTS_VALU) and what `valu_busy = 4 SQ_ACTIVE_INST_VALU / SIMD-cycles` reads on a
// stream that keeps the VALU 100 % busy by construction (scripts/summarize_calib.py).
// The `mix` kernel interleaves the classes in the proportions of the bench kernel's step (83 % FP64, of which half FMA;
// 17 % 32-bit moves / selects).
//
// Build:  hipcc --offload-arch=gfx950 -O3 -o scripts/microbench/valu_calib scripts/microbench/valu_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define REP8(i0, i1, i2, i3, i4, i5, i6, i7) i0 "\n" i1 "\n" i2 "\n" i3 "\n" i4 "\n" i5 "\n" i6 "\n" i7 "\n"

#define STREAM_KERNEL(NAME, BODY, CONSTRAINTS_OUT, CONSTRAINTS_IN)                                          \
    __global__ void __launch_bounds__(64) NAME(double* out, int iters)                                     \
    {                                                                                                       \
        double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, \
               a7 = a0 + 7;                                                                                 \
        const double m = 1.0000001, c = 1e-9;                                                               \
        float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7; \
        const float mf = 1.0000001f, cf = 1e-9f;                                                            \
        for (int i = 0; i < iters; ++i) {                                                                   \
            asm volatile(BODY BODY BODY BODY : CONSTRAINTS_OUT : CONSTRAINTS_IN);                           \
        }                                                                                                   \
        out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7; \
    }

#define D_OUT "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)
#define F_OUT "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)
#define D_IN "v"(m), "v"(c)
#define F_IN "v"(mf), "v"(cf)

// 32 instructions per loop iteration in every kernel (4 x 8)
STREAM_KERNEL(calib_fma_f64,
              REP8("v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %1, %1, %8, %9", "v_fma_f64 %2, %2, %8, %9", "v_fma_f64 %3, %3, %8, %9",
                   "v_fma_f64 %4, %4, %8, %9", "v_fma_f64 %5, %5, %8, %9", "v_fma_f64 %6, %6, %8, %9", "v_fma_f64 %7, %7, %8, %9"),
              D_OUT, D_IN)
STREAM_KERNEL(calib_mul_f64,
              REP8("v_mul_f64 %0, %0, %8", "v_mul_f64 %1, %1, %8", "v_mul_f64 %2, %2, %8", "v_mul_f64 %3, %3, %8",
                   "v_mul_f64 %4, %4, %8", "v_mul_f64 %5, %5, %8", "v_mul_f64 %6, %6, %8", "v_mul_f64 %7, %7, %8"),
              D_OUT, D_IN)
STREAM_KERNEL(calib_add_f64,
              REP8("v_add_f64 %0, %0, %9", "v_add_f64 %1, %1, %9", "v_add_f64 %2, %2, %9", "v_add_f64 %3, %3, %9",
                   "v_add_f64 %4, %4, %9", "v_add_f64 %5, %5, %9", "v_add_f64 %6, %6, %9", "v_add_f64 %7, %7, %9"),
              D_OUT, D_IN)
STREAM_KERNEL(calib_fma_f32,
              REP8("v_fma_f32 %0, %0, %8, %9", "v_fma_f32 %1, %1, %8, %9", "v_fma_f32 %2, %2, %8, %9", "v_fma_f32 %3, %3, %8, %9",
                   "v_fma_f32 %4, %4, %8, %9", "v_fma_f32 %5, %5, %8, %9", "v_fma_f32 %6, %6, %8, %9", "v_fma_f32 %7, %7, %8, %9"),
              F_OUT, F_IN)
// the packed form: two FP32 FMAs per lane and instruction on a register PAIR -- the only way to the 157.3 TFLOP/s FP32 vector peak
// (round 5: what "fraction of the FP32 peak" means for a kernel of scalar v_fma_f32, DESIGN.md §5 "fp32 kernels")
STREAM_KERNEL(calib_pk_fma_f32,
              REP8("v_pk_fma_f32 %0, %0, %8, %9", "v_pk_fma_f32 %1, %1, %8, %9", "v_pk_fma_f32 %2, %2, %8, %9", "v_pk_fma_f32 %3, %3, %8, %9",
                   "v_pk_fma_f32 %4, %4, %8, %9", "v_pk_fma_f32 %5, %5, %8, %9", "v_pk_fma_f32 %6, %6, %8, %9", "v_pk_fma_f32 %7, %7, %8, %9"),
              D_OUT, D_IN)
STREAM_KERNEL(calib_mov_b32,
              REP8("v_mov_b32 %0, %1", "v_mov_b32 %1, %2", "v_mov_b32 %2, %3", "v_mov_b32 %3, %4", "v_mov_b32 %4, %5",
                   "v_mov_b32 %5, %6", "v_mov_b32 %6, %7", "v_mov_b32 %7, %0"),
              F_OUT, F_IN)
STREAM_KERNEL(calib_mov_b64,
              REP8("v_mov_b64 %0, %1", "v_mov_b64 %1, %2", "v_mov_b64 %2, %3", "v_mov_b64 %3, %4", "v_mov_b64 %4, %5",
                   "v_mov_b64 %5, %6", "v_mov_b64 %6, %7", "v_mov_b64 %7, %0"),
              D_OUT, D_IN)
STREAM_KERNEL(calib_cndmask_b32,
              REP8("v_cndmask_b32 %0, %0, %1, vcc", "v_cndmask_b32 %1, %1, %2, vcc", "v_cndmask_b32 %2, %2, %3, vcc",
                   "v_cndmask_b32 %3, %3, %4, vcc", "v_cndmask_b32 %4, %4, %5, vcc", "v_cndmask_b32 %5, %5, %6, vcc",
                   "v_cndmask_b32 %6, %6, %7, vcc", "v_cndmask_b32 %7, %7, %0, vcc"),
              F_OUT, F_IN)
STREAM_KERNEL(calib_rcp_f64,
              REP8("v_rcp_f64 %0, %0", "v_rcp_f64 %1, %1", "v_rcp_f64 %2, %2", "v_rcp_f64 %3, %3", "v_rcp_f64 %4, %4",
                   "v_rcp_f64 %5, %5", "v_rcp_f64 %6, %6", "v_rcp_f64 %7, %7"),
              D_OUT, D_IN)
// FP64 FMA with one SGPR-pair constant operand (how the tableau constants reach the kernel's stage sums)
__global__ void __launch_bounds__(64) calib_fma_f64_sgpr(double* out, int iters)
{
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double c = 1e-9;
    double m = 1.0000001;
    asm volatile("" : "+s"(m));
    for (int i = 0; i < iters; ++i) {
#define B REP8("v_fma_f64 %0, %0, %8, %9", "v_fma_f64 %1, %1, %8, %9", "v_fma_f64 %2, %2, %8, %9", "v_fma_f64 %3, %3, %8, %9", \
               "v_fma_f64 %4, %4, %8, %9", "v_fma_f64 %5, %5, %8, %9", "v_fma_f64 %6, %6, %8, %9", "v_fma_f64 %7, %7, %8, %9")
        asm volatile(B B B B : D_OUT : "s"(m), "v"(c));
#undef B
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}
// the bench kernel's mix: per 32 instructions 13 FMA, 11 MUL, 2 ADD (FP64), 4 v_mov_b32, 1 v_mov_b64, 1 v_cndmask
__global__ void __launch_bounds__(64) calib_mix(double* out, int iters)
{
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3;
    for (int i = 0; i < iters; ++i) {
        asm volatile(
            "v_fma_f64 %0, %0, %12, %13\n v_mul_f64 %1, %1, %12\n v_fma_f64 %2, %2, %12, %13\n v_mul_f64 %3, %3, %12\n"
            "v_fma_f64 %4, %4, %12, %13\n v_mul_f64 %5, %5, %12\n v_fma_f64 %6, %6, %12, %13\n v_mul_f64 %7, %7, %12\n"
            "v_mov_b32 %8, %9\n v_fma_f64 %0, %0, %12, %13\n v_mul_f64 %1, %1, %12\n v_fma_f64 %2, %2, %12, %13\n"
            "v_mul_f64 %3, %3, %12\n v_add_f64 %4, %4, %13\n v_mov_b32 %9, %10\n v_fma_f64 %5, %5, %12, %13\n"
            "v_mul_f64 %6, %6, %12\n v_fma_f64 %7, %7, %12, %13\n v_mul_f64 %0, %0, %12\n v_mov_b64 %1, %2\n"
            "v_fma_f64 %2, %2, %12, %13\n v_mul_f64 %3, %3, %12\n v_fma_f64 %4, %4, %12, %13\n v_mov_b32 %10, %11\n"
            "v_mul_f64 %5, %5, %12\n v_fma_f64 %6, %6, %12, %13\n v_add_f64 %7, %7, %13\n v_cndmask_b32 %11, %11, %8, vcc\n"
            "v_fma_f64 %0, %0, %12, %13\n v_mul_f64 %1, %1, %12\n v_fma_f64 %2, %2, %12, %13\n v_mov_b32 %8, %11\n"
            : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3)
            : "v"(m), "v"(c));
    }
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3;
}

typedef void (*kern_t)(double*, int);

static void run(const char* name, kern_t k, double* d_out, int n_cu, int wps, double target_ms)
{
    const int blocks = n_cu * 4 * wps;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    int iters = 20000;
    float ms = 0;
    // size the launch from a probe so that every class runs for about target_ms
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, iters);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    iters = (int)(iters * target_ms / ms);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(k, dim3(blocks), dim3(64), 0, 0, d_out, iters);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        hipEventElapsedTime(&ms, e0, e1);
        const double n_inst = (double)blocks * 32.0 * iters;
        printf("{\"kernel\": \"%s\", \"waves_per_simd\": %d, \"rep\": %d, \"iters\": %d, \"wave_insts\": %.6e, \"ms\": %.3f, "
               "\"wave_inst_per_s\": %.4e}\n", name, wps, rep, iters, n_inst, ms, n_inst / (ms * 1e-3));
        fflush(stdout);
    }
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    const int wps = argc > 1 ? atoi(argv[1]) : 3;
    const double target_ms = argc > 2 ? atof(argv[2]) : 25.0;
    double* d_out;
    hipMalloc(&d_out, sizeof(double) * 64 * n_cu * 4 * 8);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"waves_per_simd\": %d}\n", prop.gcnArchName, n_cu,
           prop.clockRate / 1000, wps);
    run("calib_fma_f64", calib_fma_f64, d_out, n_cu, wps, target_ms);
    run("calib_mul_f64", calib_mul_f64, d_out, n_cu, wps, target_ms);
    run("calib_add_f64", calib_add_f64, d_out, n_cu, wps, target_ms);
    run("calib_fma_f64_sgpr", calib_fma_f64_sgpr, d_out, n_cu, wps, target_ms);
    run("calib_fma_f32", calib_fma_f32, d_out, n_cu, wps, target_ms);
    run("calib_pk_fma_f32", calib_pk_fma_f32, d_out, n_cu, wps, target_ms);
    run("calib_mov_b32", calib_mov_b32, d_out, n_cu, wps, target_ms);
    run("calib_mov_b64", calib_mov_b64, d_out, n_cu, wps, target_ms);
    run("calib_cndmask_b32", calib_cndmask_b32, d_out, n_cu, wps, target_ms);
    run("calib_rcp_f64", calib_rcp_f64, d_out, n_cu, wps, target_ms);
    run("calib_mix", calib_mix, d_out, n_cu, wps, target_ms);
    return 0;
}
