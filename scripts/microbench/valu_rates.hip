// valu_rates.hip -- issue rate of the FP vector instructions this library's kernels are made of, on gfx950:
// v_fma_f64, v_mul_f64, v_fma_f32, v_pk_fma_f32, v_mov_b32, v_rcp_f64 -- as independent streams (8 accumulators) and
// as one dependent chain, at 1 / 2 / 3 / 4 waves per SIMD.  Answers two design questions of DESIGN.md §5/§10:
//   * how much a third wave per SIMD can hide of a dependent FP64 chain (the Kerr kernel went from 2 to 3 waves);
//   * whether fp32 needs PACKED instructions to beat fp64.
// Build & run:  hipcc --offload-arch=gfx950 -O3 -o /tmp/valu_rates scripts/microbench/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x

template <int KIND, bool DEP>
__global__ void __launch_bounds__(64) k(double* out, int iters, long long* cycles)
{
    double a0 = threadIdx.x * 1e-3 + 1.0, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 1.0000001, c = 1e-9;
    float f0 = (float)a0, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7;
    const float mf = 1.0000001f, cf = 1e-9f;
    typedef float v2 __attribute__((ext_vector_type(2)));
    v2 p0 = { f0, f1 }, p1 = { f2, f3 }, p2 = { f4, f5 }, p3 = { f6, f7 }, p4 = p0 + 1.f, p5 = p1 + 1.f, p6 = p2 + 1.f, p7 = p3 + 1.f;
    const v2 mp = { mf, mf }, cp = { cf, cf };
    const long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {           // v_fma_f64
            if (DEP) { REP16(asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a0) : "v"(m), "v"(c));) }
            else { asm volatile("v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                "v_fma_f64 %0, %0, %8, %9\n v_fma_f64 %1, %1, %8, %9\n v_fma_f64 %2, %2, %8, %9\n v_fma_f64 %3, %3, %8, %9\n"
                                "v_fma_f64 %4, %4, %8, %9\n v_fma_f64 %5, %5, %8, %9\n v_fma_f64 %6, %6, %8, %9\n v_fma_f64 %7, %7, %8, %9\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)); }
        } else if (KIND == 1) {    // v_mul_f64
            if (DEP) { REP16(asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a0) : "v"(m));) }
            else { asm volatile("v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                                "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                                "v_mul_f64 %0, %0, %8\n v_mul_f64 %1, %1, %8\n v_mul_f64 %2, %2, %8\n v_mul_f64 %3, %3, %8\n"
                                "v_mul_f64 %4, %4, %8\n v_mul_f64 %5, %5, %8\n v_mul_f64 %6, %6, %8\n v_mul_f64 %7, %7, %8\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m)); }
        } else if (KIND == 2) {    // v_fma_f32
            if (DEP) { REP16(asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(mf), "v"(cf));) }
            else { asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                                "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7) : "v"(mf), "v"(cf)); }
        } else if (KIND == 3) {    // v_pk_fma_f32
            if (DEP) { REP16(asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p0) : "v"(mp), "v"(cp));) }
            else { asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                                "v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                                "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                                : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(mp), "v"(cp)); }
        } else if (KIND == 4) {    // v_mov_b32
            if (DEP) { REP16(asm volatile("v_mov_b32 %0, %0" : "+v"(f0));) }
            else { asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                                "v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                                : "+v"(f0), "+v"(f1), "+v"(f2), "+v"(f3), "+v"(f4), "+v"(f5), "+v"(f6), "+v"(f7)); }
        } else {                   // v_rcp_f64
            if (DEP) { REP16(asm volatile("v_rcp_f64 %0, %0" : "+v"(a0));) }
            else { asm volatile("v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                                "v_rcp_f64 %0, %0\n v_rcp_f64 %1, %1\n v_rcp_f64 %2, %2\n v_rcp_f64 %3, %3\n v_rcp_f64 %4, %4\n v_rcp_f64 %5, %5\n v_rcp_f64 %6, %6\n v_rcp_f64 %7, %7\n"
                                : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7)); }
        }
    }
    const long long t1 = clock64();
    out[blockIdx.x * 64 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

template <int KIND, bool DEP>
void run(const char* name, double* d_out, long long* d_cyc, int n_cu)
{
    const int iters = 4000;
    for (int wps : { 1, 2, 3, 4 }) {
        const int blocks = n_cu * 4 * wps;      // one-wave workgroups: wps waves per SIMD when the dispatcher spreads them evenly
        hipEvent_t e0, e1;
        hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, d_out, 10, d_cyc);
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<KIND, DEP>), dim3(blocks), dim3(64), 0, 0, d_out, iters, d_cyc);
        hipEventRecord(e1);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> cyc(blocks);
        hipMemcpy(cyc.data(), d_cyc, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
        double mean = 0;
        for (long long c : cyc) mean += (double)c;
        mean /= blocks;
        const double per_wave = mean / (16.0 * iters);                  // cycles between two instructions of ONE wave (s_memtime ticks)
        const double per_simd = per_wave / wps;                         // SIMD cycles per wave-instruction
        const double inst_per_s = (double)blocks * 16.0 * iters / (ms * 1e-3);
        printf("{\"inst\": \"%s\", \"dependent\": %s, \"waves_per_simd\": %d, \"ticks_per_inst_one_wave\": %.2f, \"ticks_per_inst_per_simd\": %.2f, "
               "\"wave_inst_per_s\": %.4e, \"ms\": %.3f}\n", name, DEP ? "true" : "false", wps, per_wave, per_simd, inst_per_s, ms);
    }
}

int main()
{
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const int n_cu = prop.multiProcessorCount;
    double* d_out; long long* d_cyc;
    hipMalloc(&d_out, sizeof(double) * 64 * n_cu * 16);
    hipMalloc(&d_cyc, sizeof(long long) * n_cu * 16);
    printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d}\n", prop.gcnArchName, n_cu, prop.clockRate / 1000);
    run<0, false>("v_fma_f64", d_out, d_cyc, n_cu); run<0, true>("v_fma_f64", d_out, d_cyc, n_cu);
    run<1, false>("v_mul_f64", d_out, d_cyc, n_cu); run<1, true>("v_mul_f64", d_out, d_cyc, n_cu);
    run<2, false>("v_fma_f32", d_out, d_cyc, n_cu); run<2, true>("v_fma_f32", d_out, d_cyc, n_cu);
    run<3, false>("v_pk_fma_f32", d_out, d_cyc, n_cu); run<3, true>("v_pk_fma_f32", d_out, d_cyc, n_cu);
    run<4, false>("v_mov_b32", d_out, d_cyc, n_cu); run<4, true>("v_mov_b32", d_out, d_cyc, n_cu);
    run<5, false>("v_rcp_f64", d_out, d_cyc, n_cu); run<5, true>("v_rcp_f64", d_out, d_cyc, n_cu);
    return 0;
}
