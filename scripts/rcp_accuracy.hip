// Measures the relative error of v_rcp_f64 / v_rsq_f64 seeds on gfx950 (diagnostic for DESIGN.md).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r, double* q, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) { r[i] = __builtin_amdgcn_rcp(x[i]); q[i] = __builtin_amdgcn_rsq(x[i]); } }
int main() {
    const int n = 1 << 20; double *hx = new double[n], *hr = new double[n], *hq = new double[n];
    unsigned long long s = 88172645463325252ull;
    for (int i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; double u = (s >> 11) * (1.0 / 9007199254740992.0); hx[i] = std::exp((u - 0.5) * 60.0); }
    double *dx, *dr, *dq; hipMalloc(&dx, n * 8); hipMalloc(&dr, n * 8); hipMalloc(&dq, n * 8);
    hipMemcpy(dx, hx, n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, dr, dq, n);
    hipMemcpy(hr, dr, n * 8, hipMemcpyDeviceToHost); hipMemcpy(hq, dq, n * 8, hipMemcpyDeviceToHost);
    double er = 0, eq = 0;
    for (int i = 0; i < n; ++i) { er = std::fmax(er, std::fabs(hr[i] * hx[i] - 1.0)); eq = std::fmax(eq, std::fabs(hq[i] * std::sqrt(hx[i]) - 1.0)); }
    printf("v_rcp_f64 max rel err %.3e ; v_rsq_f64 max rel err %.3e\n", er, eq);
    return 0;
}
