// Page-locking a 608 MiB result block: hipHostMalloc against mmap + MADV_HUGEPAGE + pre-fault from 8 threads + hipHostRegister
// (can the first gr_host_alloc of a session be made cheaper than 113-365 ms?)   hipcc -O2 -o /tmp/hr_thp host_register_thp.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void prefault(char* b, size_t n, unsigned nt, size_t step) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([=]() { for (size_t o = (size_t)t * step; o < n; o += (size_t)nt * step) ((volatile char*)b)[o] = 0; });
    for (auto& x : th) x.join();
}
__global__ void fill(double* p, size_t n) { size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; if (i < n) p[i] = (double)i; }
int main() {
    const size_t n = (size_t)2048 * 2048 * 152;
    void* d; hipMalloc(&d, n); hipMemset(d, 1, n); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        void* p = nullptr;
        double a0 = now(); hipError_t e0 = hipHostMalloc(&p, n, hipHostMallocPortable); double a1 = now();
        hipMemcpy(p, d, n, hipMemcpyDeviceToHost); double a2 = now();
        hipHostFree(p); double a3 = now();
        printf("hipHostMalloc %.1f ms (%d) | D2H %.1f ms | hipHostFree %.1f ms\n", a1 - a0, (int)e0, a2 - a1, a3 - a2);
        const size_t two = (size_t)2 << 20, len = (n + two - 1) / two * two;
        double b0 = now();
        char* m = (char*)mmap(nullptr, len + two, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0);
        char* al = (char*)(((size_t)m + two - 1) / two * two);
        int rc = madvise(al, len, MADV_HUGEPAGE);
        double b1 = now(); prefault(al, len, 8, 4096); double b2 = now();
        hipError_t e1 = hipHostRegister(al, len, hipHostRegisterPortable | hipHostRegisterMapped); double b3 = now();
        hipMemcpy(al, d, n, hipMemcpyDeviceToHost); double b4 = now();
        void* dp = nullptr; hipError_t e2 = hipHostGetDevicePointer(&dp, al, 0);
        double b5 = now();
        if (e2 == hipSuccess && dp) { fill<<<(unsigned)((n / 8 + 255) / 256), 256>>>((double*)dp, n / 8); hipDeviceSynchronize(); }
        double b6 = now();
        const double chk = ((double*)al)[12345];
        hipHostUnregister(al); double b7 = now();
        munmap(m, len + two); double b8 = now();
        printf("   mmap+madvise %.2f ms (rc %d) | prefault(8 thr) %.1f ms | hipHostRegister %.1f ms (%d) | D2H %.1f ms | devptr %d, kernel stores across the link %.1f ms (%.1f GB/s), check %.0f | unregister %.1f ms | munmap %.1f ms\n",
               b1 - b0, rc, b2 - b1, b3 - b2, (int)e1, b4 - b3, (int)e2, b6 - b5, n / (b6 - b5) / 1e6, chk, b7 - b6, b8 - b7);
    }
    return 0;
}
