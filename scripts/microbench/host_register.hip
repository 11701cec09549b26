// How long does it take to make a fresh 637 MB host buffer a DMA target?  (pre-fault from 8 threads, hipHostRegister,
// device-to-host copy into pageable / registered memory, hipHostUnregister)   hipcc -O2 -o /tmp/host_register host_register.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <sys/mman.h>
#include <thread>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
static void prefault(char* b, size_t n, unsigned nt) {
    std::vector<std::thread> th;
    for (unsigned t = 0; t < nt; ++t) th.emplace_back([=]() { for (size_t o = (size_t)t * 4096; o < n; o += (size_t)nt * 4096) ((volatile char*)b)[o] = 0; });
    for (auto& x : th) x.join();
}
int main() {
    const size_t n = (size_t)2048 * 2048 * 152;
    void* d; hipMalloc(&d, n); hipMemset(d, 1, n); hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        char* h = (char*)malloc(n);
        double t0 = now(); prefault(h, n, 8); double t1 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost); double t2 = now();
        hipError_t e = hipHostRegister(h, n, hipHostRegisterDefault); double t3 = now();
        hipMemcpy(h, d, n, hipMemcpyDeviceToHost); double t4 = now();
        hipHostUnregister(h); double t5 = now();
        free(h); double t6 = now();
        {   // transparent huge pages for the destination: 2 MB faults instead of 4 KB ones
            char* h3 = (char*)malloc(n);
            const size_t two = (size_t)2 << 20;
            char* a = (char*)(((size_t)h3 + two - 1) / two * two);
            double v0 = now();
            int rc = madvise(a, (h3 + n - a) / two * two, MADV_HUGEPAGE);
            double v1 = now(); prefault(h3, n, 8); double v2 = now();
            hipMemcpy(h3, d, n, hipMemcpyDeviceToHost); double v3 = now();
            free(h3); double v4 = now();
            printf("   THP: madvise %.2f ms (rc %d) | prefault %.1f ms | D2H pageable %.1f ms | free %.1f ms\n", v1 - v0, rc, v2 - v1, v3 - v2, v4 - v3);
        }
        char* h2 = (char*)malloc(n);
        double u0 = now(); hipError_t e2 = hipHostRegister(h2, n, hipHostRegisterDefault); double u1 = now();
        hipMemcpy(h2, d, n, hipMemcpyDeviceToHost); double u2 = now();
        hipHostUnregister(h2); free(h2);
        printf("prefault(8 thr) %.1f ms | D2H pageable %.1f ms | register %.1f ms (%d) | D2H registered %.1f ms | unregister %.1f ms | free %.1f ms || fresh: register %.1f ms (%d), D2H %.1f ms\n",
               t1 - t0, t2 - t1, t3 - t2, (int)e, t4 - t3, t5 - t4, t6 - t5, u1 - u0, (int)e2, u2 - u1);
    }
    return 0;
}
