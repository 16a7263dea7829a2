// cost of device allocations of the sizes the plans use (round 5: why plan creation for a chunk of 16 designs takes 30 ms ... 1.6 s)
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    hipFree(0);
    for (size_t mb : {1, 4, 32, 256, 2048, 6000}) {
        const int n = mb <= 32 ? 64 : (mb <= 256 ? 16 : 2);
        std::vector<void*> p((size_t)n);
        for (int rep = 0; rep < 2; ++rep) {
            double t0 = now();
            for (int i = 0; i < n; ++i) if (hipMalloc(&p[(size_t)i], mb << 20) != hipSuccess) { printf("alloc failed\n"); return 1; }
            double t1 = now();
            for (int i = 0; i < n; ++i) hipMemsetAsync(p[(size_t)i], 0, mb << 20, 0);
            hipDeviceSynchronize();
            double t2 = now();
            for (int i = 0; i < n; ++i) hipFree(p[(size_t)i]);
            double t3 = now();
            printf("%5zu MB x %2d (rep %d): hipMalloc %.3f ms each, first memset %.3f ms each, hipFree %.3f ms each\n", mb, n, rep, (t1 - t0) / n, (t2 - t1) / n, (t3 - t2) / n);
        }
    }
    return 0;
}
