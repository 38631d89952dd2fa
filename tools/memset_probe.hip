// Probe: does hipMemset on device memory return before the fill has run?  (It is queued on the null stream; the engine's loops work on
// non-blocking streams, which the null stream does not order.)   hipcc --offload-arch=gfx950 -O2 -o memset_probe tools/memset_probe.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <stdio.h>
int main()
{
    const size_t bytes = (size_t)8 << 30;
    char *p = nullptr;
    if (hipMalloc(&p, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; }
    hipMemset(p, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        hipMemset(p, 0, bytes);
        auto t1 = std::chrono::steady_clock::now();
        hipDeviceSynchronize();
        auto t2 = std::chrono::steady_clock::now();
        printf("hipMemset of 8 GiB returned after %.3f ms; device idle %.3f ms later\n", std::chrono::duration<double, std::milli>(t1 - t0).count(),
               std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
    return 0;
}
