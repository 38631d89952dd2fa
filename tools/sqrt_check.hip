// Probe: which float square roots / quotients of the device are correctly rounded (= bit-identical to the host's)?  Result on MI355X,
// ROCm 7.2: __fsqrt_rn differs from sqrtf in 669 172 of 4 194 304 inputs (one ulp); sqrtf, __fdiv_rn and operator/ in none.  The
// engine therefore uses sqrtf.   hipcc --offload-arch=gfx950 -O3 -o sqrt_check tools/sqrt_check.hip && ./sqrt_check
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
__global__ void k(int n, const float *x, float *a, float *b, float *c, float *d)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    a[i] = __fsqrt_rn(x[i]);
    b[i] = sqrtf(x[i]);
    c[i] = __fdiv_rn(1.0f, x[i]);
    d[i] = 1.0f / x[i];
}
int main()
{
    const int n = 1 << 22;
    std::vector<float> x(n), a(n), b(n), c(n), d(n);
    srand(1);
    for (int i = 0; i < n; ++i) x[i] = ldexpf((float)rand() / RAND_MAX + 0.5f, rand() % 40 - 20);
    float *dx, *da, *db, *dc, *dd;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dd, n * 4);
    hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, n / 256, 256, 0, 0, n, dx, da, db, dc, dd);
    hipMemcpy(a.data(), da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b.data(), db, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(c.data(), dc, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    int ea = 0, eb = 0, ec = 0, ed = 0;
    for (int i = 0; i < n; ++i) {
        volatile float hs = sqrtf(x[i]), hd = 1.0f / x[i];
        ea += a[i] != hs; eb += b[i] != hs; ec += c[i] != hd; ed += d[i] != hd;
    }
    printf("mismatches of %d: __fsqrt_rn %d  sqrtf %d  __fdiv_rn %d  operator/ %d\n", n, ea, eb, ec, ed);
    return 0;
}
