// Micro-benchmark: issue rates of f32 vs bf16 MFMA on gfx950 and whether VALU work of the partner wave overlaps them.
// hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip && ./mfma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // bit0: bf16 mfma, bit1: f32 mfma, bit2: valu
__global__ void __launch_bounds__(512, 2) k(int iters, float *out, float seed)
{
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = seed * (t + r);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i + threadIdx.x); b[i] = (__bf16)(seed * i); }
    float v[16];
    for (int i = 0; i < 16; ++i) v[i] = seed + i + threadIdx.x;
    float fa = seed, fb = seed * 2;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int rep = 0; rep < 6; ++rep)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[t], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, acc[t], 0, 0, 0);
        }
        if (MODE & 4) {
#pragma unroll
            for (int rep = 0; rep < 8; ++rep)
#pragma unroll
                for (int i = 0; i < 16; ++i) v[i] = __builtin_fmaf(v[i], 1.0001f, 0.5f);
        }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
float run(int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, 512, 512, 0, 0, 10, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, 512, 512, 0, 0, iters, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main()
{
    float *out; hipMalloc(&out, 512 * 512 * 4);
    const int iters = 2000;
    // per iteration and wave: bf16: 24 MFMA 32x32x16 (= K 96 on 4 tiles); f32: 32 MFMA 32x32x2 (= K 16 on 4 tiles); valu: 128 fma
    // 512 blocks x 8 waves = 4096 waves on 1024 SIMDs: 2 resident per SIMD, 2 rounds
    float t1 = run<1>(iters, out), t2 = run<2>(iters, out), t4 = run<4>(iters, out), t5 = run<5>(iters, out), t6 = run<6>(iters, out);
    const double waves_per_simd = 4.0;
    auto cyc = [&](float ms, double n) { return ms * 1e-3 * 2.4e9 / (iters * waves_per_simd * n); };
    printf("bf16 32x32x16 : %.3f ms  -> %.1f cycles/MFMA (at 2.4 GHz)\n", t1, cyc(t1, 24));
    printf("f32  32x32x2  : %.3f ms  -> %.1f cycles/MFMA\n", t2, cyc(t2, 32));
    printf("valu fma      : %.3f ms  -> %.2f cycles/op\n", t4, cyc(t4, 128));
    printf("bf16 + valu   : %.3f ms  (sum of parts %.3f, max %.3f)\n", t5, t1 + t4, t1 > t4 ? t1 : t4);
    printf("f32  + valu   : %.3f ms  (sum of parts %.3f, max %.3f)\n", t6, t2 + t4, t2 > t4 ? t2 : t4);
    return 0;
}
