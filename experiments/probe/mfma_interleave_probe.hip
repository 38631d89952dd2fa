// Micro-benchmark: does VALU work placed between bf16 MFMAs of the SAME wave overlap with them on gfx950?
// Patterns per iteration and wave (2 accumulators alternating, like the hidden layers of the split kernel):
//   M: 24 MFMA 32x32x16 bf16        V: 24 x NV independent VALU fma        I: the same interleaved (1 MFMA, NV VALU, ...)
// hipcc --offload-arch=gfx950 -O3 -o mfma_interleave_probe mfma_interleave_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int NV>   // MODE 1 = M, 2 = V, 3 = interleaved
__global__ void __launch_bounds__(512, 2) k(int iters, float *out, float seed)
{
    f32x16 acc[2];
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) acc[t][r] = seed * (t + r);
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(seed + i + threadIdx.x); b[i] = (__bf16)(seed * i); }
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i + threadIdx.x;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 24; ++m) {
            if (MODE & 1) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 1], 0, 0, 0);
            if (MODE & 2) {
#pragma unroll
                for (int q = 0; q < NV; ++q) v[q] = __builtin_fmaf(v[q], 1.0001f, 0.5f);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float s = 0;
    for (int t = 0; t < 2; ++t) for (int r = 0; r < 16; ++r) s += acc[t][r];
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE, int NV>
float run(int blocks, int threads, int iters, float *out)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<MODE, NV>), blocks, threads, 0, 0, 10, out, 1.0f);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<MODE, NV>), blocks, threads, 0, 0, iters, out, 1.0f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

template <int NV>
void report(const char *name, int blocks, int threads, float *out)
{
    const int iters = 4000;
    const float m = run<1, NV>(blocks, threads, iters, out), v = run<2, NV>(blocks, threads, iters, out), i = run<3, NV>(blocks, threads, iters, out);
    printf("%s NV=%d: MFMA only %.3f ms, VALU only %.3f ms, interleaved %.3f ms (sum %.3f, max %.3f)\n", name, NV, m, v, i, m + v, m > v ? m : v);
}

int main()
{
    float *out; hipMalloc(&out, 256 * 512 * 4);
    // 256 blocks = one per CU; 256 threads = 1 wave per SIMD, 512 threads = 2 waves per SIMD
    report<4>("1 wave/SIMD ", 256, 256, out);
    report<7>("1 wave/SIMD ", 256, 256, out);
    report<4>("2 waves/SIMD", 256, 512, out);
    report<7>("2 waves/SIMD", 256, 512, out);
    return 0;
}
