// Micro-benchmark: ceiling of the access pattern of the gather on MI355X: random 256-byte rows of a 256 MB fp32 table
// (1M rows x 64 floats), 16 lanes x 16 B per row, U rows in flight per 16-lane group, nothing else in the kernel.
// hipcc --offload-arch=gfx950 -O3 -o gather_probe gather_probe.hip && ./gather_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));

template <int U>
__global__ void __launch_bounds__(256) k(const int *idx, long n_idx, const float *table, float *out)
{
    const int gl = threadIdx.x & 15;
    const long group = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 4, n_groups = ((long)gridDim.x * blockDim.x) >> 4;
    v4f acc = {0, 0, 0, 0};
    for (long e = group * U; e + U <= n_idx; e += n_groups * U) {
        v4f x[U];
#pragma unroll
        for (int u = 0; u < U; ++u) x[u] = *(const v4f *)(table + (long)idx[e + u] * 64 + gl * 4);
#pragma unroll
        for (int u = 0; u < U; ++u) acc += x[u];
    }
    if (acc.x + acc.y + acc.z + acc.w == 12345.678f) out[0] = acc.x;
}

template <int U>
void run(const int *idx, long n_idx, const float *table, float *out, int blocks)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<U>, blocks, 256, 0, 0, idx, n_idx, table, out);
    hipEventRecord(e0);
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<U>, blocks, 256, 0, 0, idx, n_idx, table, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    ms /= 5;
    printf("U=%2d blocks=%5d : %.3f ms  %.2f TB/s (rows) \n", U, blocks, ms, n_idx * 256.0 / (ms * 1e-3) / 1e12);
}

int main()
{
    const long N = 1000000, E = 10000000;
    std::vector<int> h(E);
    srand(1);
    for (long i = 0; i < E; ++i) h[i] = (int)(((long)rand() * 32768 + rand()) % N);
    int *idx; float *table, *out;
    hipMalloc(&idx, E * 4); hipMalloc(&table, N * 256); hipMalloc(&out, 4096);
    hipMemcpy(idx, h.data(), E * 4, hipMemcpyHostToDevice);
    hipMemset(table, 0, N * 256);
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<4>(idx, E, table, out, blocks);
        run<8>(idx, E, table, out, blocks);
        run<16>(idx, E, table, out, blocks);
        run<32>(idx, E, table, out, blocks);
    }
    return 0;
}
