// Probe for the round-2 intermittent parity failure (DESIGN.md "Initialisation order"): can a creation-time hipMemset (null stream)
// land AFTER work that a hipStreamNonBlocking stream was given later?  Two sequences of the engine as it was, replayed with plain HIP:
//   A  gnn_graph_derive + gnn_graph_update_labels:   hipMalloc, hipMemset(0), kernel on the non-blocking stream writes the buffer at once
//   B  gnn_loop_create + set_state0 + gnn_loop_run:  hipMalloc x 2, hipMemset(0) x 2, the allocations / pinned allocations / events of
//      gnn_loop_create, an H2D copy + stream synchronisation (set_state0), then a D2D copy into the buffer on the non-blocking stream
// Each is run with an idle device and with a long kernel busy on ANOTHER non-blocking stream (the hardware queues are few and shared).
// Reports how many trials ended with zeros where the stream's data should be.
//   hipcc --offload-arch=gfx950 -O2 -o memset_race_probe tools/memset_race_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

__global__ void k_fill(float *p, size_t n, float v)
{
    const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) p[t] = v;
}
__global__ void k_count_zero(const float *p, size_t n, unsigned long long *out)
{
    size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; t < n; t += (size_t)gridDim.x * blockDim.x) c += p[t] == 0.0f;
    if (c) atomicAdd(out, c);
}
__global__ void k_busy(long long cycles, int *sink)
{
    const long long t0 = wall_clock64();          // 100 MHz
    while (wall_clock64() - t0 < cycles) { }
    if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = 1;
}

static unsigned long long zeros_in(const float *p, size_t n, unsigned long long *d_cnt, hipStream_t st)
{
    unsigned long long h = 0;
    hipMemsetAsync(d_cnt, 0, sizeof(h), st);
    k_count_zero<<<1024, 256, 0, st>>>(p, n, d_cnt);
    hipMemcpyAsync(&h, d_cnt, sizeof(h), hipMemcpyDeviceToHost, st);
    hipStreamSynchronize(st);
    return h;
}

int main()
{
    hipStream_t s_nb, s_other;
    hipStreamCreateWithFlags(&s_nb, hipStreamNonBlocking);
    hipStreamCreateWithFlags(&s_other, hipStreamNonBlocking);
    unsigned long long *d_cnt;
    hipMalloc(&d_cnt, 8);
    int *d_sink;
    hipMalloc(&d_sink, 4);
    const size_t sizes[] = {(size_t)31225, (size_t)4160 * 64, (size_t)1 << 22, (size_t)1 << 26};     // floats: derived labels of the failing test, its state, 16 MB, 256 MB
    const int trials = 200;
    for (int busy = 0; busy < 2; ++busy)
        for (size_t n : sizes) {
            int bad_a = 0, bad_b = 0;
            unsigned long long worst_a = 0, worst_b = 0;
            const int tr = n >= ((size_t)1 << 26) ? 20 : trials;
            for (int t = 0; t < tr; ++t) {
                // ---- A ----
                float *p = nullptr;
                hipMalloc(&p, n * 4);
                if (busy) k_busy<<<1, 64, 0, s_other>>>(200000LL, d_sink);          // ~2 ms on another non-blocking stream
                hipMemset(p, 0, n * 4);
                k_fill<<<(unsigned)((n + 255) / 256), 256, 0, s_nb>>>(p, n, 1.0f);
                hipStreamSynchronize(s_nb);
                hipDeviceSynchronize();
                unsigned long long z = zeros_in(p, n, d_cnt, s_nb);
                if (z) { ++bad_a; if (z > worst_a) worst_a = z; }
                hipFree(p);
                // ---- B ----
                float *st0 = nullptr, *st1 = nullptr, *init = nullptr, *x[6] = {};
                void *pin[2] = {};
                hipEvent_t ev[2];
                std::vector<float> host(n < ((size_t)1 << 22) ? n : ((size_t)1 << 22), 1.0f);
                hipMalloc(&init, host.size() * 4);
                hipMalloc(&st0, n * 4);
                hipMalloc(&st1, n * 4);
                if (busy) k_busy<<<1, 64, 0, s_other>>>(200000LL, d_sink);
                hipMemset(st0, 0, n * 4);
                hipMemset(st1, 0, n * 4);
                for (int i = 0; i < 6; ++i) hipMalloc(&x[i], 4096);
                for (int i = 0; i < 2; ++i) hipHostMalloc(&pin[i], 4096);
                for (int i = 0; i < 2; ++i) hipEventCreate(&ev[i]);
                hipMemcpyAsync(init, host.data(), host.size() * 4, hipMemcpyHostToDevice, s_nb);
                hipStreamSynchronize(s_nb);
                hipMemcpyAsync(st0, init, host.size() * 4, hipMemcpyDeviceToDevice, s_nb);
                hipStreamSynchronize(s_nb);
                hipDeviceSynchronize();
                z = zeros_in(st0, host.size(), d_cnt, s_nb);
                if (z) { ++bad_b; if (z > worst_b) worst_b = z; }
                for (int i = 0; i < 6; ++i) hipFree(x[i]);
                for (int i = 0; i < 2; ++i) hipHostFree(pin[i]);
                for (int i = 0; i < 2; ++i) hipEventDestroy(ev[i]);
                hipFree(st0); hipFree(st1); hipFree(init);
            }
            printf("%s device, %zu floats: A (derive -> relabel) wiped in %d / %d trials (worst %llu zeros); B (loop_create -> run) wiped in %d / %d trials (worst %llu zeros)\n",
                   busy ? "busy" : "idle", n, bad_a, tr, worst_a, bad_b, tr, worst_b);
            fflush(stdout);
        }
    return 0;
}
