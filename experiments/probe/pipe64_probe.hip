// Probe for the open design of DESIGN.md 4.1 / VERDICT r3 item 3: 64-node tiles on ONE wave per SIMD (4-wave workgroups, 512 VGPRs), every
// weight fragment feeding two MFMAs (the weight stream per node halves), and - unlike the serial form measured in round 3 (0.87 ms) - the NEXT
// tile's gather software-pipelined under the CURRENT tile's matrix phase by the same wave.
//
// Per 64-node tile and wave, as k_fused's BASELINE shape would have it:
//   * matrix phase: 504 weight fragments (252 KiB of 16-byte-per-lane loads from a packed image every wave reads in the same order: L2 /
//     vector-L1 traffic), each consumed by TWO v_mfma_f32_32x32x16_bf16 (1008 MFMAs on 256 accumulator registers), prefetched WD fragments
//     ahead, V vector-ALU instructions behind every MFMA (activation + piece cutting of k_fused: 5 - 7 per MFMA);
//   * gather of the next tile: 10 batches of 64 random 256-byte rows of a 256 MB table (16 rows per 16-lane group and batch, 16 B per lane),
//     D batches in flight in VGPRs (16 KiB each), every row weighted into a 4-register accumulator (2 packed fmas) and flushed to LDS every
//     10th row; batch b is consumed and batch b + D requested at fixed points of the matrix phase - either all through it (X tile of the
//     next tile double-buffered: not possible at 160 KB, shown as the bound) or only behind layer 0 (the last 57 % of the fragments: the
//     current tile's LDS image is dead by then), RESTRICT;
//   * own rows in (16 KiB coalesced), new rows out (16 KiB coalesced) per tile.
// Reference points in the same probe: k_fused's shape (8 waves per CU, 32-node tiles, gather then matrix phase, 252 KiB per 32 nodes).
// hipcc --offload-arch=gfx950 -O3 -o pipe64_probe pipe64_probe.hip && ./pipe64_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <utility>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ v4f gl4(const float *p) { return *(const GLOBAL v4f *)p; }
template <int J>
__device__ __forceinline__ int row_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true); }
template <int J>
__device__ __forceinline__ float row_bcast_f(float v) { return __int_as_float(row_bcast_i<J>(__float_as_int(v))); }
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
// one batch: entry J of the group's 16 (held by lane J of the 16-lane row) broadcast by DPP, the 16 lanes request the 256-byte row
template <int... J>
__device__ __forceinline__ void gather16(int my_id, float my_w, __amdgpu_buffer_rsrc_t rs, int voff0, v4f (&x)[16], float (&w)[16], std::integer_sequence<int, J...>)
{
    ((w[J] = row_bcast_f<J>(my_w), x[J] = bload(rs, (row_bcast_i<J>(my_id) << 8) + voff0, 0)), ...);
}

// ---- the pipelined one-wave-per-SIMD form -----------------------------------------------------------------------------------------------
template <int D, int V, bool RESTRICT, int WD>
__global__ void __launch_bounds__(256, 1) k_pipe64(const int *idx, const float *ew, int tiles, const float *table, float *out, int *ctr, const v4f *img,
                                                   const float *own, float *dst)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = lane & 15, grp = lane >> 4;
    float *X = lds + (size_t)wave * 64 * 68;                 // aggregated-state block of the wave's tile: 64 rows x 64 floats (+ pad)
    f32x16 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    bf16x8 b0, b1;
#pragma unroll
    for (int i = 0; i < 8; ++i) { b0[i] = (__bf16)0.5f; b1[i] = (__bf16)0.25f; }
    float vv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) vv[q] = (float)(lane + q);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, 256000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4f *>(img), 0, 512 * 1024, 0x00020000);
    int t = 0;
    if (lane == 0) t = atomicAdd(ctr, 1);
    t = __builtin_amdgcn_readfirstlane(t);
    while (t < tiles) {
        int tn = 0;
        if (lane == 0) tn = atomicAdd(ctr, 1);
        tn = __builtin_amdgcn_readfirstlane(tn);
        const int *e = idx + (size_t)(tn < tiles ? tn : t) * 640;     // the NEXT tile's entries (10 batches x 64 rows); last tile: re-gathers its own
        const float *w_e = ew + (size_t)(tn < tiles ? tn : t) * 640;
        // own rows of the current tile (coalesced, 16 KiB): requested first, used in the "epilogue"
        v4f ownr[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) ownr[u] = gl4(own + (size_t)t * 4096 + u * 1024 + lane * 4 + (wave & 0) );
        v4f x[D][16];
        float wq[D][16];
        v2f a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        int flushed = 0;
        auto issue = [&](int b, v4f (&xx)[16], float (&ww)[16]) {
            const int my_id = e[b * 64 + lane];                      // coalesced: lane (grp, gl) holds entry gl of its group's batch
            const float my_w = w_e[b * 64 + lane];
            gather16(my_id, my_w, trs, gl * 16, xx, ww, std::make_integer_sequence<int, 16>{});
        };
        auto consume = [&](v4f (&xx)[16], float (&ww)[16]) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a01 = __builtin_elementwise_fma(v2f{ww[u], ww[u]}, xx[u].lo, a01);
                a23 = __builtin_elementwise_fma(v2f{ww[u], ww[u]}, xx[u].hi, a23);
                if (u == 5 || u == 15) {                              // a row boundary about every 10 entries: flush to the LDS tile
                    *reinterpret_cast<v4f *>(X + (size_t)((flushed & 15) * 4 + grp) * 68 + gl * 4) = v4f{a01.x, a01.y, a23.x, a23.y};
                    a01 = v2f{0.f, 0.f}; a23 = v2f{0.f, 0.f};
                    ++flushed;
                }
            }
        };
        // weight prefetch: WD fragments ahead; the matrix phase runs in chunks of WD steps (register indices static inside a chunk)
        v4f w[WD];
        int pos = 0;
#pragma unroll
        for (int j = 0; j < WD; ++j) { w[j] = bload(wrs, lane * 16, __builtin_amdgcn_readfirstlane(pos * 1024)); ++pos; }
        constexpr int STEPS = 512, NCH = STEPS / WD;                 // (504 fragments rounded up to whole chunks)
        auto chunks = [&](int n) {
            for (int c = 0; c < n; ++c) {
#pragma unroll
                for (int s = 0; s < WD; ++s) {
                    const bf16x8 a = __builtin_bit_cast(bf16x8, w[s]);
                    const int j = s & 7;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < V; ++q) vv[q & 7] = __builtin_fmaf(vv[q & 7], 1.0000001f, 0.5f);
                    acc[8 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b1, acc[8 + j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < V; ++q) vv[(q + 3) & 7] = __builtin_fmaf(vv[(q + 3) & 7], 1.0000001f, 0.5f);
                    w[s] = bload(wrs, lane * 16, __builtin_amdgcn_readfirstlane((pos & 511) * 1024)); ++pos;
                }
            }
        };
        constexpr int FIRST = RESTRICT ? (NCH * 7) / 16 : 0;        // layer 0 is 432 of 1008 MFMAs
        constexpr int GAP = (NCH - FIRST) / 11;
        chunks(FIRST);
#pragma unroll
        for (int d = 0; d < D; ++d) issue(d, x[d], wq[d]);
#pragma unroll
        for (int b = 0; b < 10; ++b) {
            chunks(GAP);
            consume(x[b % D], wq[b % D]);
            if (b + D < 10) issue(b + D, x[b % D], wq[b % D]);
        }
        chunks(NCH - FIRST - 10 * GAP);
        // "epilogue": new rows out (16 KiB per tile, coalesced), own rows consumed
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            v4f o = ownr[u];
            o.x += acc[u][0] + a01.x; o.y += acc[u + 4][1]; o.z += vv[u]; o.w += a23.y;
            *(GLOBAL v4f *)(dst + (size_t)t * 4096 + u * 1024 + lane * 4) = o;
        }
        t = tn;
    }
    float s_ = 0.f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s_ += acc[j][0];
#pragma unroll
    for (int j = 0; j < 8; ++j) s_ += vv[j];
    if (s_ == 12345.678f) out[0] = s_ + X[lane];
}

// ---- k_fused's shape for reference: 8 waves per CU, 32-node tiles, gather (5 batches, one in flight) THEN matrix phase (504 MFMAs, 252 fragments) ----
template <int V, int WD>
__global__ void __launch_bounds__(512, 2) k_ref32(const int *idx, const float *ew, int tiles, const float *table, float *out, int *ctr, const v4f *img, const float *own,
                                                  float *dst)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, gl = lane & 15, grp = lane >> 4;
    float *X = lds + (size_t)wave * 32 * 68;
    f32x16 acc[8];
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    bf16x8 b0;
#pragma unroll
    for (int i = 0; i < 8; ++i) b0[i] = (__bf16)0.5f;
    float vv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) vv[q] = (float)(lane + q);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, 256000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4f *>(img), 0, 512 * 1024, 0x00020000);
    // start-up spread as k_fused
    {
        const int rounds = (int)((((unsigned)blockIdx.x * 8 + (unsigned)wave) * 0x9E3779B1u) >> 16) % 21;
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
    for (;;) {
        int t = 0;
        if (lane == 0) t = atomicAdd(ctr, 1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t >= tiles) break;
        const int *e = idx + (size_t)t * 320;
        const float *w_e = ew + (size_t)t * 320;
        v4f ownr[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) ownr[u] = gl4(own + (size_t)t * 2048 + u * 1024 + lane * 4);
        v2f a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        int flushed = 0;
        for (int b = 0; b < 5; ++b) {
            v4f x[16];
            float ww[16];
            gather16(e[b * 64 + lane], w_e[b * 64 + lane], trs, gl * 16, x, ww, std::make_integer_sequence<int, 16>{});
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a01 = __builtin_elementwise_fma(v2f{ww[u], ww[u]}, x[u].lo, a01);
                a23 = __builtin_elementwise_fma(v2f{ww[u], ww[u]}, x[u].hi, a23);
                if (u == 5 || u == 15) {
                    *reinterpret_cast<v4f *>(X + (size_t)((flushed & 7) * 4 + grp) * 68 + gl * 4) = v4f{a01.x, a01.y, a23.x, a23.y};
                    a01 = v2f{0.f, 0.f}; a23 = v2f{0.f, 0.f};
                    ++flushed;
                }
            }
        }
        v4f w[WD];
        int pos = 0;
#pragma unroll
        for (int j = 0; j < WD; ++j) { w[j] = bload(wrs, lane * 16, __builtin_amdgcn_readfirstlane(pos * 1024)); ++pos; }
        for (int c = 0; c < 256 / WD; ++c) {
#pragma unroll
            for (int s = 0; s < WD; ++s) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, w[s]);
                const int j = s & 3;
                acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < V; ++q) vv[q & 7] = __builtin_fmaf(vv[q & 7], 1.0000001f, 0.5f);
                acc[4 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b0, acc[4 + j], 0, 0, 0);
#pragma unroll
                for (int q = 0; q < V; ++q) vv[(q + 3) & 7] = __builtin_fmaf(vv[(q + 3) & 7], 1.0000001f, 0.5f);
                w[s] = bload(wrs, lane * 16, __builtin_amdgcn_readfirstlane((pos & 255) * 1024)); ++pos;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            v4f o = ownr[u];
            o.x += acc[u][0] + a01.x; o.y += acc[u + 4][1]; o.z += vv[u]; o.w += a23.y;
            *(GLOBAL v4f *)(dst + (size_t)t * 2048 + u * 1024 + lane * 4) = o;
        }
    }
    float s_ = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s_ += acc[j][0] + vv[j];
    if (s_ == 12345.678f) out[0] = s_ + X[lane];
}

template <class K>
float timeit(K launch, int *ctr)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipMemset(ctr, 0, 4); hipDeviceSynchronize();
    launch();
    hipDeviceSynchronize();
    float best = 1e9f;
    for (int r = 0; r < 4; ++r) {
        hipMemset(ctr, 0, 4); hipDeviceSynchronize();
        hipEventRecord(e0);
        launch();
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    hipError_t err = hipGetLastError();
    if (err != hipSuccess) printf("  (HIP error: %s)\n", hipGetErrorString(err));
    return best;
}

int main()
{
    const long N = 1000000;
    const int tiles32 = 31250, tiles64 = 15625;
    const long E = (long)tiles32 * 320;
    std::vector<int> h(E);
    std::vector<float> hw(E);
    srand(1);
    for (long i = 0; i < E; ++i) { h[i] = (int)(((long)rand() * 32768 + rand()) % N); hw[i] = 0.1f; }
    int *idx, *ctr; float *table, *out, *ew, *own, *dst;
    hipMalloc(&idx, E * 4); hipMalloc(&ew, E * 4); hipMalloc(&table, N * 256); hipMalloc(&own, N * 256); hipMalloc(&dst, N * 256); hipMalloc(&out, 4096); hipMalloc(&ctr, 4);
    hipMemcpy(idx, h.data(), E * 4, hipMemcpyHostToDevice);
    hipMemcpy(ew, hw.data(), E * 4, hipMemcpyHostToDevice);
    hipMemset(table, 0, N * 256); hipMemset(own, 0, N * 256);
    v4f *img;
    hipMalloc(&img, 512 * 1024);
    hipMemset(img, 0, 512 * 1024);
    const size_t lds64 = 4 * 64 * 68 * 4, lds32 = 8 * 32 * 68 * 4;
#define RUN64(D, V, R, WD)                                                                                                          \
    {                                                                                                                               \
        float ms = timeit([&] { hipLaunchKernelGGL((k_pipe64<D, V, R, WD>), 256, 256, lds64, 0, idx, ew, tiles64, table, out, ctr, img, own, dst); }, ctr); \
        printf("pipe64  D=%d batches in flight  V=%d VALU/MFMA  gather %s  weights %2d ahead: %.3f ms\n", D, V, R ? "behind layer 0 only" : "all through      ", WD, ms); \
        fflush(stdout);                                                                                                             \
    }
#define RUN32(V, WD)                                                                                                                \
    {                                                                                                                               \
        float ms = timeit([&] { hipLaunchKernelGGL((k_ref32<V, WD>), 256, 512, lds32, 0, idx, ew, tiles32, table, out, ctr, img, own, dst); }, ctr); \
        printf("ref32   k_fused's shape (8 waves, 32-node tiles, serial)  V=%d  weights %2d ahead: %.3f ms\n", V, WD, ms);           \
        fflush(stdout);                                                                                                             \
    }
    RUN32(4, 8) RUN32(6, 8)
    RUN64(1, 4, false, 8) RUN64(2, 4, false, 8) RUN64(2, 6, false, 8)
    RUN64(1, 4, true, 8) RUN64(2, 4, true, 8) RUN64(2, 6, true, 8)
    RUN64(2, 0, true, 8) RUN64(2, 4, true, 16) RUN64(2, 4, false, 16)
    return 0;
}
