// Probe for the last structural candidate of k_fused (VERDICT r4 item 1 / DESIGN.md 8.4): a wave PAIR sharing one 32-node LDS tile, each wave
// gathering 16 of its nodes and producing HALF of every layer's output features; the activations cross between the two waves as bf16 pieces
// through LDS, one LDS flag per hand-over, no s_barrier.
//
// What the split does and does not change per 32-node tile (BASELINE shape, 135 -> 128 -> 128 -> 64):
//   * weight fragments: wave A streams the fragments of its feature half (126 KiB), wave B those of the other half (126 KiB) - 252 KiB per
//     32 nodes, EXACTLY what one wave of k_fused streams per 32 nodes.  The bytes through the CU's vector L1 per node do not change; they
//     would halve only if a pair shared a 64-NODE tile (every fragment feeding two node halves), and that tile + its piece buffers
//     (64 x 128 x 6 B = 48 KiB per layer boundary) does not fit four times into 160 KiB of LDS;
//   * MFMAs: 252 per wave and tile (504 per tile, as before); accumulators 32 + 32 registers instead of 64 + 64;
//   * new: layer-boundary exchange - every wave writes its 64 features x 32 nodes as 3 bf16 pieces (12 ds_write_b128 per lane... 12 KiB) and
//     reads all 128 (24 ds_read_b128 per layer), and the waves meet 5 - 7 times per tile (after the gather, around each exchange, before the row
//     stores, before the tile buffer is re-used).
// This probe puts that structure - real gather of random 256-byte rows, real weight stream, real MFMA count, V vector-ALU fillers per MFMA, the
// LDS exchange and the flag hand-overs - beside k_fused's own shape (ref32: pipe64_probe.hip's reference kernel) in ONE harness.
// hipcc --offload-arch=gfx950 -O3 -o pair_probe pair_probe.hip && ./pair_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <utility>
#include <vector>
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define GLOBAL __attribute__((address_space(1)))

__device__ __forceinline__ v4f gl4(const float *p) { return *(const GLOBAL v4f *)p; }
template <int J>
__device__ __forceinline__ int row_bcast_i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true); }
template <int J>
__device__ __forceinline__ float row_bcast_f(float v) { return __int_as_float(row_bcast_i<J>(__float_as_int(v))); }
__device__ __forceinline__ v4f bload(__amdgpu_buffer_rsrc_t r, int voff, int soff) { return __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0)); }
template <int... J>
__device__ __forceinline__ void gather16(int my_id, float my_w, __amdgpu_buffer_rsrc_t rs, int voff0, v4f (&x)[16], float (&w)[16], std::integer_sequence<int, J...>)
{
    ((w[J] = row_bcast_f<J>(my_w), x[J] = bload(rs, (row_bcast_i<J>(my_id) << 8) + voff0, 0)), ...);
}

// ---- the wave pair -------------------------------------------------------------------------------------------------------------------------
// PX: partner = wave ^ PX (1: the partner sits on another SIMD; 4: the two waves of one SIMD)
// S2: 2 hand-overs per layer boundary (readers done -> writers done: the piece buffer is re-used in place, the LDS budget that fits four pairs
//     per CU) or 1 (two piece buffers: does not fit, shown as the bound)
// XV: vector-ALU instructions a wave issues OUTSIDE the MFMA gaps per tile (the layer-0 operand cut before the first meeting: 220, and the half of
//     every publish step that has no MFMAs of its own wave to hide behind: 2 x 150) - 0 = everything interleaved (the optimistic bound)
template <int V, int WD, int PX, int S2, int XV = 0>
__global__ void __launch_bounds__(512, 2) k_pair32(const int *idx, const float *ew, int tiles, const float *table, float *out, int *ctr, const v4f *img,
                                                   const float *own, float *dst)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), gl = lane & 15, grp = lane >> 4;
    const int partner = wave ^ PX;
    const int pair = PX == 1 ? (wave >> 1) : (wave & 3);
    const int side = PX == 1 ? (wave & 1) : (wave >> 2);            // which half of the tile's nodes / of every layer's features
    // per pair: X tile 32 x 84 floats (the columns the gather writes), piece buffer 8 chunks x 3 pieces x 1 KiB, then flags / ticket slots
    constexpr int PAIR_FLOATS = 32 * 84 + 8 * 3 * 256;
    float *X = lds + (size_t)pair * PAIR_FLOATS;
    v4i *H = reinterpret_cast<v4i *>(X + 32 * 84);
    volatile int *flags = reinterpret_cast<volatile int *>(lds + 4 * PAIR_FLOATS);       // [8] one word per wave, [8..11] next tile of the pair
    if (threadIdx.x < 16) flags[threadIdx.x] = 0;
    __syncthreads();
    int phase = 0;
    auto meet = [&]() {                     // this wave's LDS writes are in order in front of the flag; the partner's are read behind its flag
        ++phase;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) flags[wave] = phase;
        while (flags[partner] < phase) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    f32x16 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
    float vv[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) vv[q] = (float)(lane + q);
    const __amdgpu_buffer_rsrc_t trs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(table), 0, 256000000, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<v4f *>(img), 0, 512 * 1024, 0x00020000);
    {
        const int rounds = (int)((((unsigned)blockIdx.x * 4 + (unsigned)pair) * 0x9E3779B1u) >> 16) % 21;      // start-up spread per PAIR
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
    // the first tile of a pair is static, the following ones are drawn by the pair's leader one tile ahead
    int t = (int)blockIdx.x * 4 + pair;
    const int first_ticket_base = (int)gridDim.x * 4;
    while (t < tiles) {
        int tn = 0x3fffffff;
        if (side == 0 && lane == 0) tn = atomicAdd(ctr, 1) + first_ticket_base;      // result needed at the end of the tile
        // ---- gather: this wave's 16 nodes = 160 entries = 2 or 3 batches of 64 (alternating, so that a pair's tile is 5 batches) ----
        const int nb = ((t + side) & 1) ? 3 : 2;
        const int *e = idx + (size_t)t * 320 + side * 128 + (((t + side) & 1) ? 0 : 0);
        const float *w_e = ew + (size_t)t * 320 + side * 128;
        const v4f ownr = gl4(own + (size_t)t * 2048 + side * 1024 + lane * 4);
        v2f a01 = {0.f, 0.f}, a23 = {0.f, 0.f};
        int flushed = 0;
        for (int b = 0; b < nb; ++b) {
            v4f x[16];
            float ww[16];
            gather16(e[(b % 2) * 64 + lane], w_e[(b % 2) * 64 + lane], trs, gl * 16, x, ww, std::make_integer_sequence<int, 16>{});
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                a01 = __builtin_elementwise_fma(v2f{ww[u], ww[u]}, x[u].lo, a01);
                a23 = __builtin_elementwise_fma(v2f{ww[u], ww[u]}, x[u].hi, a23);
                if (u == 5 || u == 15) {
                    *reinterpret_cast<v4f *>(X + (size_t)(16 * side + (flushed & 3) * 4 + grp) * 84 + 4 + gl * 4) = v4f{a01.x, a01.y, a23.x, a23.y};
                    a01 = v2f{0.f, 0.f}; a23 = v2f{0.f, 0.f};
                    ++flushed;
                }
            }
        }
#pragma unroll
        for (int q = 0; q < XV * 220 / 520; ++q) vv[q & 7] = __builtin_fmaf(vv[q & 7], 1.0000001f, 0.5f);      // layer-0 cut of the own rows
        meet();                                                      // the tile is complete
        // ---- dense layers: fragments of this wave's feature half, every fragment feeds two MFMAs, B operands from LDS ----
        v4f w[WD];
        int pos = side * 128;                                        // the two waves stream different halves of the image
#pragma unroll
        for (int j = 0; j < WD; ++j) { w[j] = bload(wrs, lane * 16, __builtin_amdgcn_readfirstlane((pos & 255) * 1024)); ++pos; }
        auto layer = [&](int frags, auto cf) {                       // frags fragments; a new B operand (3 pieces) every cf fragments (WD is a multiple)
            constexpr int chunk_frags = decltype(cf)::value;
            bf16x8 b[3];
            int c = 0;
            for (int f0 = 0; f0 < frags; f0 += WD) {
#pragma unroll
                for (int s = 0; s < WD; ++s) {
                    if (s % chunk_frags == 0) {
#pragma unroll
                        for (int pc = 0; pc < 3; ++pc) b[pc] = __builtin_bit_cast(bf16x8, H[((c & 7) * 3 + pc) * 64 + lane]);
                        ++c;
                    }
                    const bf16x8 a = __builtin_bit_cast(bf16x8, w[s]);
                    const int j = s & 1;
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[s % 3], acc[j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < V; ++q) vv[q & 7] = __builtin_fmaf(vv[q & 7], 1.0000001f, 0.5f);
                    acc[2 + j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b[(s + 1) % 3], acc[2 + j], 0, 0, 0);
#pragma unroll
                    for (int q = 0; q < V; ++q) vv[(q + 3) & 7] = __builtin_fmaf(vv[(q + 3) & 7], 1.0000001f, 0.5f);
                    w[s] = bload(wrs, lane * 16, __builtin_amdgcn_readfirstlane((pos & 255) * 1024)); ++pos;
                }
            }
        };
        auto publish = [&](int chunks) {                             // this wave's output features as 3 pieces per chunk
#pragma unroll
            for (int q = 0; q < XV * 150 / 520; ++q) vv[q & 7] = __builtin_fmaf(vv[q & 7], 1.0000001f, 0.5f);  // activation + cut of the last output tile
            if (S2 == 2) meet();                                     // both waves are done READING the buffer
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < chunks)
#pragma unroll
                    for (int pc = 0; pc < 3; ++pc)
                        H[(((side * 4 + c) & 7) * 3 + pc) * 64 + lane] = v4i{__float_as_int(acc[c & 3][pc]), __float_as_int(vv[c]), __float_as_int(acc[c & 3][4 + pc]), lane};
            meet();                                                  // both halves are in place
        };
        layer((54 + WD - 1) / WD * WD, std::integral_constant<int, 6>{});                                 // layer 0: 9 chunks x 2 tiles x 3 pieces (rounded up to whole groups of WD)
        publish(4);
        layer(48, std::integral_constant<int, 6>{});                                                // 128 -> 128: 8 chunks x 2 tiles x 3
        publish(4);
        layer(24, std::integral_constant<int, 3>{});                                                // 128 -> 64: 8 chunks x 1 tile x 3
        // ---- epilogue: this wave's 32 features of the new state to LDS, then the rows of its 16 nodes out ----
        if (S2 == 2) meet();
#pragma unroll
        for (int c = 0; c < 4; ++c) H[(side * 4 + c) * 64 + lane] = v4i{__float_as_int(acc[c][0]), __float_as_int(acc[c][1]), __float_as_int(acc[c][2]), __float_as_int(acc[c][3])};
        meet();
        {
            const v4i nv = H[(side * 4 + (lane & 3)) * 64 + (lane ^ 21)];
            v4f o = ownr;
            o.x += __int_as_float(nv.x) + a01.x; o.y += __int_as_float(nv.y); o.z += vv[1]; o.w += a23.y;
            *(GLOBAL v4f *)(dst + (size_t)t * 2048 + side * 1024 + lane * 4) = o;
        }
        // the leader's ticket for the next tile, handed to the partner with the last meeting of the tile
        if (side == 0 && lane == 0) flags[8 + pair] = tn;
        meet();
        t = flags[8 + pair];
        t = __builtin_amdgcn_readfirstlane(t);
    }
    float s_ = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) s_ += acc[j][0];
#pragma unroll
    for (int j = 0; j < 8; ++j) s_ += vv[j];
    if (s_ == 12345.678f) out[0] = s_ + X[lane];
}

// ---- k_fused's shape for reference (as in pipe64_probe.hip): 8 waves per CU, 32-node tiles, gather (5 batches, one in flight) THEN matrix phase ----
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

int main(int argc, char **argv)
{
    const long N = 1000000;
    // tiles: BASELINE size by default; the second argument scales it (mid sizes: 3906 = 125 k nodes)
    const int tiles32 = argc > 1 ? atoi(argv[1]) : 31250;
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
    const size_t lds32 = 8 * 32 * 68 * 4, ldsp = 4 * (32 * 84 + 8 * 3 * 256) * 4 + 64;
    printf("# %d tiles of 32 nodes (%ld nodes), 10 random 256-byte rows per node, 252 KiB of weight fragments and 504 MFMAs per tile\n", tiles32, (long)tiles32 * 32);
#define RUNP(V, WD, PX, S2, ...)                                                                                                         \
    {                                                                                                                               \
        hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pair32<V, WD, PX, S2, ##__VA_ARGS__>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        float ms = timeit([&] { hipLaunchKernelGGL((k_pair32<V, WD, PX, S2, ##__VA_ARGS__>), 256, 512, ldsp, 0, idx, ew, tiles32, table, out, ctr, img, own, dst); }, ctr); \
        printf("pair32  partner = wave ^ %d (%s)  %d meetings per layer boundary  V=%d VALU/MFMA  weights %2d ahead  exposed VALU (%s): %.3f ms\n", PX,   \
               PX == 1 ? "other SIMD" : "same SIMD ", S2, V, WD, #__VA_ARGS__, ms);                                                                \
        fflush(stdout);                                                                                                             \
    }
#define RUN32(V, WD)                                                                                                                \
    {                                                                                                                               \
        float ms = timeit([&] { hipLaunchKernelGGL((k_ref32<V, WD>), 256, 512, lds32, 0, idx, ew, tiles32, table, out, ctr, img, own, dst); }, ctr); \
        printf("ref32   k_fused's shape (8 waves, 32-node tiles, one wave per tile)  V=%d  weights %2d ahead: %.3f ms\n", V, WD, ms); \
        fflush(stdout);                                                                                                             \
    }
    RUN32(4, 8) RUN32(6, 8)
    RUNP(4, 6, 1, 2) RUNP(6, 6, 1, 2) RUNP(4, 6, 4, 2) RUNP(6, 6, 4, 2)
    RUNP(4, 6, 1, 1) RUNP(4, 6, 4, 1)
    RUNP(4, 12, 1, 2) RUNP(6, 12, 1, 2) RUNP(4, 12, 4, 2)
    RUNP(4, 6, 1, 2, 520) RUNP(6, 6, 1, 2, 520) RUNP(3, 6, 1, 2, 520) RUNP(5, 6, 1, 2, 520)
    RUN32(4, 4) RUN32(6, 4) RUN32(4, 8)
    return 0;
}
