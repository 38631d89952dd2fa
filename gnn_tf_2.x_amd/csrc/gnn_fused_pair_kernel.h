// Fused iteration kernel, WAVE-PAIR form (round 5; device code).  Included by gnn_fused_p{2,3}.hip.
//
// Same contract as k_fused<.., SPLIT = true, FULL = true> (gnn_fused_kernel.h): one launch = one iteration of GNN.Loop (reference
// GNN/GNN.py:223-242 + :202-220) in the split arithmetic of the default path (fp32 operands as three exact bf16 pieces on
// v_mfma_f32_32x32x16_bf16), state width 64, 128-wide hidden layers, identical results bit for bit.  What changes is who does what:
//
//   * TWO waves (w, w ^ 1: different SIMDs) share ONE 32-node tile.  Wave `side` gathers the 16 nodes 16 side .. 16 side + 15 (four
//     16-lane groups of 4 nodes each, 16 rows = 16 KiB in flight per wave as before) and produces HALF of every layer's output features:
//     feature tiles {2 side, 2 side + 1} of a hidden layer (32 accumulator registers instead of 64), tile `side` of the last layer.
//   * Activations cross between the two waves through LDS, already cut into bf16 pieces, in the MFMA's B-operand order
//     P[chunk][piece][lane] (one ds_write_b128 / ds_read_b128 per lane, chunk and piece).  Layer 0's operand is cut ONCE, each wave
//     cutting the rows it gathered itself (lane = (node, k-half, chunk parity): all 64 lanes busy).
//   * The waves meet through LDS words (one per wave, monotone phase counters; no s_barrier, the other pairs of the workgroup are
//     never involved): after the layer-0 operand is in place, around every hand-over of activations (readers done -> writers done: the
//     piece buffer is re-used in place, which is what lets four pairs fit into 160 KiB), after the new state is in LDS, and before the tile
//     buffer is re-used: 7 meetings per tile.
//   * The own-state columns of the concat never enter LDS: layer 0's first four chunks are cut from registers (loaded in operand
//     order), and the same registers are the "old state" of the convergence test.
//
// LDS per pair: X'[32][XS] (the 80 columns [nodes | hole | aggregated state | aggregated labels | 0] the gather writes; the new state goes
// over the aggregated-state columns) + P[CH0 = 9][3][64] x 16 B = 10.5 + 27 KiB; four pairs + row pointers + staged vectors = 153 KiB.
//
// What it does NOT change: the weight bytes per node through the vector L1 (each wave streams the fragments of its feature half for every
// tile: 126 KiB per wave and tile = 252 KiB per 32 nodes, as k_fused), the MFMA count per node, the vector-ALU work per node.
#pragma once
#include "gnn_fused_kernel.h"

namespace gnn_fused_dev {

constexpr int GNN_PAIR_CH0 = 9;            // K = 16 chunks of layer 0 (concat width + hole in (128, 144])

// weight fragments of ONE wave's feature half: NL loads of 1 KiB per chunk (2 tiles x 3 pieces, or 1 x 3), chunks STRIDE bytes apart.
// Two scalar offsets (the instruction's immediate offset field ends at 4095), advanced by opaque scalar adds (see WStream: as plain
// constants the offsets of an unrolled layer are hoisted out of the tile loop and spilled).
template <int NL, int STRIDE>
struct WStreamHalf {
    int lo, hi;
    __device__ __forceinline__ WStreamHalf(int start) : lo(start), hi(start + 4096) {}
    __device__ __forceinline__ v4i load(int n, __amdgpu_buffer_rsrc_t r, int voff) const      // n: a constant after unrolling
    {
        return n < 4 ? bload4i(r, voff + 1024 * n, lo) : bload4i(r, voff + 1024 * (n - 4), hi);
    }
    __device__ __forceinline__ void advance()
    {
        asm volatile("s_add_u32 %0, %0, %1" : "+s"(lo) : "n"(STRIDE) : "scc");
        if constexpr (NL > 4) asm volatile("s_add_u32 %0, %0, %1" : "+s"(hi) : "n"(STRIDE) : "scc");
    }
};

// the two waves of a pair meet: this wave's earlier LDS writes are complete before its word moves on, the partner's are read behind its word
__device__ __forceinline__ void pair_meet(volatile int *words, int me, int partner, int &phase, int lane)
{
    ++phase;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) words[me] = phase;
    while (words[partner] - phase < 0) __builtin_amdgcn_s_sleep(1);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
}

// row pointers of the 16 nodes of one side of a tile (17 values, lanes 0..16); rows past the end of the range get the range's last pointer
__device__ __forceinline__ int pair_rowptr_request(const GnnFusedArgs &a, int tile, int side, int lane)
{
    const int64_t r0 = (int64_t)tile * 32 + 16 * side;
    if (r0 >= a.n_rows || tile < 0) return 0;
    const int nv = (int)((a.n_rows - r0) < 16 ? (a.n_rows - r0) : 16);
    return (lane <= nv) ? gstream1(a.indptr + r0 + lane) : 0;
}
__device__ __forceinline__ int pair_rowptr_clamp(const GnnFusedArgs &a, int tile, int side, int lane, int raw)
{
    const int64_t r0 = (int64_t)tile * 32 + 16 * side;
    if (r0 >= a.n_rows || tile < 0) return 0;
    const int nv = (int)((a.n_rows - r0) < 16 ? (a.n_rows - r0) : 16);
    const int last_ip = shfl_i(raw, nv);
    return lane <= nv ? raw : last_ip;
}
// lane group g = lane >> 4 owns rows 4g .. 4g+3 of the side: ids / weights of its first 16 entries
__device__ __forceinline__ void pair_first_ids(const GnnFusedArgs &a, int ip, int lane, int &src, float &w)
{
    const int gl = lane & 15, grp = lane >> 4;
    const int e_begin = shfl_i(ip, grp * 4), e_end = shfl_i(ip, grp * 4 + 4);
    src = 0; w = 0.0f;
    if (e_begin + gl < e_end) { src = gstream1(a.adj_src + e_begin + gl); w = gstream1(a.adj_w + e_begin + gl); }
}

// gather of one side: 16 nodes, four lane groups of 4 nodes, as load_tile_fast64 (batches of 16 rows per group, fmaf chain in stored
// order, flush at every row boundary).  XR = X' + 16 side rows; ca = column of the aggregated-state block in X'.
__device__ __forceinline__ void pair_gather(const GnnFusedArgs &a, float *XR, const int *ipt, int lane, int XS, int ca, int my_src, float my_w)
{
    constexpr int GB = 16;
    const int gl = lane & 15, grp = lane >> 4;
    int node = grp * 4;
    const int node_end = node + 4;
    const int e_end = ipt[node_end];
    int base = ipt[node];
    int next_end = ipt[node + 1];
    v2f acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.state_cur), 0, (int)a.state_bytes, 0x00020000);
    const int voff0 = gl * 16;
    float *xo = XR + ca + gl * 4;
#define GNN_PAIR_ROW_BOUNDARY(e)                                                                \
    while ((e) >= next_end) {                                                                   \
        *reinterpret_cast<v4f *>(xo + node * XS) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};     \
        acc01 = v2f{0.f, 0.f}; acc23 = v2f{0.f, 0.f};                                           \
        ++node;                                                                                 \
        next_end = ipt[node + 1];                                                               \
    }
    for (; base + GB <= e_end; base += GB) {
        float w[GB];
        v4f x[GB];
        gather_batch<GB>(my_src, my_w, rsrc, voff0, w, x, std::make_integer_sequence<int, GB>{});
        const int nb = base + GB + gl;
        my_src = 0; my_w = 0.0f;
        if (nb < e_end) { my_src = gstream1(a.adj_src + nb); my_w = gstream1(a.adj_w + nb); }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            GNN_PAIR_ROW_BOUNDARY(base + u)
            gather_fma(acc01, acc23, w[u], x[u]);
        }
    }
    {
        const int cnt = e_end - base;
        float w[GB];
        v4f x[GB];
        gather_batch<GB>(my_src, my_w, rsrc, voff0, w, x, std::make_integer_sequence<int, GB>{});
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            if (u < cnt) {
                GNN_PAIR_ROW_BOUNDARY(base + u)
                gather_fma(acc01, acc23, w[u], x[u]);
            }
        }
    }
#undef GNN_PAIR_ROW_BOUNDARY
    for (; node < node_end; ++node) {
        *reinterpret_cast<v4f *>(xo + node * XS) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};
        acc01 = v2f{0.f, 0.f}; acc23 = v2f{0.f, 0.f};
    }
}

// the folded SELU between the dense layers (GNN_S1_E of layer_split_from_regs): accumulator v' = log2(e) v  ->  operand of the next layer
template <int ACT>
__device__ __forceinline__ float pair_act(float v)
{
    if constexpr (ACT == GNN_ACT_SELU) {
        constexpr float AL2 = 1.6732632423543772f * 1.44269504088896341f;
        return v > 0.0f ? v : __builtin_fmaf(__builtin_amdgcn_exp2f(v), AL2, -AL2);
    } else return act_fast<ACT>(v);
}

// One dense layer of a wave's feature half.  B operand: the pieces of all 32 nodes in P (CH chunks); A operand: the wave's T tiles of
// the layer, 3 T fragments per chunk, requested DEPTH chunks ahead; accumulators start from the bias.  Fully unrolled.
template <int CH, int T, int STRIDE, int DEPTH>
__device__ __forceinline__ void pair_layer(const v4i *P, int lane, __amdgpu_buffer_rsrc_t wrs, int voff, int soff, f32x16 (&acc)[T],
                                           const float *bias_lds, int jt0, int half)
{
    v4i w[CH][T][3];
    v4i b[CH][3];
    WStreamHalf<3 * T, STRIDE> ws(soff);
#define GNN_PAIR_LOADW(C)                                                                           \
    {                                                                                               \
        _Pragma("unroll") for (int t = 0; t < T; ++t)                                               \
            _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) w[C][t][pc] = ws.load(3 * t + pc, wrs, voff);   \
        ws.advance();                                                                               \
    }
#define GNN_PAIR_LOADB(C) _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) b[C][pc] = P[((C) * 3 + pc) * 64 + lane];
#pragma unroll
    for (int c = 0; c < DEPTH && c < CH; ++c) GNN_PAIR_LOADW(c)
    GNN_PAIR_LOADB(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (c + DEPTH < CH) GNN_PAIR_LOADW(c + DEPTH)
        if (c + 1 < CH) { GNN_PAIR_LOADB(c + 1) }
        __builtin_amdgcn_sched_barrier(0);
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // term order as mfma_split: smallest products first
#pragma unroll
        for (int term = 0; term < 6; ++term)
#pragma unroll
            for (int t = 0; t < T; ++t)
                acc[t] = mfma_bf16(w[c][t][PA[term]], b[c][PB[term]], (c == 0 && term == 0) ? bias_tile(bias_lds, jt0 + t, half) : acc[t]);
        __builtin_amdgcn_sched_barrier(0);
    }
#undef GNN_PAIR_LOADW
#undef GNN_PAIR_LOADB
}

// activation + cut of a wave's T output tiles into operand pieces (registers): chunk 2 t + q of the wave's half = accumulator registers 8 q .. 8 q + 7 of tile t
template <int T, int ACT>
__device__ __forceinline__ void pair_cut(f32x16 (&h)[T], v4i (&pp)[2 * T][3])
{
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = pair_act<ACT>(h[t][8 * q + i]);
            split8(v, pp[2 * t + q][0], pp[2 * t + q][1], pp[2 * t + q][2]);
        }
}

template <int LAYERS, int ACT>
__global__ void __launch_bounds__(GNN_FUSED_THREADS, 2) k_fused_pair(const GnnFusedArgs a0)
{
    constexpr int NT = 4, NTL = 2, CH0 = GNN_PAIR_CH0;
    const GnnFusedArgs &a = a0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!gnn_gate_open(a.gate, a.world)) return;
#ifdef GNN_DIAG      // GNN_POISON=1: NaN over the whole LDS allocation before anything is staged (a read of a never-written word shows as a NaN)
    if (a.lds_floats) {
        for (int t = threadIdx.x; t < a.lds_floats; t += blockDim.x) lds[t] = __builtin_nanf("");
        __syncthreads();
    }
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int px = (a.variant & 512) ? 4 : 1;                     // partner on another SIMD (default) or the SIMD's other wave (experiment)
    const int partner = wave ^ px;
    const int pair = px == 1 ? (wave >> 1) : (wave & 3), side = px == 1 ? (wave & 1) : (wave >> 2);
    const int XS = a.KP;                                          // row stride of X' (floats; XS / 4 odd)
    const int ca = a.c_aggs - 64;                                 // column of the aggregated-state block in X'
    const int PAIRF = 32 * XS + CH0 * 768;
    float *X = lds + (size_t)pair * PAIRF;
    float *XR = X + 16 * side * XS;                               // this side's 16 rows
    v4i *P = reinterpret_cast<v4i *>(X + 32 * XS);
    float *tail = lds + (size_t)4 * PAIRF;
    volatile int *words = reinterpret_cast<volatile int *>(tail);            // [8] phase word per wave, [8 .. 11] next-but-one tile of the pair
    int *ipt = reinterpret_cast<int *>(tail + 16) + wave * 20;
    float *ep = tail + 16 + GNN_FUSED_WAVES * 20;                 // last layer's bias, BatchNormalization scale / shift: [3][32 NTL]
    float *hb = ep + 3 * 32 * NTL;                                // hidden biases [LAYERS - 1][32 NT] (x log2(e): folded SELU)
    const int PAIRS = (int)gridDim.x * 4;                         // pairs of the launch
    const int p_launch = pair * (int)gridDim.x + (int)blockIdx.x; // pair-major over the workgroups: a partial round spreads over all CUs
    const bool third_round = (int64_t)2 * PAIRS * 32 < a.n_rows;
    int tile = p_launch, next_tile = p_launch + PAIRS;
    const int ip_first_raw = pair_rowptr_request(a, tile, side, lane);
    if (threadIdx.x < 16) words[threadIdx.x] = 0;
    for (int t = threadIdx.x; t < 3 * 32 * NTL; t += blockDim.x) {
        const int which = t / (32 * NTL), f = t - which * 32 * NTL;
        ep[t] = which == 0 ? a.bias[LAYERS - 1][f] : (a.bn_scale ? (which == 1 ? a.bn_scale[f] : a.bn_shift[f]) : 0.0f);
    }
    for (int t = threadIdx.x; t < (LAYERS - 1) * 32 * NT; t += blockDim.x)
        hb[t] = a.bias[t / (32 * NT)][t % (32 * NT)] * (ACT == GNN_ACT_SELU ? 1.44269504088896341f : 1.0f);
    // zero columns of this side's rows, once: the alignment hole, the padding behind the concat (nothing in a tile's life writes them)
    {
        const int hole0 = a.NLc, holew = ca - hole0, pad0 = a.in_s - 64, padw = 16 * CH0 - 64 - pad0;
        for (int t = lane; t < 16 * (holew + padw); t += 64) {
            const int r = t / (holew + padw), c = t - r * (holew + padw);
            XR[r * XS + (c < holew ? hole0 + c : pad0 + (c - holew))] = 0.0f;
        }
    }
    __syncthreads();
    if (a.stagger > 0) {
        const int rounds = (int)((((unsigned)blockIdx.x * 4 + (unsigned)pair) * 0x9E3779B1u) >> 16) % (unsigned)(a.stagger + 1);
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int phase = 0;
    int ip_cur = pair_rowptr_clamp(a, tile, side, lane, ip_first_raw);
    int src_cur = 0;
    float w_cur = 0.0f;
    pair_first_ids(a, ip_cur, lane, src_cur, w_cur);
    const int half = lane >> 5;
    // lane roles of the layer-0 cut / the convergence test: node nl of the side, k-half h, chunk parity cp
    const int nl = lane & 15, ch = (lane >> 5) & 1, cp = (lane >> 4) & 1;
  for (;;) {
    const int64_t i0 = (int64_t)tile * 32;
    if (i0 >= a.n_rows) break;                        // the same for both waves of the pair
    const int nvalid = (int)((a0.n_rows - i0) < 32 ? (a0.n_rows - i0) : 32);
    GnnFusedArgs a = a0;
    asm volatile("" : "+s"(a.state_cur), "+s"(a.state_nxt), "+s"(a.inv), "+s"(a.adj_src), "+s"(a.adj_w));
#ifndef GNN_DIAG
#define GNN_PSTAMP(slot) do { } while (0)
#else     // diagnostic build, GNN_FUSED_STAMPS=<file>: s_memtime of side 0 at the phase boundaries of a tile (16 slots per tile)
    unsigned long long *stamp = (a.stamps && side == 0) ? a.stamps + ((size_t)tile << 4) : nullptr;
#define GNN_PSTAMP(slot)                                                                     \
    do {                                                                                     \
        if (stamp) {                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
            if (lane == 0) stamp[slot] = t_;                                                 \
            __builtin_amdgcn_sched_barrier(0);                                               \
        }                                                                                    \
    } while (0)
#endif
    GNN_PSTAMP(0);
    if (lane <= 16) ipt[lane] = ip_cur;
    const int ip_next_raw = pair_rowptr_request(a, next_tile, side, lane);
    // ---- own state of this side's 16 rows in operand order: floats [16 c + 8 h + 4 j', +4) of row nl for c = cp, cp + 2 ----
    v4f own[4];
    {
        const float *src = a.state_cur + (a.row_begin + i0 + 16 * side + nl) * 64 + 16 * cp + 8 * ch;
#pragma unroll
        for (int j = 0; j < 4; ++j) own[j] = gload4(src + 32 * (j >> 1) + 4 * (j & 1));
    }
    float lab[2];
    const int IW = a.IW, nlab = 16 * IW;
    {
        const float *src = a.inv + (i0 + 16 * side) * IW;
#pragma unroll
        for (int u = 0; u < 2; ++u) lab[u] = (lane + 64 * u < nlab) ? gload1(src + lane + 64 * u) : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                   // ipt visible to the whole wave
    if (a.variant & 1) __builtin_amdgcn_s_setprio(3);
    pair_gather(a, XR, ipt, lane, XS, ca, src_cur, w_cur);
    if (a.variant & 1) __builtin_amdgcn_s_setprio(0);
    {   // label columns of this side's rows: [nodes] in front, [agg nodes | agg arcs] behind the aggregated state
        const float inv_iw = 1.0f / (float)(IW > 0 ? IW : 1);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int t = lane + 64 * u;
            if (t < nlab) {
                const int i = (int)(((float)t + 0.5f) * inv_iw), c = t - i * IW;
                XR[i * XS + (c < a.NLc ? c : ca + 64 + (c - a.NLc))] = lab[u];
            }
        }
        if (nlab > 128)
            for (int t = 128 + lane; t < nlab; t += 64) {
                const int i = t / IW, c = t - i * IW;
                XR[i * XS + (c < a.NLc ? c : ca + 64 + (c - a.NLc))] = gload1(a.inv + (i0 + 16 * side) * IW + t);
            }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    GNN_PSTAMP(1);
    // ---- layer-0 operand: this side's rows cut into pieces, chunks of parity cp per lane ----
    {
        const int pl = 16 * side + nl + 32 * ch;                  // MFMA lane (node of the tile, k-half) this lane produces the pieces of
#pragma unroll
        for (int cc = 0; cc < (CH0 + 1) / 2; ++cc) {
            const int c = 2 * cc + cp;                            // (wave-divergent only between the two lane quarters)
            float v[8];
            if (cc < 2) {
                const v4f lo = own[2 * cc], hi = own[2 * cc + 1];
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            } else {
                const float *xr = XR + nl * XS + 16 * (c - 4) + 8 * ch;
                const int cq = c < CH0 ? 0 : -16;                 // (parity 1 has no chunk CH0: re-reads its last one, result unused)
                const v4f lo = *reinterpret_cast<const v4f *>(xr + cq), hi = *reinterpret_cast<const v4f *>(xr + cq + 4);
                v[0] = lo.x; v[1] = lo.y; v[2] = lo.z; v[3] = lo.w; v[4] = hi.x; v[5] = hi.y; v[6] = hi.z; v[7] = hi.w;
            }
            v4i p0, p1, p2;
            split8(v, p0, p1, p2);
            if (c < CH0) {
                P[(c * 3 + 0) * 64 + pl] = p0;
                P[(c * 3 + 1) * 64 + pl] = p1;
                P[(c * 3 + 2) * 64 + pl] = p2;
            }
        }
    }
    GNN_PSTAMP(2);
    pair_meet(words, wave, partner, phase, lane);                 // (1) layer-0 operand complete
    GNN_PSTAMP(3);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.Ws_base), 0, a.ws_bytes, 0x00020000);
    const int wv = lane * 16;
    f32x16 out[1];
    {
        f32x16 h1[2];
        pair_layer<CH0, 2, NT * 3072, 3>(P, lane, wrs, wv, a.ws_off[0] + side * 6144, h1, hb, 2 * side, half);
        GNN_PSTAMP(4);
        v4i pp[4][3];
        pair_cut<2, ACT>(h1, pp);
        GNN_PSTAMP(5);
        pair_meet(words, wave, partner, phase, lane);             // (2) both waves have read the layer-0 operand
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) P[((4 * side + c) * 3 + pc) * 64 + lane] = pp[c][pc];
    }
    pair_meet(words, wave, partner, phase, lane);                 // (3) hidden activations of layer 0 complete
    GNN_PSTAMP(6);
    if constexpr (LAYERS == 3) {
        f32x16 h2[2];
        pair_layer<2 * NT, 2, NT * 3072, 3>(P, lane, wrs, wv, a.ws_off[1] + side * 6144, h2, hb + 32 * NT, 2 * side, half);
        GNN_PSTAMP(7);
        v4i pp[4][3];
        pair_cut<2, ACT>(h2, pp);
        GNN_PSTAMP(8);
        pair_meet(words, wave, partner, phase, lane);             // (4)
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int pc = 0; pc < 3; ++pc) P[((4 * side + c) * 3 + pc) * 64 + lane] = pp[c][pc];
        pair_meet(words, wave, partner, phase, lane);             // (5)
        GNN_PSTAMP(9);
    }
    pair_layer<2 * NT, 1, NTL * 3072, 4>(P, lane, wrs, wv, a.ws_off[LAYERS - 1] + side * 3072, out, ep, side, half);
    GNN_PSTAMP(10);
    // ---- requests behind the last weight loads (vector-memory results return in order): ticket, next tile's first ids, gate words ----
    int next2_tile = 0x3fffffff;
    if (third_round && side == 0 && lane == 0) next2_tile = atomicAdd(a0.tile_ctr, 1) + 2 * PAIRS;
    const int ip_next = pair_rowptr_clamp(a, next_tile, side, lane, ip_next_raw);
    int src_next = 0;
    float w_next = 0.0f;
    pair_first_ids(a, ip_next, lane, src_next, w_next);
    GnnFlagPeek peek = {0, 0, 0};
    if (lane == 0) peek = gnn_flag_peek(a.flag_out);
    // ---- last-layer epilogue of this wave's 32 features, new state into X' over the aggregated-state columns ----
    if (a.bn_scale) tile_epilogue<ACT, true, true, true, true>(out[0], ep, ep + 32 * NTL, ep + 64 * NTL, side, half);
    else tile_epilogue<ACT, false, true, true, true>(out[0], ep, nullptr, nullptr, side, half);
    {
        float *xrow = X + (lane & 31) * XS + ca + 32 * side + 4 * half;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<v4f *>(xrow + 8 * q) = v4f{out[0][4 * q], out[0][4 * q + 1], out[0][4 * q + 2], out[0][4 * q + 3]};
    }
    GNN_PSTAMP(11);
    pair_meet(words, wave, partner, phase, lane);                 // (6) the new state of all 32 nodes is in LDS
    GNN_PSTAMP(12);
    {   // condition() of this side's 16 nodes: lane (nl, ch, cp) holds the old features [16 c + 8 ch, +8), c = cp, cp + 2
        float d2 = 0.0f, o2 = 0.0f;
        const float *xn = XR + nl * XS + ca + 16 * cp + 8 * ch;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const v4f nw = *reinterpret_cast<const v4f *>(xn + 32 * (j >> 1) + 4 * (j & 1));
            const v4f o = own[j], d = nw - o;
            d2 = __builtin_fmaf(d.x, d.x, d2); d2 = __builtin_fmaf(d.y, d.y, d2); d2 = __builtin_fmaf(d.z, d.z, d2); d2 = __builtin_fmaf(d.w, d.w, d2);
            o2 = __builtin_fmaf(o.x, o.x, o2); o2 = __builtin_fmaf(o.y, o.y, o2); o2 = __builtin_fmaf(o.z, o.z, o2); o2 = __builtin_fmaf(o.w, o.w, o2);
        }
        d2 = d2 + shfl_f(d2, lane ^ 16); o2 = o2 + shfl_f(o2, lane ^ 16);
        d2 = d2 + shfl_f(d2, lane ^ 32); o2 = o2 + shfl_f(o2, lane ^ 32);
        const float root = sqrtf(d2), nrm = sqrtf(o2);
        const bool voter = 16 * side + nl < nvalid;
        const float rhs = a.thr * nrm, band = GNN_BAND_ABS * nrm + GNN_BAND_REL * rhs;
        const bool am = __any(voter && root > rhs), ar = __any(voter && root > rhs + band), ab = __any(voter && gnn_gate_borderline(root, rhs, band));
        if (lane == 0) gnn_flag_raise_peeked(a.flag_out, peek, am, ar, ab);
        // coalesced row stores of this side's 16 rows
        const int64_t dst = (i0 + 16 * side) * 64 + lane * 4;                // flat element 256 u + 4 lane = row 4 u + lane / 16
        const float *xs = XR + (lane >> 4) * XS + ca + (lane & 15) * 4;
        v4f v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4f *>(xs + 4 * u * XS);
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (16 * side + 4 * u + (lane >> 4) < nvalid) gstore_row4(a.state_nxt, dst + 256 * u, v[u]);
    }
    GNN_PSTAMP(13);
    if (side == 0 && lane == 0) words[8 + pair] = next2_tile;
    pair_meet(words, wave, partner, phase, lane);                 // (7) the tile buffers are free; the leader's ticket is in its slot
    GNN_PSTAMP(14);
#undef GNN_PSTAMP
    tile = next_tile;
    next_tile = third_round ? __builtin_amdgcn_readfirstlane(words[8 + pair]) : 0x3fffffff;
    ip_cur = ip_next; src_cur = src_next; w_cur = w_next;
  }
}

template <int LAYERS, int ACT>
inline void launch_pair_one(const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused_pair<LAYERS, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) raised[dev] = true;
    }
    hipLaunchKernelGGL((k_fused_pair<LAYERS, ACT>), grid, GNN_FUSED_THREADS, lds_bytes, st, a);
}

template <int LAYERS>
inline bool launch_pair_act(int act, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    switch (act) {
    case GNN_ACT_LINEAR: launch_pair_one<LAYERS, GNN_ACT_LINEAR>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_RELU: launch_pair_one<LAYERS, GNN_ACT_RELU>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_SELU: launch_pair_one<LAYERS, GNN_ACT_SELU>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_ELU: launch_pair_one<LAYERS, GNN_ACT_ELU>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_TANH: launch_pair_one<LAYERS, GNN_ACT_TANH>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_SIGMOID: launch_pair_one<LAYERS, GNN_ACT_SIGMOID>(a, grid, lds_bytes, st); return true;
    default: return false;
    }
}

}   // namespace gnn_fused_dev
