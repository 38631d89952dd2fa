// EXPERIMENT, DIAGNOSTIC BUILD ONLY (round 3; `make DIAG=1`, GNN_FUSED_TILE16=1).  Result on MI355X, bench workload at N nodes
// (tools/bench_midsize.py, ms per iteration, same box, k_fused 32-node tiles / this kernel):
//     N = 31 k: 0.093 / 0.115     62 k: 0.128 / 0.164     125 k: 0.154 / 0.214     250 k: 0.227 / 0.311     500 k: 0.357 / 0.550
// Correct (one body within 9e-7 of the exact path, the split-path tolerance tests pass with it), and slower everywhere: each wave
// streams the 270 KB weight image per 16 instead of 32 nodes, 2 GB per launch at N = 125 k = 8 MB per CU against a vector-L1 fill rate
// of 64 B / clk: 61 us of L1 time alone, and a ring of five 3 KiB units per wave is all the look-ahead 168 VGPRs leave.  Kept as the
// record of the attempt.
//
// 16-node-tile form of the fused iteration kernel (split arithmetic, state width 64) for launches that are bound by the LATENCY of a
// tile rather than by throughput: graphs of 10^4 .. 3 10^5 nodes (the per-rank shards of BASELINE configs[3], LGNN layers on
// mid-size graphs), where a 256-CU launch of k_fused has only one to four 32-node tiles per wave.
//
// One launch = one iteration of GNN.Loop (reference GNN/GNN.py:223-242 + :202-220), same arithmetic contract as k_fused<.., SPLIT>:
// every fp32 operand cut into three exact bf16 pieces, six piece products per term, fp32 accumulation (gnn_fused_kernel.h).
// What differs is the tile shape.  k_fused: 32 nodes per wave on v_mfma_f32_32x32x16_bf16, two 64-register accumulator sets,
// 256 VGPRs and 18.9 KB of LDS per wave => 2 waves per SIMD, and a tile takes ~40 us from ticket to store.  Here: 16 nodes per wave
// on v_mfma_f32_16x16x32_bf16 (same FLOP rate per SIMD): accumulators are 4 registers per 16 x 16 tile, so a hidden layer of 128
// features is 32 registers, the whole wave fits in < 168 VGPRs and 9.5 KB of LDS => 3 waves per SIMD (12 per CU), twice as many tiles
// of half the work each.  The price is the weight stream: every wave reads the bf16-piece image once per 16 instead of 32 nodes
// (2 x the L1 / L2 traffic per node), which is why the host selects this kernel only for small grids (gnn_fused.hip).
//
// Operand layout of the 16x16x32 MFMA (cdna_hip_programming.md, MFMA operand layouts): lane l carries A[row l & 15][k = 8 (l >> 4) + j],
// B[k = 8 (l >> 4) + j][col l & 15], j < 8, and D[row 4 (l >> 4) + r][col l & 15], r < 4.  Weights are A (row = output feature), node
// activations B (col = node): lane (node n, group g) ends a layer holding features 16 t + 4 g + r of node n for every output tile t.
// As in k_fused the k order of a hidden layer is free as long as A and B agree, so a K = 32 chunk c of a hidden layer takes the two
// accumulator tiles 2 c, 2 c + 1 as they are:    k slot (g, i) = feature 32 c + 16 (i >> 2) + 4 g + (i & 3)      - no cross-lane
// exchange between layers.  Layer 0 reads 8 consecutive floats of the lane's LDS row: k slot (g, i) = LDS column 32 c + 8 g + i.
// gnn_fused_pack writes a second image in exactly this order: [chunk][out tile][piece][lane][8 bf16].
#pragma once
#include "../gnn_fused_kernel.h"

namespace gnn_fused_dev {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(v4i a, v4i b, f32x4 c)
{
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// One dense layer on 16-node tiles: CH chunks of K = 32, NO output tiles of 16 features, fully unrolled.  The weight stream is read
// strictly in image order [chunk][out tile][piece] through a ring of R single-tile units (3 x 16 B per lane each): the loads of unit
// u + R - 1 are issued before the MFMAs of unit u, i.e. (R - 1) x 6 MFMAs (~ 100 cycles each unit) ahead - with three waves per SIMD the
// other waves cover what that does not.  operand(c, dst): the three bf16 pieces of this lane's 8 input values of chunk c; the pieces of
// chunk c + 1 are cut before the MFMAs of chunk c are issued.
template <int CH, int NO, class OperandFn>
__device__ __forceinline__ void dense16(OperandFn operand, f32x4 (&acc)[NO], const float *bias_lds, int g, __amdgpu_buffer_rsrc_t wrs, int voff, int soff)
{
    constexpr int U = CH * NO, R = 5;
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};      // smallest products first, as mfma_split
#pragma unroll
    for (int t = 0; t < NO; ++t) acc[t] = *reinterpret_cast<const f32x4 *>(bias_lds + 16 * t + 4 * g);     // accumulators start from the bias
    v4i w[R][3];
    v4i b[2][3];
    WStream ws(soff);
#pragma unroll
    for (int u = 0; u < R - 1 && u < U; ++u)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) w[u % R][pc] = ws.next(wrs, voff);
    operand(0, b[0]);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        if (c + 1 < CH) operand(c + 1, b[(c + 1) & 1]);
#pragma unroll
        for (int t = 0; t < NO; ++t) {
            const int u = c * NO + t;
            if (u + R - 1 < U) {
#pragma unroll
                for (int pc = 0; pc < 3; ++pc) w[(u + R - 1) % R][pc] = ws.next(wrs, voff);
            }
#pragma unroll
            for (int term = 0; term < 6; ++term) acc[t] = mfma16(w[u % R][PA[term]], b[c & 1][PB[term]], acc[t]);
        }
    }
}

// layer 0: input = the wave's LDS tile, xr = X + (lane & 15) * KP + 8 * (lane >> 4); always GNN_F16_CH0 chunks (the image is zero
// beyond the net's width).  Columns >= in_cols do not exist in the tile: their operand values are forced to zero (their weights are
// zero too, but whatever the LDS holds there could be a NaN pattern).
constexpr int GNN_F16_CH0 = 5;                      // K = 160 >= the widest concat the fused path covers (144) + alignment hole
template <int NF>
__device__ __forceinline__ void layer0_split16(const float *xr, int in_cols, __amdgpu_buffer_rsrc_t wrs, int voff, int soff, f32x4 (&acc)[NF],
                                               const float *bias_lds, int g)
{
    auto operand = [&](int c, v4i (&dst)[3]) {
        const v4f lo = *reinterpret_cast<const v4f *>(xr + 32 * c), hi = *reinterpret_cast<const v4f *>(xr + 32 * c + 4);
        float xv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
        if (32 * c + 32 > 128) {                    // (compile time per chunk: the concat is at least 2 Ds = 128 columns wide)
            const int col0 = 32 * c + 8 * g;
#pragma unroll
            for (int i = 0; i < 8; ++i) xv[i] = (col0 + i < in_cols) ? xv[i] : 0.0f;
        }
        split8(xv, dst[0], dst[1], dst[2]);
    };
    dense16<GNN_F16_CH0, NF>(operand, acc, bias_lds, g, wrs, voff, soff);
}

// hidden / last layer: input = the previous layer's accumulators (bias included).  Activation of the split path (folded SELU: the
// image carries log2(e) on the producing and scale / log2(e) on the consuming layer, see gnn_fused_pack and GNN_S1_E), then the cut
// into bf16 pieces.
template <int NI, int NO, int ACT>
__device__ __forceinline__ void layer16_from_regs(f32x4 (&hin)[NI], const float *bias_lds, int g, f32x4 (&acc)[NO], __amdgpu_buffer_rsrc_t wrs,
                                                  int voff, int soff)
{
    static_assert(NI % 2 == 0, "K = 32 chunks are pairs of input tiles");
    auto operand = [&](int c, v4i (&dst)[3]) {
        float v[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float x = hin[2 * c + (i >> 2)][i & 3];
            if constexpr (ACT == GNN_ACT_SELU) {
                constexpr float AL2 = 1.6732632423543772f * 1.44269504088896341f;
                v[i] = x > 0.0f ? x : __builtin_fmaf(__builtin_amdgcn_exp2f(x), AL2, -AL2);
            } else
                v[i] = act_fast<ACT>(x);
        }
        split8(v, dst[0], dst[1], dst[2]);
    };
    dense16<NI / 2, NO>(operand, acc, bias_lds, g, wrs, voff, soff);
}

// Ds == 64 gather for a 16-node tile: lane group g (16 lanes, 16 B per lane = one 256 B state row per group and instruction) owns the
// 4 consecutive nodes 4 g .. 4 g + 3 and walks their contiguous CSR entries in batches of GB (see load_tile_fast64; same fmaf chain in
// stored order).  The tile's padding columns are zeroed once per wave by the caller.
template <int GB>
__device__ __forceinline__ void load_tile16(const GnnFusedArgs &a, float *X, const int *ipt, int64_t i0, int lane, int KP, int c_aggs, int my_src,
                                            float my_w)
{
    constexpr int Ds = 64;
    const int gl = lane & 15, grp = lane >> 4;
    v4f own[4];
    {
        const float *src = a.state_cur + (a.row_begin + i0) * Ds + lane * 4;
#pragma unroll
        for (int u = 0; u < 4; ++u) own[u] = gload4(src + u * 256);
    }
    float lab[4];
    const int IW = a.IW, nlab = 16 * IW;
    {
        const float *src = a.inv + i0 * IW;
#pragma unroll
        for (int u = 0; u < 4; ++u) lab[u] = (lane + 64 * u < nlab) ? gload1(src + lane + 64 * u) : 0.0f;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                   // ipt visible to the whole wave
    int node = grp * 4;
    const int node_end = node + 4;
    const int e_begin = ipt[node], e_end = ipt[node_end];
    int next_end = ipt[node + 1];
    v2f acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.state_cur), 0, (int)a.state_bytes, 0x00020000);
    const int voff0 = gl * 16;
    float *xo = X + c_aggs + gl * 4;
#define GNN_ROW_BOUNDARY16(e)                                                                   \
    while ((e) >= next_end) {                                                                   \
        *reinterpret_cast<v4f *>(xo + node * KP) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};     \
        acc01 = v2f{0.f, 0.f}; acc23 = v2f{0.f, 0.f};                                           \
        ++node;                                                                                 \
        next_end = ipt[node + 1];                                                               \
    }
    int base = e_begin;
    for (; base + GB <= e_end; base += GB) {                                 // full batches: no guards
        float w[GB];
        v4f x[GB];
        gather_batch<GB>(my_src, my_w, rsrc, voff0, w, x, std::make_integer_sequence<int, GB>{});
        const int nb = base + GB + gl;                                       // ids / weights of the next batch (lanes gl < GB carry them)
        my_src = 0; my_w = 0.0f;
        if (gl < GB && nb < e_end) { my_src = gload1(a.adj_src + nb); my_w = gload1(a.adj_w + nb); }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            GNN_ROW_BOUNDARY16(base + u)
            acc01 = __builtin_elementwise_fma(v2f{w[u], w[u]}, x[u].lo, acc01);
            acc23 = __builtin_elementwise_fma(v2f{w[u], w[u]}, x[u].hi, acc23);
        }
    }
    {                                                                        // tail batch: cnt in [0, GB)
        const int cnt = e_end - base;
        float w[GB];
        v4f x[GB];
        gather_batch<GB>(my_src, my_w, rsrc, voff0, w, x, std::make_integer_sequence<int, GB>{});      // (slots >= cnt re-read a valid row, unused)
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            if (u < cnt) {
                GNN_ROW_BOUNDARY16(base + u)
                acc01 = __builtin_elementwise_fma(v2f{w[u], w[u]}, x[u].lo, acc01);
                acc23 = __builtin_elementwise_fma(v2f{w[u], w[u]}, x[u].hi, acc23);
            }
        }
    }
#undef GNN_ROW_BOUNDARY16
    for (; node < node_end; ++node) {                                        // last row with entries, then empty rows
        *reinterpret_cast<v4f *>(xo + node * KP) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};
        acc01 = v2f{0.f, 0.f}; acc23 = v2f{0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < 4; ++u)                                              // own state: flat element 256 u + 4 lane = row 4 u + lane / 16
        *reinterpret_cast<v4f *>(X + (4 * u + (lane >> 4)) * KP + (lane & 15) * 4) = own[u];
    const float inv_iw = 1.0f / (float)(IW > 0 ? IW : 1);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = lane + 64 * u;
        if (t < nlab) {
            const int i = (int)(((float)t + 0.5f) * inv_iw), c = t - i * IW;
            X[i * KP + label_col(c, Ds, a.NLc, c_aggs)] = lab[u];
        }
    }
    if (nlab > 256)
        for (int t = 256 + lane; t < nlab; t += 64) {
            const int i = t / IW, c = t - i * IW;
            X[i * KP + label_col(c, Ds, a.NLc, c_aggs)] = gload1(a.inv + i0 * IW + t);
        }
}

// ids / weights of the first gather batch of a 16-node tile: lane group g owns rows 4 g .. 4 g + 3
template <int GB>
__device__ __forceinline__ void tile16_first_ids(const GnnFusedArgs &a, int ip, int lane, int &src, float &w)
{
    const int gl = lane & 15, grp = lane >> 4;
    const int e_begin = shfl_i(ip, grp * 4), e_end = shfl_i(ip, grp * 4 + 4);
    src = 0; w = 0.0f;
    if (gl < GB && e_begin + gl < e_end) { src = gload1(a.adj_src + e_begin + gl); w = gload1(a.adj_w + e_begin + gl); }
}

__device__ __forceinline__ int tile16_rowptr_request(const GnnFusedArgs &a, int tile, int lane)
{
    const int64_t i0 = (int64_t)tile * 16;
    if (i0 >= a.n_rows) return 0;
    const int nvalid = (int)((a.n_rows - i0) < 16 ? (a.n_rows - i0) : 16);
    return (lane <= nvalid) ? gload1(a.indptr + i0 + lane) : 0;
}
__device__ __forceinline__ int tile16_rowptr_clamp(const GnnFusedArgs &a, int tile, int lane, int raw)
{
    const int64_t i0 = (int64_t)tile * 16;
    if (i0 >= a.n_rows) return 0;
    const int nvalid = (int)((a.n_rows - i0) < 16 ? (a.n_rows - i0) : 16);
    const int last_ip = shfl_i(raw, nvalid);
    return lane <= nvalid ? raw : last_ip;
}

// last-layer epilogue (activation, BatchNormalization), condition() for the next body and coalesced row stores.  out[t][r] = feature
// 16 t + 4 g + r of node lane & 15; each lane sums its 16 features of (new - old)^2 and old^2, the four lanes of a node are added with
// two cross-row exchanges.  Rows >= nvalid (partial last tile) neither vote nor get stored.
template <int ACT>
__device__ __forceinline__ void finish16(const GnnFusedArgs &a, float *X, f32x4 (&out)[4], const float *ep, int64_t i0, int lane, int KP, int c_aggs,
                                         int nvalid)
{
    const int g = lane >> 4, node = lane & 15;
    float *xrow = X + node * KP;
    float d2 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int f0 = 16 * t + 4 * g;
        v4f nw;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float v = act_fast<ACT>(out[t][r]);
            if (a.bn_scale) { const float m = v * ep[64 + f0 + r]; v = m + ep[128 + f0 + r]; }
            nw[r] = v;
        }
        const v4f o = *reinterpret_cast<const v4f *>(xrow + f0);
        *reinterpret_cast<v4f *>(xrow + c_aggs + f0) = nw;
        const v4f d = nw - o;
        d2 = __builtin_fmaf(d.x, d.x, d2); d2 = __builtin_fmaf(d.y, d.y, d2); d2 = __builtin_fmaf(d.z, d.z, d2); d2 = __builtin_fmaf(d.w, d.w, d2);
        o2 = __builtin_fmaf(o.x, o.x, o2); o2 = __builtin_fmaf(o.y, o.y, o2); o2 = __builtin_fmaf(o.z, o.z, o2); o2 = __builtin_fmaf(o.w, o.w, o2);
    }
    d2 = d2 + shfl_f(d2, lane ^ 16); o2 = o2 + shfl_f(o2, lane ^ 16);
    d2 = d2 + shfl_f(d2, lane ^ 32); o2 = o2 + shfl_f(o2, lane ^ 32);
    const float root = sqrtf(d2), nrm = sqrtf(o2);
    const int moved = (node < nvalid) && (root > a.thr * nrm);
    if (__any(moved) && lane == 0) gnn_flag_raise(a.flag_out);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float *dst = a.state_nxt + i0 * 64 + lane * 4;                           // flat element 256 u + 4 lane = row 4 u + lane / 16
    const float *xs = X + (lane >> 4) * KP + c_aggs + (lane & 15) * 4;
    v4f v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = *reinterpret_cast<const v4f *>(xs + 4 * u * KP);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (4 * u + (lane >> 4) < nvalid) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(dst) + 256 * u) = v[u];
}

constexpr int GNN_F16_WAVES = 12;                  // 3 per SIMD
constexpr int GNN_F16_GB = 8;                      // neighbour rows in flight per lane group (8 KiB per wave, 96 KiB per CU)

// LAYERS in {2, 3}; NF: 16-feature tiles of every hidden layer (4 or 8); the last layer has 4 (state width 64).
template <int LAYERS, int NF, int ACT>
__global__ void __launch_bounds__(64 * GNN_F16_WAVES) k_fused16(const GnnFusedArgs a0)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!gnn_gate_open(a0.gate, a0.world)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = a0.KP, c_aggs = a0.c_aggs;
    float *X = lds + (size_t)wave * 16 * KP;
    float *tail = lds + (size_t)GNN_F16_WAVES * 16 * KP + 32;                // (32 floats of slack: layer 0's last chunk reads past the last row)
    int *ipt = reinterpret_cast<int *>(tail) + wave * 20;
    float *ep = tail + GNN_F16_WAVES * 20;                                   // last layer: bias | BN scale | BN shift, 64 each
    float *hb = ep + 192;                                                    // hidden biases [LAYERS - 1][16 NF]
    for (int t = threadIdx.x; t < 192; t += blockDim.x) {
        const int which = t >> 6, f = t & 63;
        ep[t] = which == 0 ? a0.bias[LAYERS - 1][f] : (a0.bn_scale ? (which == 1 ? a0.bn_scale[f] : a0.bn_shift[f]) : 0.0f);
    }
    for (int t = threadIdx.x; t < (LAYERS - 1) * 16 * NF; t += blockDim.x)
        hb[t] = a0.bias[t / (16 * NF)][t % (16 * NF)] * (ACT == GNN_ACT_SELU ? 1.44269504088896341f : 1.0f);      // folded SELU (GNN_S1_E)
    __syncthreads();
    if (a0.stagger > 0) {
        const int rounds = (int)((((unsigned)blockIdx.x * GNN_F16_WAVES + (unsigned)wave) * 0x9E3779B1u) >> 16) % (unsigned)(a0.stagger + 1);
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int tile = 0, next_tile = 0;
    if (lane == 0) { tile = atomicAdd(a0.tile_ctr, 1); next_tile = atomicAdd(a0.tile_ctr, 1); }
    tile = __builtin_amdgcn_readfirstlane(tile);
    next_tile = __builtin_amdgcn_readfirstlane(next_tile);
    int ip_cur = tile16_rowptr_clamp(a0, tile, lane, tile16_rowptr_request(a0, tile, lane));
    int src_cur = 0;
    float w_cur = 0.0f;
    tile16_first_ids<GNN_F16_GB>(a0, ip_cur, lane, src_cur, w_cur);
    {   // padding columns of the tile and the alignment hole: zeroed once (no tile ever writes them)
        GnnFusedArgs az = a0;
        const int padw = KP - az.in_s;
        for (int t = lane; t < 16 * padw; t += 64) X[(t / padw) * KP + az.in_s + (t % padw)] = 0.0f;
        const int hole0 = az.Ds + az.NLc, holew = az.c_aggs - hole0;
        if (holew > 0 && lane < 16)
            for (int c = 0; c < holew; ++c) X[lane * KP + hole0 + c] = 0.0f;
    }
    const int g = lane >> 4;
  for (;;) {
    const int64_t i0 = (int64_t)tile * 16;
    if (i0 >= a0.n_rows) break;                       // wave-uniform
    const int nvalid = (int)((a0.n_rows - i0) < 16 ? (a0.n_rows - i0) : 16);
    GnnFusedArgs a = a0;                              // fresh, compiler-opaque copies per tile (see k_fused)
    asm volatile("" : "+s"(a.state_cur), "+s"(a.state_nxt), "+s"(a.inv), "+s"(a.adj_src), "+s"(a.adj_w));
    if (lane <= 16) ipt[lane] = ip_cur;
    const int ip_next_raw = tile16_rowptr_request(a, next_tile, lane);
    if (a.variant & 1) __builtin_amdgcn_s_setprio(3);
    load_tile16<GNN_F16_GB>(a, X, ipt, i0, lane, KP, c_aggs, src_cur, w_cur);
    if (a.variant & 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.Ws_base), 0, a.ws_bytes, 0x00020000);
    const int wv = lane * 16;
    const float *xr = X + (lane & 15) * KP + 8 * g;
    f32x4 out[4];
    {
        f32x4 h1[NF];
        layer0_split16<NF>(xr, a.in_s, wrs, wv, a.ws_off[0], h1, hb, g);
        if constexpr (LAYERS == 2) {
            layer16_from_regs<NF, 4, ACT>(h1, ep, g, out, wrs, wv, a.ws_off[1]);
        } else {
            f32x4 h2[NF];
            layer16_from_regs<NF, NF, ACT>(h1, hb + 16 * NF, g, h2, wrs, wv, a.ws_off[1]);
            layer16_from_regs<NF, 4, ACT>(h2, ep, g, out, wrs, wv, a.ws_off[2]);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // requests for the tile after next (ticket) and the next tile (first gather ids): behind the last weight loads (see k_fused)
    int next2_tile = 0;
    if (lane == 0) next2_tile = atomicAdd(a0.tile_ctr, 1);
    const int ip_next = tile16_rowptr_clamp(a, next_tile, lane, ip_next_raw);
    int src_next = 0;
    float w_next = 0.0f;
    tile16_first_ids<GNN_F16_GB>(a, ip_next, lane, src_next, w_next);
    finish16<ACT>(a, X, out, ep, i0, lane, KP, c_aggs, nvalid);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the next tile re-uses this wave's LDS region
    tile = next_tile;
    next_tile = __builtin_amdgcn_readfirstlane(next2_tile);
    ip_cur = ip_next; src_cur = src_next; w_cur = w_next;
  }
}

template <int LAYERS, int NF, int ACT>
inline void launch16_one(const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused16<LAYERS, NF, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) raised[dev] = true;
    }
    hipLaunchKernelGGL((k_fused16<LAYERS, NF, ACT>), grid, 64 * GNN_F16_WAVES, lds_bytes, st, a);
}

template <int LAYERS>
inline bool launch16_act(int act, int nf, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
#define GNN_F16_CASE(A)                                                                             \
    case A:                                                                                         \
        if (nf == 8) launch16_one<LAYERS, 8, A>(a, grid, lds_bytes, st);                            \
        else if (nf == 4) launch16_one<LAYERS, 4, A>(a, grid, lds_bytes, st);                       \
        else return false;                                                                          \
        return true;
    switch (act) {
        GNN_F16_CASE(GNN_ACT_LINEAR) GNN_F16_CASE(GNN_ACT_RELU) GNN_F16_CASE(GNN_ACT_SELU) GNN_F16_CASE(GNN_ACT_ELU) GNN_F16_CASE(GNN_ACT_TANH)
        GNN_F16_CASE(GNN_ACT_SIGMOID)
    default: return false;
    }
#undef GNN_F16_CASE
}

}   // namespace gnn_fused_dev
