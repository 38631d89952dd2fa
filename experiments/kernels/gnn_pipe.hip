// EXPERIMENT, NOT BUILT, NOT SHIPPED (round 2).  Result on MI355X, BASELINE configs[1] (983 040 nodes, 15.7 M edges):
//   stage 1 (synchronous gather per tile)              1.257 ms / launch   (k_fused, the shipped kernel: 0.70 ms)
//   stage 2 (this file: cross-tile gather pipeline)    1.51 ms  (38 spilled VGPRs; a scratch reload waits for vmcnt(0) and so for
//                                                       every neighbour row in flight - the overlap never happens)
//   the same with no neighbour rows requested at all   1.42 ms  = the floor of the barrier-separated dense stages
// The kernel is correct (the 36 split-path parity tests pass with it), but one workgroup per CU (132 KB of LDS, 256 VGPRs x 8
// waves) runs its seven stages strictly one after another: nothing else is resident to fill the time a stage spends in LDS
// round trips and barriers, and the dense stages alone take twice k_fused's whole launch.  Kept as the record of the attempt
// (DESIGN.md 4.1); to try it again: add the file to SRC, declare gnn_pipe_launch in gnn_fused.h and call it from
// gnn_fused_iteration for (split, 3 layers, NT 4, NTL 2, Ds 64, <= 9 layer-0 chunks, IW <= 16) on the full tiles.
//
// Weight-stationary iteration kernel for the BASELINE net shape (state width 64, net_state [<= 144] -> 128 -> 128 -> 64, split
// arithmetic): one launch = one iteration of GNN.Loop (reference GNN/GNN.py:223-242 + :202-220), like k_fused, but organised
// around the workgroup instead of the wave.
//
// k_fused gives every wave a whole 32-node tile and streams the 282 KB bf16-piece weight image past it, per tile and wave: 8 GB
// per launch through the vector L1, two 64-register accumulator sets per wave, and two in-order waves per SIMD that cannot keep
// HBM, the matrix pipe and the VALU busy at the same time.  Here the eight waves of a workgroup work on ONE
// tile at a time and every wave keeps ITS share of the weight image in registers for the whole launch (120 - 132 VGPRs: the
// image is read once per workgroup, not once per tile and wave):
//   gather   wave w owns nodes 4w .. 4w+3 of the tile, one per 16-lane group: CSR gather of the neighbour rows (fmaf chain in
//            stored order), own state row and label columns; every fp32 value is cut into its three bf16 pieces on the spot and
//            stored in LDS in MFMA B-operand order ([chunk][piece][k half][node][8 k])
//   layer 0  wave w computes output tile w & 3 over its half of K (chunks 0-4 / 5-8); the two partial tiles of a pair (w, w + 4)
//            are exchanged through LDS, each wave finishes one half of the registers: activation, pieces of the hidden
//            activations to LDS
//   layer 1  the same with 4 + 4 chunks of K = 128
//   layer 2  wave w: output tile w & 1 over a quarter of K; the four partials are added in a fixed order, last epilogue
//            (BatchNormalization), relative-L2 test per node with a fixed-order reduction, coalesced row stores
// Stages are separated by workgroup barriers.  The neighbour rows of the NEXT tile are requested before the dense layers of the
// current one (16 rows per lane group in registers), so HBM latency overlaps the matrix work of the same waves.
// Arithmetic: as k_fused's split mode (three exact bf16 pieces per fp32 operand, six piece products, fp32 accumulate); partial
// sums over K are added in a fixed order, so results are run-to-run identical and within the same tolerance of the oracle.
#include "gnn_fused_kernel.h"

namespace gnn_fused_dev {

constexpr int PIPE_WAVES = 8, PIPE_THREADS = 64 * PIPE_WAVES;
constexpr int PIPE_C0 = 9;                 // K = 16 chunks of layer 0 (concat width incl. alignment hole <= 144)
// LDS map (dwords)
constexpr int L_XP = 0;                                    // layer-0 operand pieces [9][3][256]
constexpr int L_H1 = L_XP + PIPE_C0 * 3 * 256;             // hidden-1 pieces [8][3][256]
constexpr int L_H2 = L_H1 + 8 * 3 * 256;                   // hidden-2 pieces [8][3][256]
constexpr int L_PART = L_H2 + 8 * 3 * 256;                 // partial accumulators [8 waves][16][64]
constexpr int L_OLD = L_PART + PIPE_WAVES * 16 * 64;       // old state f32 [32][64]
constexpr int L_NEW = L_OLD + 32 * 64;                     // new state f32 [32][64]
constexpr int L_NORM = L_NEW + 32 * 64;                    // norm partials [32 nodes][16][2]
constexpr int L_BIAS = L_NORM + 32 * 16 * 2;               // biases [128 + 128 + 64], BN scale / shift [64 + 64]
constexpr int L_END = L_BIAS + 128 + 128 + 64 + 64 + 64;

// Workgroup barrier that orders LDS traffic only: __syncthreads() also waits for every outstanding vector-memory operation
// (s_waitcnt vmcnt(0)), i.e. for the neighbour rows of the NEXT tile that are meant to stay in flight across the dense layers.
__device__ __forceinline__ void pipe_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <class F, int... J>
__device__ __forceinline__ void for_each_slot(F &&f, std::integer_sequence<int, J...>)
{
    (f(std::integral_constant<int, J>{}), ...);
}

__device__ __forceinline__ void pieces3(float v0, float v1, float v2, float v3, unsigned (&lo)[3], unsigned (&hi)[3])
{
    int p0, p1, p2;
    split_pair(v0, v1, p0, p1, p2);
    lo[0] = (unsigned)p0; lo[1] = (unsigned)p1; lo[2] = (unsigned)p2;
    split_pair(v2, v3, p0, p1, p2);
    hi[0] = (unsigned)p0; hi[1] = (unsigned)p1; hi[2] = (unsigned)p2;
}

// four consecutive columns col .. col + 3 (col % 4 == 0) of node `node` -> the three piece blocks of their chunk
__device__ __forceinline__ void store_pieces4(int *blk, int col, int node, float v0, float v1, float v2, float v3)
{
    unsigned lo[3], hi[3];
    pieces3(v0, v1, v2, v3, lo, hi);
    const int c = col >> 4, k = col & 15;
    int *p = blk + c * 3 * 256 + (k >> 3) * 128 + node * 4 + ((k & 7) >> 1);     // dword index inside the piece block: [half][node][4 dwords]
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) *reinterpret_cast<uint2 *>(p + pc * 256) = uint2{lo[pc], hi[pc]};
}

// one column -> its bf16 slot in the three piece blocks (label columns)
__device__ __forceinline__ void store_piece1(int *blk, int col, int node, float v)
{
    const unsigned a = __float_as_uint(v);
    const float r1 = v - __uint_as_float(a & 0xffff0000u);
    const unsigned b = __float_as_uint(r1);
    const float r2 = r1 - __uint_as_float(b & 0xffff0000u);
    const unsigned pcs[3] = {a >> 16, b >> 16, __float_as_uint(r2) >> 16};
    const int c = col >> 4, k = col & 15;
    unsigned short *p = reinterpret_cast<unsigned short *>(blk + c * 3 * 256 + (k >> 3) * 128 + node * 4) + (k & 7);
#pragma unroll
    for (int pc = 0; pc < 3; ++pc) p[pc * 512] = (unsigned short)pcs[pc];
}

// n_active (wave-uniform) <= NCH chunks are multiplied
template <int NCH>
__device__ __forceinline__ void dense_part(const int *blk, int c0, const v4i (&w)[NCH][3], int lane, f32x16 &acc, int n_active = NCH)
{
    const int *bp = blk + (lane >> 5) * 128 + (lane & 31) * 4;
#pragma unroll
    for (int ci = 0; ci < NCH; ++ci) {
        if (ci >= n_active) break;
        const int *b = bp + (c0 + ci) * 3 * 256;
        const v4i b0 = *reinterpret_cast<const v4i *>(b), b1 = *reinterpret_cast<const v4i *>(b + 256), b2 = *reinterpret_cast<const v4i *>(b + 512);
        acc = mfma_bf16(w[ci][0], b2, acc);          // smallest terms first (as mfma_split)
        acc = mfma_bf16(w[ci][2], b0, acc);
        acc = mfma_bf16(w[ci][1], b1, acc);
        acc = mfma_bf16(w[ci][0], b1, acc);
        acc = mfma_bf16(w[ci][1], b0, acc);
        acc = mfma_bf16(w[ci][0], b0, acc);
    }
}

// hidden activation of 8 accumulator registers (one K = 16 chunk of the next layer) -> pieces in LDS
template <int ACT>
__device__ __forceinline__ void hidden_out(int *blk, int chunk, int lane, const float (&v)[8])
{
    float a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        if constexpr (ACT == GNN_ACT_SELU) {          // folded SELU (gnn_fused_pack scales the image to match)
            constexpr float AL2 = 1.6732632423543772f * 1.44269504088896341f;
            a[i] = v[i] > 0.0f ? v[i] : __builtin_fmaf(__builtin_amdgcn_exp2f(v[i]), AL2, -AL2);
        } else
            a[i] = act_fast<ACT>(v[i]);
    }
    v4i p0, p1, p2;
    split8(a, p0, p1, p2);
    int *p = blk + chunk * 3 * 256 + (lane >> 5) * 128 + (lane & 31) * 4;
    *reinterpret_cast<v4i *>(p) = p0;
    *reinterpret_cast<v4i *>(p + 256) = p1;
    *reinterpret_cast<v4i *>(p + 512) = p2;
}

template <int ACT>
__global__ void __launch_bounds__(PIPE_THREADS, 2) k_pipe(const GnnFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) int lds[];
    if (!gnn_gate_open(a.gate, a.world)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int half = lane >> 5, gl = lane & 15, grp = lane >> 4;
    float *fl = reinterpret_cast<float *>(lds);
    const int NLc = a.NLc, AL = a.AL, IW = a.IW, c_aggs = a.c_aggs;

    // ---- once per workgroup: biases / BN into LDS, zero the piece columns that no tile writes, this wave's weights into registers
    for (int t = threadIdx.x; t < 128 + 128 + 64 + 64 + 64; t += PIPE_THREADS) {
        float v;
        if (t < 128) v = a.bias[0][t] * (ACT == GNN_ACT_SELU ? 1.44269504088896341f : 1.0f);
        else if (t < 256) v = a.bias[1][t - 128] * (ACT == GNN_ACT_SELU ? 1.44269504088896341f : 1.0f);
        else if (t < 320) v = a.bias[2][t - 256];
        else if (t < 384) v = a.bn_scale ? a.bn_scale[t - 320] : 1.0f;
        else v = a.bn_shift ? a.bn_shift[t - 384] : 0.0f;
        fl[L_BIAS + t] = v;
    }
    for (int t = threadIdx.x; t < PIPE_C0 * 3 * 256; t += PIPE_THREADS) lds[L_XP + t] = 0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.Ws_base), 0, a.ws_bytes, 0x00020000);
    const int jt01 = wave & 3, kp = wave >> 2;            // layers 0 / 1: output tile, K half
    const int jt2 = wave & 1, kq = wave >> 1;             // layer 2: output tile, K quarter
    v4i w0[5][3], w1[4][3], w2[2][3];
#pragma unroll
    for (int ci = 0; ci < 5; ++ci)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) {
            const int c = kp * 5 + ci;                   // chunks 0-4 | 5-9 (chunk 9: zero slack of the image, never multiplied)
            w0[ci][pc] = bload4i(wrs, lane * 16, a.ws_off[0] + (((c < PIPE_C0 ? c : PIPE_C0) * 4 + jt01) * 3 + pc) * 1024);
        }
#pragma unroll
    for (int ci = 0; ci < 4; ++ci)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) w1[ci][pc] = bload4i(wrs, lane * 16, a.ws_off[1] + (((kp * 4 + ci) * 4 + jt01) * 3 + pc) * 1024);
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
        for (int pc = 0; pc < 3; ++pc) w2[ci][pc] = bload4i(wrs, lane * 16, a.ws_off[2] + (((kq * 2 + ci) * 2 + jt2) * 3 + pc) * 1024);
    const __amdgpu_buffer_rsrc_t srs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.state_cur), 0, (int)a.state_bytes, 0x00020000);
    __syncthreads();

    const int n_tiles = (int)(a.n_rows >> 5);
    const int node = 4 * wave + grp;                                      // this lane group's node of every tile
    // ---- software pipeline of the gather over the workgroup's tiles (tile j of this workgroup: blockIdx.x + j gridDim.x) --------
    // A tile's gather is a chain of three dependent loads (row pointers -> ids / weights -> neighbour rows).  Each link is
    // requested one tile earlier than the next, so that a tile finds the complete aggregate of its nodes in registers:
    //   entering tile t:  agg(t), own(t), lab(t) ready | ids0(t+1), e(t+1) loaded | e(t+2) loaded
    //   during tile t:    rows of the first batch of t+1 + own(t+1) + lab(t+1) requested after the pieces of t are written, ids0(t+2)
    //                     and e(t+3) requested with them; consumed (and the batches beyond 16 entries gathered) behind the last
    //                     dense layer of t
    auto row_of = [&](int t) -> int64_t { return (int64_t)(t < n_tiles ? t : n_tiles - 1) * 32 + node; };      // clamped: valid memory
    auto load_e = [&](int t, int &e0, int &e1) { const int64_t r = row_of(t); e0 = gload1(a.indptr + r); e1 = gload1(a.indptr + r + 1); };
    auto load_ids = [&](int e0, int e1, int &sr, float &wt) {
        sr = 0; wt = 0.0f;
        if (e0 + gl < e1) { sr = gload1(a.adj_src + e0 + gl); wt = gload1(a.adj_w + e0 + gl); }
    };
    v2f acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};           // agg of the CURRENT tile (complete at loop entry)
    v4f own;
    float lab;
    v4f x[16];
    int e0n, e1n, e0n2, e1n2, e0n3, e1n3, srcn, srcn2;
    float wn, wn2;
    // consume the 16 prefetched rows of a tile (weights broadcast from the id register), then whatever lies beyond 16 entries
    auto finish_gather = [&](int e0, int e1, int sr0, float w0_) {
        acc01 = v2f{0.f, 0.f}; acc23 = v2f{0.f, 0.f};
        auto step = [&](auto uc) {
            constexpr int U = decltype(uc)::value;
            if (e0 + U < e1) {
                const float wv = row_bcast_f<U>(w0_);
                acc01 = __builtin_elementwise_fma(v2f{wv, wv}, x[U].lo, acc01);
                acc23 = __builtin_elementwise_fma(v2f{wv, wv}, x[U].hi, acc23);
            }
        };
        for_each_slot(step, std::make_integer_sequence<int, 16>{});
        for (int base = e0 + 16; base < e1; base += 16) {                  // rare: more than 16 entries
            int ms; float mw;
            load_ids(base, e1, ms, mw);
            float w[16];
            v4f xx[16];
            gather_batch<16>(ms, mw, srs, gl * 16, w, xx, std::make_integer_sequence<int, 16>{});
#pragma unroll
            for (int u = 0; u < 16; ++u)
                if (base + u < e1) {
                    acc01 = __builtin_elementwise_fma(v2f{w[u], w[u]}, xx[u].lo, acc01);
                    acc23 = __builtin_elementwise_fma(v2f{w[u], w[u]}, xx[u].hi, acc23);
                }
        }
    };
    auto request_rows = [&](int sr) {
        auto step = [&](auto uc) {
            constexpr int U = decltype(uc)::value;
            x[U] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(srs, (row_bcast_i<U>(sr) << 8) + gl * 16, 0, 0));
        };
        for_each_slot(step, std::make_integer_sequence<int, 16>{});
    };
    {   // prologue: the first tile synchronously, the links of the next two requested
        const int t0 = blockIdx.x, t1 = t0 + gridDim.x, t2 = t1 + gridDim.x;
        int e0, e1, sr;
        float wt;
        load_e(t0, e0, e1);
        load_e(t1, e0n, e1n);
        load_e(t2, e0n2, e1n2);
        load_ids(e0, e1, sr, wt);
        request_rows(sr);
        own = gload4(a.state_cur + (a.row_begin + row_of(t0)) * 64 + gl * 4);
        lab = (gl < IW) ? gload1(a.inv + row_of(t0) * IW + gl) : 0.0f;
        load_ids(e0n, e1n, srcn, wn);
        finish_gather(e0, e1, sr, wt);
    }
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t i0 = (int64_t)tile * 32;
        // ---- tile columns -> pieces (everything is in registers) --------------------------------------------------------------
        store_pieces4(lds + L_XP, gl * 4, node, own.x, own.y, own.z, own.w);
        store_pieces4(lds + L_XP, c_aggs + gl * 4, node, acc01.x, acc01.y, acc23.x, acc23.y);
        *reinterpret_cast<v4f *>(fl + L_OLD + node * 64 + gl * 4) = own;
        if (gl < IW) store_piece1(lds + L_XP, gl < NLc ? 64 + gl : c_aggs + 64 + (gl - NLc), node, lab);
        // ---- requests for the next tiles: rows (t+1), own / labels (t+1), ids (t+2), row pointers (t+3) ---------------------------
        const int tn = tile + gridDim.x;
        request_rows(srcn);
        own = gload4(a.state_cur + (a.row_begin + row_of(tn)) * 64 + gl * 4);
        lab = (gl < IW) ? gload1(a.inv + row_of(tn) * IW + gl) : 0.0f;
        load_ids(e0n2, e1n2, srcn2, wn2);
        load_e(tn + 2 * (int)gridDim.x, e0n3, e1n3);
        pipe_barrier();                                                           // B1: X pieces complete

        // ---- layer 0: tile jt01, K half kp; pair exchange; hidden-1 pieces ---------------------------------------------------
        {
            f32x16 acc;
            if (kp == 0) acc = bias_tile(fl + L_BIAS, jt01, half);
            else acc = f32x16{};
            dense_part<5>(lds + L_XP, kp * 5, w0, lane, acc, kp == 0 ? 5 : PIPE_C0 - 5);
            // the partner finishes the other eight registers: hand them over (kp 0 keeps 0..7, kp 1 keeps 8..15)
            float *part = fl + L_PART + wave * 1024;
#pragma unroll
            for (int i = 0; i < 8; ++i) part[i * 64 + lane] = acc[kp == 0 ? 8 + i : i];
            pipe_barrier();                                                       // B2
            const float *other = fl + L_PART + (wave ^ 4) * 1024;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = kp == 0 ? acc[i] + other[i * 64 + lane] : other[i * 64 + lane] + acc[8 + i];
            hidden_out<ACT>(lds + L_H1, 2 * jt01 + kp, lane, v);
        }
        pipe_barrier();                                                           // B3: hidden-1 pieces complete

        // ---- layer 1 ------------------------------------------------------------------------------------------------------------
        {
            f32x16 acc;
            if (kp == 0) acc = bias_tile(fl + L_BIAS + 128, jt01, half);
            else acc = f32x16{};
            dense_part<4>(lds + L_H1, kp * 4, w1, lane, acc);
            float *part = fl + L_PART + wave * 1024;
#pragma unroll
            for (int i = 0; i < 8; ++i) part[i * 64 + lane] = acc[kp == 0 ? 8 + i : i];
            pipe_barrier();                                                       // B4
            const float *other = fl + L_PART + (wave ^ 4) * 1024;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = kp == 0 ? acc[i] + other[i * 64 + lane] : other[i * 64 + lane] + acc[8 + i];
            hidden_out<ACT>(lds + L_H2, 2 * jt01 + kp, lane, v);
        }
        pipe_barrier();                                                           // B5: hidden-2 pieces complete

        // ---- layer 2: tile jt2, K quarter kq; four partials added in the order kq = 0, 1, 2, 3; last epilogue; norm partials --------
        {
            f32x16 acc;
            if (kq == 0) acc = bias_tile(fl + L_BIAS + 256, jt2, half);
            else acc = f32x16{};
            dense_part<2>(lds + L_H2, kq * 2, w2, lane, acc);
            float *part = fl + L_PART + wave * 1024;
#pragma unroll
            for (int r = 0; r < 16; ++r) part[r * 64 + lane] = acc[r];
            pipe_barrier();                                                       // B6
            // this wave finishes registers 4 kq .. 4 kq + 3 of tile jt2: features 32 jt2 + 8 kq + 4 half + (0..3)
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float s = fl[L_PART + (jt2 + 0) * 1024 + (4 * kq + i) * 64 + lane];
                s = s + fl[L_PART + (jt2 + 2) * 1024 + (4 * kq + i) * 64 + lane];
                s = s + fl[L_PART + (jt2 + 4) * 1024 + (4 * kq + i) * 64 + lane];
                s = s + fl[L_PART + (jt2 + 6) * 1024 + (4 * kq + i) * 64 + lane];
                v[i] = s;
            }
            const int f0 = 32 * jt2 + 8 * kq + 4 * half, nd = lane & 31;
            const v4f sc = *reinterpret_cast<const v4f *>(fl + L_BIAS + 320 + f0), sh = *reinterpret_cast<const v4f *>(fl + L_BIAS + 384 + f0);
            const float scv[4] = {sc.x, sc.y, sc.z, sc.w}, shv[4] = {sh.x, sh.y, sh.z, sh.w};
            const v4f o = *reinterpret_cast<const v4f *>(fl + L_OLD + nd * 64 + f0);
            const float ov[4] = {o.x, o.y, o.z, o.w};
            float d2 = 0.0f, o2 = 0.0f, nw[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float y = act_fast<ACT>(v[i]);
                if (a.bn_scale) { const float m = y * scv[i]; y = m + shv[i]; }
                nw[i] = y;
                const float d = y - ov[i];
                d2 = __builtin_fmaf(d, d, d2);
                o2 = __builtin_fmaf(ov[i], ov[i], o2);
            }
            *reinterpret_cast<v4f *>(fl + L_NEW + nd * 64 + f0) = v4f{nw[0], nw[1], nw[2], nw[3]};
            const int slot = (jt2 * 4 + kq) * 2 + half;                            // 16 partials per node, summed in slot order below
            fl[L_NORM + (nd * 16 + slot) * 2] = d2;
            fl[L_NORM + (nd * 16 + slot) * 2 + 1] = o2;
        }
        pipe_barrier();                                                           // B7: new state + norm partials complete
        if (wave == 0) {
            int moved = 0;
            if (lane < 32) {
                float d2 = 0.0f, o2 = 0.0f;
#pragma unroll
                for (int s = 0; s < 16; ++s) { d2 = d2 + fl[L_NORM + (lane * 16 + s) * 2]; o2 = o2 + fl[L_NORM + (lane * 16 + s) * 2 + 1]; }
                moved = sqrtf(d2) > a.thr * sqrtf(o2);
            }
            if (__any(moved) && lane == 0) gnn_flag_raise(a.flag_out);
        }
        {   // coalesced row stores: 32 rows x 256 B = 512 threads x 16 B
            const v4f v = *reinterpret_cast<const v4f *>(fl + L_NEW + threadIdx.x * 4);
            *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(a.state_nxt + i0 * 64 + threadIdx.x * 4)) = v;
        }
        // ---- the next tile's aggregate: its first 16 rows have been in flight through the three dense layers -------------------
        finish_gather(e0n, e1n, srcn, wn);
        e0n = e0n2; e1n = e1n2; srcn = srcn2; wn = wn2;
        e0n2 = e0n3; e1n2 = e1n3;
        pipe_barrier();                                                           // the next tile rewrites X pieces / OLD / NEW
    }
}

}   // namespace gnn_fused_dev

bool gnn_pipe_launch(int act, const GnnFusedArgs &a, unsigned grid, hipStream_t st)
{
    using namespace gnn_fused_dev;
    const size_t lds_bytes = sizeof(int) * (size_t)L_END;
#define GNN_PIPE_CASE(A)                                                                                                   \
    case A: {                                                                                                              \
        static bool raised[64] = {false};                                                                                  \
        int dev = 0;                                                                                                       \
        (void)hipGetDevice(&dev);                                                                                          \
        if (dev < 0 || dev >= 64 || !raised[dev]) {                                                                        \
            (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_pipe<A>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);   \
            if (dev >= 0 && dev < 64) raised[dev] = true;                                                                  \
        }                                                                                                                  \
        hipLaunchKernelGGL((k_pipe<A>), grid, PIPE_THREADS, lds_bytes, st, a);                                              \
        return true;                                                                                                       \
    }
    switch (act) {
        GNN_PIPE_CASE(GNN_ACT_LINEAR) GNN_PIPE_CASE(GNN_ACT_RELU) GNN_PIPE_CASE(GNN_ACT_SELU) GNN_PIPE_CASE(GNN_ACT_ELU)
        GNN_PIPE_CASE(GNN_ACT_TANH) GNN_PIPE_CASE(GNN_ACT_SIGMOID)
    default: return false;
    }
#undef GNN_PIPE_CASE
}
