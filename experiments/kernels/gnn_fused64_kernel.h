// EXPERIMENT of round 4, built and MEASURED SLOWER than k_fused (0.97 ms against 0.70 ms per launch at BASELINE size; phase stamps and the reason in
// profiles/r04_fused64_stamps.txt).  Diagnostic build only (make DIAG=1, GNN_FUSED_WIDE=1); bit-identical to k_fused's impl 2
// (tools/check_wide_tiles.py).  Kept because it is the measured answer to "64-node tiles, one wave per SIMD, the next tile's gather under the matrix phase".
//
// 64-node tiles on ONE wave per SIMD.  Included by experiments/gnn_fused_w{2,3}.hip.
//
// Same iteration as k_fused (reference GNN/GNN.py:223-242 + :202-220: CSR neighbour gather -> net_state -> convergence test), same split
// arithmetic (three exact bf16 pieces per fp32 operand, six piece products per term, the same product order per accumulator - results are
// BIT-IDENTICAL to k_fused's impl 2), state width 64, a 3- or 2-layer net_state with 128-wide hidden layers.  What differs is the shape of
// the work (round 4; probe: tools/probe/pipe64_probe.hip, profiles/r04_pipe64_probe.txt):
//   * a wave owns 64 destination nodes - two 32-node halves A / B that share every weight fragment: one 1 KiB fragment load feeds TWO
//     MFMAs, so the weight stream through the vector L1 per node halves (k_fused: 252 KiB per 32 nodes; its separator from the gather-only
//     floor, profiles/r03_gather_dma_probe.txt);
//   * 4 waves per workgroup, one per SIMD (256 + 256 registers per wave): the accumulators of two layers x two halves (256 registers) fill
//     the accumulator file, the weight ring, the operand pieces of both halves and the activation temporaries the vector file - which
//     leaves NO registers for neighbour rows in flight (a first version that staged them in VGPRs spilled 500 registers), so
//   * the gather's rows in flight live in LDS: the tile image keeps only the columns the gather writes ([labels | aggregated state |
//     aggregated labels], 84 of 148 columns) and the 16 KiB this frees per wave is a ring for ONE batch of 16 neighbour rows per lane
//     group, filled by LDS-DMA (buffer_load_dwordx4 ... lds with per-lane row addresses); the own-state columns of layer 0 come straight
//     from memory in B-operand order (requested a tile ahead, 64 registers that die in layer 0);
//   * the gather of the NEXT tile runs inside the CURRENT tile's matrix phase, by the same wave: behind layer 0 the tile image is dead, so
//     from there on a batch is in flight all the time - it is consumed (ds_read_b128, weighted into the aggregate in stored order,
//     flushed into the image at row boundaries) and the next one requested at every second weight unit of layers 1 and 2, right behind
//     that unit's weight requests (vector-memory results return in order: the fragments of the next units are then OLDER than the rows and
//     never wait for them; the wait for a batch is a counted s_waitcnt vmcnt(N) that leaves the younger weight requests in flight).  What
//     the fixed schedule has not consumed by the end of layer 2 (tiles with more arcs than average) is finished by a plain loop in front
//     of the epilogue;
//   * the epilogue re-reads the old state rows from memory (L2) for the condition and transposes the new rows through the idle ring.
// No barrier anywhere; waves never synchronise.  Tiles are handed out by the iteration's ticket counter, one tile ahead.
#pragma once
#include "../gnn_fused_kernel.h"

namespace gnn_fused_dev {

constexpr int GNN_F64_WAVES = 4;
constexpr int GNN_F64_THREADS = 64 * GNN_F64_WAVES;
constexpr int GNN_F64_IPT = 68;               // row-pointer slots per wave (65 used)
constexpr int GNN_F64_RING = 4096;            // floats: one batch = 16 DMA instructions x (4 lane groups x 256-byte row)
#define GNN_LDSP __attribute__((address_space(3)))

// s_waitcnt vmcnt(n) with n known only after unrolling (the compiler folds the switch); anything above the cap waits for a few more
__device__ __forceinline__ void wait_vmcnt(int n)
{
    switch (n < 0 ? 0 : (n > 20 ? 20 : n)) {
    case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
    case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
    case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
    case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
    case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
    case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
    case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
    case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
    case 10: asm volatile("s_waitcnt vmcnt(10)" ::: "memory"); break;
    case 11: asm volatile("s_waitcnt vmcnt(11)" ::: "memory"); break;
    case 12: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
    case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
    case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
    case 15: asm volatile("s_waitcnt vmcnt(15)" ::: "memory"); break;
    case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
    case 17: asm volatile("s_waitcnt vmcnt(17)" ::: "memory"); break;
    case 18: asm volatile("s_waitcnt vmcnt(18)" ::: "memory"); break;
    case 19: asm volatile("s_waitcnt vmcnt(19)" ::: "memory"); break;
    default: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
    }
}

// ---- gather of a 64-node tile as a resumable state machine, rows in flight in an LDS ring -----------------------------------------------
// Lane group g (16 lanes, 16 B per lane = one 256-byte state row per group and load) owns the 16 consecutive nodes 16 g .. 16 g + 15 and
// walks their contiguous CSR entries in batches of 16.  issue(): 16 LDS-DMA loads, load u bringing row u of every group's batch to
// ring[u][group][256 B] (ids held by the group's lanes, broadcast by DPP), then the ids / weights of the batch after it - always 18
// vector-memory instructions.  consume(): the fmaf chain of the batch in stored order (bit-identical to the oracle's chain), flushing to
// the tile at every row boundary.  A group past its last entry is masked off.
struct Gather64 {
    int base, e_end, node, node_end, next_end, my_src;
    float my_w, cur_w;
    v2f acc01, acc23;

    __device__ __forceinline__ void start(const int *ipt, int lane, int src0, float w0)
    {
        const int grp = lane >> 4;
        node = grp * 16;
        node_end = node + 16;
        base = ipt[node];
        e_end = ipt[node_end];
        next_end = ipt[node + 1];
        my_src = src0;
        my_w = w0;
        cur_w = 0.0f;
        acc01 = v2f{0.f, 0.f};
        acc23 = v2f{0.f, 0.f};
    }
    __device__ __forceinline__ bool more() const { return __any(base < e_end) != 0; }          // wave-uniform
    template <int... J>
    __device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, float *ring, int voff0, std::integer_sequence<int, J...>)
    {
        (__builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (GNN_LDSP void *)(ring + J * 256), 16, (row_bcast_i<J>(my_src) << 8) + voff0, 0, 0, 0), ...);
    }
    __device__ __forceinline__ void issue(const GnnFusedArgs &a, __amdgpu_buffer_rsrc_t rsrc, float *ring, int lane)
    {
        if (base < e_end) {                                  // (per lane group; inactive groups request nothing)
            dma16(rsrc, ring, (lane & 15) * 16, std::make_integer_sequence<int, 16>{});
            cur_w = my_w;
            int nb = base + 16 + (lane & 15);
            const bool ok = nb < e_end;
            nb = ok ? nb : e_end - 1;                        // clamp: a real entry (the loads are always issued: the counted waits rely on it)
            const int s_ = gload1(a.adj_src + nb);
            const float w_ = gload1(a.adj_w + nb);
            my_src = ok ? s_ : 0;
            my_w = ok ? w_ : 0.0f;
        }
    }
    // xo = image + c_aggs + 4 (lane & 15) (column chunk of this lane in the aggregated-state block); rg = ring + 64 (lane >> 4) + 4 (lane & 15)
    template <int... J>
    __device__ __forceinline__ void consume4(float *xo, const float *rg, const int *ipt, int KP, int u0, std::integer_sequence<int, J...>)
    {
        v4f x[4];
        ((x[J] = *reinterpret_cast<const v4f *>(rg + (u0 + J) * 256)), ...);
        float w[4];
        ((w[J] = 0.0f), ...);
        // (u0 is a compile-time constant at every call site: the DPP selectors below are immediates)
        if (u0 == 0) { w[0] = row_bcast_f<0>(cur_w); w[1] = row_bcast_f<1>(cur_w); w[2] = row_bcast_f<2>(cur_w); w[3] = row_bcast_f<3>(cur_w); }
        else if (u0 == 4) { w[0] = row_bcast_f<4>(cur_w); w[1] = row_bcast_f<5>(cur_w); w[2] = row_bcast_f<6>(cur_w); w[3] = row_bcast_f<7>(cur_w); }
        else if (u0 == 8) { w[0] = row_bcast_f<8>(cur_w); w[1] = row_bcast_f<9>(cur_w); w[2] = row_bcast_f<10>(cur_w); w[3] = row_bcast_f<11>(cur_w); }
        else { w[0] = row_bcast_f<12>(cur_w); w[1] = row_bcast_f<13>(cur_w); w[2] = row_bcast_f<14>(cur_w); w[3] = row_bcast_f<15>(cur_w); }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = base + u0 + q;
            if (e < e_end) {
                while (e >= next_end) {                      // row boundary (a loop: rows without arcs)
                    *reinterpret_cast<v4f *>(xo + node * KP) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};
                    acc01 = v2f{0.f, 0.f};
                    acc23 = v2f{0.f, 0.f};
                    ++node;
                    next_end = ipt[node + 1];
                }
                acc01 = __builtin_elementwise_fma(v2f{w[q], w[q]}, x[q].lo, acc01);
                acc23 = __builtin_elementwise_fma(v2f{w[q], w[q]}, x[q].hi, acc23);
            }
        }
    }
    // n_younger: vector-memory instructions issued after the batch's last DMA load (they may stay in flight)
    __device__ __forceinline__ void consume(float *xo, const float *rg, const int *ipt, int KP, int n_younger)
    {
        wait_vmcnt(n_younger);
        consume4(xo, rg, ipt, KP, 0, std::make_integer_sequence<int, 4>{});
        consume4(xo, rg, ipt, KP, 4, std::make_integer_sequence<int, 4>{});
        consume4(xo, rg, ipt, KP, 8, std::make_integer_sequence<int, 4>{});
        consume4(xo, rg, ipt, KP, 12, std::make_integer_sequence<int, 4>{});
        base += 16;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the ring is read: the next issue may overwrite it
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    // the last row with arcs, then the rows without
    __device__ __forceinline__ void finish(float *xo, int KP)
    {
        for (; node < node_end; ++node) {
            *reinterpret_cast<v4f *>(xo + node * KP) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};
            acc01 = v2f{0.f, 0.f};
            acc23 = v2f{0.f, 0.f};
        }
    }
};

// own-state rows of a 64-node tile in B-operand order (lane = (row, k half): the 8 consecutive columns 16 c + 8 half .. of chunk c = 0 .. 3,
// for the row of this lane in half A and in half B) and its label columns: requested a tile ahead
struct TileRows64 {
    float ownA[4][8], ownB[4][8];
    float lab[8];
    __device__ __forceinline__ void load(const GnnFusedArgs &a, int64_t i0, int lane)
    {
        const float *sA = a.state_cur + (a.row_begin + i0 + (lane & 31)) * 64 + 8 * (lane >> 5), *sB = sA + 32 * 64;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const v4f a0 = gload4(sA + 16 * c), a1 = gload4(sA + 16 * c + 4), b0 = gload4(sB + 16 * c), b1 = gload4(sB + 16 * c + 4);
            ownA[c][0] = a0.x; ownA[c][1] = a0.y; ownA[c][2] = a0.z; ownA[c][3] = a0.w; ownA[c][4] = a1.x; ownA[c][5] = a1.y; ownA[c][6] = a1.z; ownA[c][7] = a1.w;
            ownB[c][0] = b0.x; ownB[c][1] = b0.y; ownB[c][2] = b0.z; ownB[c][3] = b0.w; ownB[c][4] = b1.x; ownB[c][5] = b1.y; ownB[c][6] = b1.z; ownB[c][7] = b1.w;
        }
        const int nlab = 64 * a.IW;
        const float *ls = a.inv + i0 * a.IW;
#pragma unroll
        for (int u = 0; u < 8; ++u) lab[u] = (lane + 64 * u < nlab) ? gload1(ls + lane + 64 * u) : 0.0f;
    }
    // Xv: the image addressed with the FULL column numbers of k_fused's tile (Xv = image - 64: columns 0 .. 63 do not exist), stride KP
    __device__ __forceinline__ void store_labels(const GnnFusedArgs &a, float *Xv, int64_t i0, int lane, int KP, int c_aggs) const
    {
        const int IW = a.IW, nlab = 64 * IW;
        const float inv_iw = 1.0f / (float)(IW > 0 ? IW : 1);                // t / IW without an integer division: exact for t < 2^16
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int t = lane + 64 * u;
            if (t < nlab) {
                const int i = (int)(((float)t + 0.5f) * inv_iw), c = t - i * IW;
                Xv[i * KP + label_col(c, 64, a.NLc, c_aggs)] = lab[u];
            }
        }
        if (nlab > 512)                                                      // wide label blocks: the rest, plainly
            for (int t = 512 + lane; t < nlab; t += 64) {
                const int i = t / IW, c = t - i * IW;
                Xv[i * KP + label_col(c, 64, a.NLc, c_aggs)] = gload1(a.inv + i0 * IW + t);
            }
    }
};

// ---- layer 0 on both halves: chunks 0 .. 3 (own state) from registers, the rest from the LDS image; every weight fragment of a chunk
// feeds the two halves.  xrA / xrB = Xv + (row of this lane in half A / B) * KP + 8 * (lane >> 5).  Two register sets in rotation, product
// order per accumulator as layer0_split.  n_chunks >= 5.
template <int NO>
__device__ __forceinline__ void layer0_split2(const TileRows64 &rows, const float *xrA, const float *xrB, __amdgpu_buffer_rsrc_t wrs, int voff, int soff,
                                              int n_chunks, f32x16 (&accA)[NO], f32x16 (&accB)[NO], const float *bias_lds, int half)
{
    v4i wa[NO][3], wb[NO][3];
    float xaA[8], xaB[8], xbA[8], xbB[8];
    v4i paA[3], paB[3], pbA[3], pbB[3];
    WStream ws(soff);
#define GNN_W0_LOADW(W)                                                                             \
    _Pragma("unroll") for (int jt = 0; jt < NO; ++jt)                                               \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                            \
            W[jt][pc] = ws.next(wrs, voff);                                                         \
    __builtin_amdgcn_sched_barrier(0);
#define GNN_W0_LDSX(XA, XB, C)                                                                      \
    {                                                                                               \
        const v4f lo_ = *reinterpret_cast<const v4f *>(xrA + 16 * (C)), hi_ = *reinterpret_cast<const v4f *>(xrA + 16 * (C) + 4);   \
        XA[0] = lo_.x; XA[1] = lo_.y; XA[2] = lo_.z; XA[3] = lo_.w; XA[4] = hi_.x; XA[5] = hi_.y; XA[6] = hi_.z; XA[7] = hi_.w;    \
        const v4f lp_ = *reinterpret_cast<const v4f *>(xrB + 16 * (C)), hp_ = *reinterpret_cast<const v4f *>(xrB + 16 * (C) + 4);   \
        XB[0] = lp_.x; XB[1] = lp_.y; XB[2] = lp_.z; XB[3] = lp_.w; XB[4] = hp_.x; XB[5] = hp_.y; XB[6] = hp_.z; XB[7] = hp_.w;    \
    }                                                                                               \
    __builtin_amdgcn_sched_barrier(0);
    // the MFMAs of the chunk in W / (PA_, PB_): per term NO tiles x 2 halves; behind the first 2 NO of them the pieces of the NEXT chunk
    // (values XNA / XNB, already in registers) are cut, one element pair of one half per MFMA
#define GNN_W0_MFMA(W, PA_, PB_, XNA, XNB, PNA, PNB, Z)                                             \
    {                                                                                               \
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};                       \
        _Pragma("unroll") for (int term = 0; term < 6; ++term)                                      \
            _Pragma("unroll") for (int t = 0; t < NO; ++t) {                                        \
                const int mm = term * NO + t;                                                       \
                accA[t] = mfma_bf16(W[t][TA[term]], PA_[TB[term]], (Z && term == 0) ? bias_tile(bias_lds, t, half) : accA[t]);     \
                if (mm >= NO && mm < NO + 4) {                                                      \
                    const int j = mm - NO;                                                          \
                    int q0, q1, q2;                                                                 \
                    split_pair(XNA[2 * j], XNA[2 * j + 1], q0, q1, q2);                             \
                    PNA[0][j] = q0; PNA[1][j] = q1; PNA[2][j] = q2;                                 \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
                accB[t] = mfma_bf16(W[t][TA[term]], PB_[TB[term]], (Z && term == 0) ? bias_tile(bias_lds, t, half) : accB[t]);     \
                if (mm >= NO && mm < NO + 4) {                                                      \
                    const int j = mm - NO;                                                          \
                    int q0, q1, q2;                                                                 \
                    split_pair(XNB[2 * j], XNB[2 * j + 1], q0, q1, q2);                             \
                    PNB[0][j] = q0; PNB[1][j] = q1; PNB[2][j] = q2;                                 \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
            }                                                                                       \
    }
    GNN_W0_LOADW(wa)                                                          // chunk 0
    split8(rows.ownA[0], paA[0], paA[1], paA[2]);
    split8(rows.ownB[0], paB[0], paB[1], paB[2]);
    GNN_W0_LOADW(wb)                                                          // chunk 1
    GNN_W0_MFMA(wa, paA, paB, rows.ownA[1], rows.ownB[1], pbA, pbB, true)     // chunk 0 starts the accumulators from the layer's bias
    GNN_W0_LOADW(wa)                                                          // chunk 2
    GNN_W0_MFMA(wb, pbA, pbB, rows.ownA[2], rows.ownB[2], paA, paB, false)
    GNN_W0_LOADW(wb)                                                          // chunk 3
    GNN_W0_MFMA(wa, paA, paB, rows.ownA[3], rows.ownB[3], pbA, pbB, false)
    GNN_W0_LOADW(wa)                                                          // chunk 4: the first one from the image
    GNN_W0_LDSX(xaA, xaB, 4)
    GNN_W0_MFMA(wb, pbA, pbB, xaA, xaB, paA, paB, false)
    for (int c = 4; c < n_chunks; c += 2) {                                   // set A holds chunk c
        GNN_W0_LOADW(wb)
        GNN_W0_LDSX(xbA, xbB, c + 1)
        GNN_W0_MFMA(wa, paA, paB, xbA, xbB, pbA, pbB, false)
        GNN_W0_LOADW(wa)
        GNN_W0_LDSX(xaA, xaB, c + 2)
        if (c + 1 < n_chunks) GNN_W0_MFMA(wb, pbA, pbB, xaA, xaB, paA, paB, false)
    }
#undef GNN_W0_LOADW
#undef GNN_W0_LDSX
#undef GNN_W0_MFMA
}

// ---- hidden / last layer on both halves: input = the previous layer's accumulators --------------------------------------------------------
// As layer_split_from_regs (same folded SELU, same product order per accumulator), with every weight unit consumed by the two halves and
// the weight ring DEPTH units deep.  hook(u, n) runs right behind the weight requests of unit u (u counted over this layer; n = vector-memory
// instructions those requests were; u == -1: the layer's first DEPTH units): the gather events of the next tile.
template <int NI, int NO, int ACT, class Hook>
__device__ __forceinline__ void layer_split_from_regs2(f32x16 (&hA)[NI], f32x16 (&hB)[NI], const float *bias_lds, int half, f32x16 (&accA)[NO],
                                                       f32x16 (&accB)[NO], __amdgpu_buffer_rsrc_t wrs, int voff, int soff, Hook &&hook)
{
    constexpr int TPU = NO >= 2 ? 2 : 1, UPC = NO / TPU, CH = 2 * NI, U = CH * UPC;
#ifndef GNN_F64_DEPTH
#define GNN_F64_DEPTH 2
#endif
    constexpr int DEPTH = GNN_F64_DEPTH;
    constexpr int NM = 2 * 6 * NO, NTASK = 24;          // MFMAs per chunk (both halves); VALU tasks per chunk: 8 E + 8 E, then 4 S + 4 S
    v4i w[U][TPU][3];
    int bpA[2][3][4], bpB[2][3][4];
    WStream ws(soff);
#define GNN_W1_LOAD(UU)                                                                             \
    _Pragma("unroll") for (int t = 0; t < TPU; ++t)                                                 \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                            \
            w[UU][t][pc] = ws.next(wrs, voff);
#define GNN_W1_H(H, C, I) H[(C) >> 1][8 * ((C) & 1) + (I)]
#define GNN_W1_E(H, C, I)                                                                           \
    if constexpr (ACT == GNN_ACT_SELU) {                                                            \
        constexpr float AL2_ = 1.6732632423543772f * 1.44269504088896341f;                          \
        const float v_ = GNN_W1_H(H, C, I);                                                         \
        GNN_W1_H(H, C, I) = v_ > 0.0f ? v_ : __builtin_fmaf(__builtin_amdgcn_exp2f(v_), AL2_, -AL2_);  \
    } else GNN_W1_H(H, C, I) = act_fast<ACT>(GNN_W1_H(H, C, I));
#define GNN_W1_S(H, C, J, DST) split_pair(GNN_W1_H(H, C, 2 * (J)), GNN_W1_H(H, C, 2 * (J) + 1), DST[0][J], DST[1][J], DST[2][J]);
#pragma unroll
    for (int u = 0; u < DEPTH && u < U; ++u) { GNN_W1_LOAD(u) }
    hook(-1, 3 * TPU * (DEPTH < U ? DEPTH : U));
    {   // prologue: E(0), S(0), E(1) of both halves
#pragma unroll
        for (int i = 0; i < 8; ++i) { GNN_W1_E(hA, 0, i) GNN_W1_E(hB, 0, i) }
#pragma unroll
        for (int j = 0; j < 4; ++j) { GNN_W1_S(hA, 0, j, bpA[0]) GNN_W1_S(hB, 0, j, bpB[0]) }
#pragma unroll
        for (int i = 0; i < 8; ++i) { GNN_W1_E(hA, 1, i) GNN_W1_E(hB, 1, i) }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int up = 0; up < UPC; ++up) {
            const int u = c * UPC + up;
            if (u + DEPTH < U) { GNN_W1_LOAD(u + DEPTH) }
            hook(u, u + DEPTH < U ? 3 * TPU : 0);
            __builtin_amdgcn_sched_barrier(0);
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0}, TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int term = 0; term < 6; ++term) {
#pragma unroll
                for (int t = 0; t < TPU; ++t) {
#pragma unroll
                    for (int hf = 0; hf < 2; ++hf) {
                        const int *bq = hf ? bpB[c & 1][TB[term]] : bpA[c & 1][TB[term]];
                        const int ot = up * TPU + t;
                        if (hf == 0)
                            accA[ot] = mfma_bf16(w[u][t][TA[term]], v4i{bq[0], bq[1], bq[2], bq[3]}, (c == 0 && term == 0) ? bias_tile(bias_lds, ot, half) : accA[ot]);
                        else
                            accB[ot] = mfma_bf16(w[u][t][TA[term]], v4i{bq[0], bq[1], bq[2], bq[3]}, (c == 0 && term == 0) ? bias_tile(bias_lds, ot, half) : accB[ot]);
                        // VALU tasks due after MFMA number m of the chunk: [(m - 1) NTASK / NM, m NTASK / NM)
                        const int m = ((up * 6 + term) * TPU + t) * 2 + hf + 1;
                        const int k0 = (m - 1) * NTASK / NM, k1 = m * NTASK / NM;
#pragma unroll
                        for (int k = 0; k < NTASK; ++k) {
                            if (k >= k0 && k < k1) {
                                if (k < 8) { if (c + 2 < CH) { GNN_W1_E(hA, c + 2, k) } }
                                else if (k < 16) { if (c + 2 < CH) { GNN_W1_E(hB, c + 2, k - 8) } }
                                else if (k < 20) { if (c + 1 < CH) { GNN_W1_S(hA, c + 1, k - 16, bpA[(c + 1) & 1]) } }
                                else { if (c + 1 < CH) { GNN_W1_S(hB, c + 1, k - 20, bpB[(c + 1) & 1]) } }
                            }
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        }
    }
#undef GNN_W1_LOAD
#undef GNN_W1_E
#undef GNN_W1_S
#undef GNN_W1_H
}

// ---- last-layer epilogue output of one half -> condition() for the next body, coalesced row stores ---------------------------------------
// old[jt][q]: the old state of this lane's row, features 32 jt + 8 q + 4 half .. + 3 (re-read from memory: the image has no own-state
// columns); stg: staging for the transposition, [32][68] floats in the idle gather ring.  Sums and order as finish_fast64_aligned.
__device__ __forceinline__ void finish_half64(const GnnFusedArgs &a, float *stg, f32x16 (&out)[2], const v4f (&old)[2][4], int64_t i0h, int lane)
{
    const int half = lane >> 5;
    float *srow = stg + (lane & 31) * 68;
    float d2 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * jt + 8 * q + 4 * half;
            const v4f o = old[jt][q];
            const v4f nw = {out[jt][4 * q], out[jt][4 * q + 1], out[jt][4 * q + 2], out[jt][4 * q + 3]};
            *reinterpret_cast<v4f *>(srow + f0) = nw;
            const v4f d = nw - o;
            d2 = __builtin_fmaf(d.x, d.x, d2); d2 = __builtin_fmaf(d.y, d.y, d2); d2 = __builtin_fmaf(d.z, d.z, d2); d2 = __builtin_fmaf(d.w, d.w, d2);
            o2 = __builtin_fmaf(o.x, o.x, o2); o2 = __builtin_fmaf(o.y, o.y, o2); o2 = __builtin_fmaf(o.z, o.z, o2); o2 = __builtin_fmaf(o.w, o.w, o2);
        }
    d2 = d2 + shfl_f(d2, lane ^ 32);
    o2 = o2 + shfl_f(o2, lane ^ 32);
    const float root = sqrtf(d2), nrm = sqrtf(o2);
    {   // certified gate (gnn_common.h)
        const float rhs = a.thr * nrm, band = GNN_BAND_ABS * nrm + GNN_BAND_REL * rhs;
        const bool am = __any(root > rhs), ar = __any(root > rhs + band), ab = __any(__builtin_fabsf(root - rhs) <= band);
        if (lane == 0) gnn_flag_raise_certified(a.flag_out, am, ar, ab);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float *dst = a.state_nxt + i0h * 64 + lane * 4;                          // flat element 256 u + 4 lane = row 4u + lane/16
    const float *xs = stg + (lane >> 4) * 68 + (lane & 15) * 4;
    v4f v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const v4f *>(xs + 4 * u * 68);
#pragma unroll
    for (int u = 0; u < 8; ++u) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(dst) + 256 * u) = v[u];
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                   // the staging is read: the other half / the next gather may overwrite it
}

// row pointers of a 64-node tile: lane l holds indptr[i0 + l], every lane indptr[i0 + 64] in `last` (tiles past the end: zeros)
__device__ __forceinline__ int tile64_rowptr(const GnnFusedArgs &a, int tile, int lane, int &last)
{
    const int64_t i0 = (int64_t)tile * 64;
    last = 0;
    if (i0 >= a.n_rows) return 0;
    last = gload1(a.indptr + i0 + 64);
    return gload1(a.indptr + i0 + lane);
}
// ids / weights of the first batch of every lane group (group g owns rows 16 g .. 16 g + 15; ip = tile64_rowptr's value of this lane)
__device__ __forceinline__ void tile64_first_ids(const GnnFusedArgs &a, int ip, int ip_last, int lane, int &src, float &w)
{
    const int gl = lane & 15, grp = lane >> 4;
    const int e_begin = shfl_i(ip, grp * 16), e_next = shfl_i(ip, (grp * 16 + 16) & 63);
    const int e_end = grp == 3 ? ip_last : e_next;
    src = 0; w = 0.0f;
    if (e_begin + gl < e_end) { src = gload1(a.adj_src + e_begin + gl); w = gload1(a.adj_w + e_begin + gl); }
}

// LAYERS 2 or 3, hidden layers of 4 feature tiles (65 .. 128 wide), state width 64 (NTL == 2), one activation for all layers, split
// arithmetic; the row count of the launch is a multiple of 64 (host: gnn_fused.hip).  a.KP is k_fused's row stride (all 148 columns);
// the image here has KP - 64 columns per row.
template <int LAYERS, int ACT>
__global__ void __launch_bounds__(GNN_F64_THREADS, 1) k_fused64(const GnnFusedArgs a0)
{
    constexpr int NT = 4, NTL = 2;
    const GnnFusedArgs &a = a0;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!gnn_gate_open(a.gate, a.world)) return;
#ifdef GNN_DIAG      // GNN_POISON=1: NaN over the whole LDS allocation before anything is staged
    if (a.lds_floats) {
        for (int t = threadIdx.x; t < a.lds_floats; t += blockDim.x) lds[t] = __builtin_nanf("");
        __syncthreads();
    }
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = a.KP - 64, c_aggs = a.c_aggs;      // row stride of the image; columns keep k_fused's numbers (64 .. in_s)
    constexpr int WAVE_FLOATS_FIXED = GNN_F64_RING;
    float *Xp = lds + (size_t)wave * (64 * KP + WAVE_FLOATS_FIXED);
    float *Xv = Xp - 64;                              // Xv[row * KP + column], column >= 64
    float *ring = Xp + 64 * KP;
    int *ipt = reinterpret_cast<int *>(lds + (size_t)GNN_F64_WAVES * (64 * KP + WAVE_FLOATS_FIXED) + 32) + wave * GNN_F64_IPT;
    float *ep = lds + (size_t)GNN_F64_WAVES * (64 * KP + WAVE_FLOATS_FIXED) + 32 + GNN_F64_WAVES * GNN_F64_IPT;
    for (int t = threadIdx.x; t < 3 * 32 * NTL; t += blockDim.x) {
        const int which = t / (32 * NTL), f = t - which * 32 * NTL;
        ep[t] = which == 0 ? a.bias[LAYERS - 1][f] : (a.bn_scale ? (which == 1 ? a.bn_scale[f] : a.bn_shift[f]) : 0.0f);
    }
    float *hb = ep + 3 * 32 * NTL;                    // hidden-layer biases: [LAYERS - 1][32 NT], folded SELU factor as k_fused
    for (int t = threadIdx.x; t < (LAYERS - 1) * 32 * NT; t += blockDim.x)
        hb[t] = a.bias[t / (32 * NT)][t % (32 * NT)] * (ACT == GNN_ACT_SELU ? 1.44269504088896341f : 1.0f);
    __syncthreads();
    if (a.stagger > 0) {                              // start-up spread (as k_fused)
        const int rounds = (int)((((unsigned)blockIdx.x * GNN_F64_WAVES + (unsigned)wave) * 0x9E3779B1u) >> 16) % (unsigned)(a.stagger + 1);
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
    {   // once: no tile ever writes the padding columns behind the concat or the alignment hole in front of the aggregated-state block
        const int padw = a.KP - a.in_s, hole0 = 64 + a.NLc, holew = c_aggs - hole0;
        if (padw > 0) {
            RowCol rc(lane, padw);
            for (int t = lane; t < 64 * padw; t += 64, rc.next()) Xv[rc.i * KP + a.in_s + rc.c] = 0.0f;
        }
        for (int c = 0; c < holew; ++c) Xv[lane * KP + hole0 + c] = 0.0f;
    }
    const int half = lane >> 5;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.state_cur), 0, (int)a.state_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.Ws_base), 0, a.ws_bytes, 0x00020000);
    const int wv = lane * 16;
    float *xo = Xv + c_aggs + (lane & 15) * 4;
    const float *rg = ring + 64 * (lane >> 4) + 4 * (lane & 15);

    int tile = 0, next_tile = 0;
    if (lane == 0) { tile = atomicAdd(a.tile_ctr, 1); next_tile = a.single_ticket ? 0x1fffffff : atomicAdd(a.tile_ctr, 1); }
    tile = __builtin_amdgcn_readfirstlane(tile);
    next_tile = __builtin_amdgcn_readfirstlane(next_tile);
    if ((int64_t)tile * 64 >= a.n_rows) return;       // (wave-uniform; nothing below synchronises the workgroup)

    Gather64 g;
    TileRows64 rows;
    {   // the wave's first tile: its image is built here, gather and all, before the first matrix phase
        int ip_last = 0;
        const int ip = tile64_rowptr(a, tile, lane, ip_last);
        int src0 = 0;
        float w0 = 0.0f;
        tile64_first_ids(a, ip, ip_last, lane, src0, w0);
        rows.load(a, (int64_t)tile * 64, lane);
        ipt[lane] = ip;
        if (lane == 0) ipt[64] = ip_last;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        g.start(ipt, lane, src0, w0);
        bool go = g.more();
        if (go) g.issue(a, rsrc, ring, lane);
        while (go) {
            g.consume(xo, rg, ipt, KP, 0);
            go = g.more();
            if (go) g.issue(a, rsrc, ring, lane);
        }
        g.finish(xo, KP);
        rows.store_labels(a, Xv, (int64_t)tile * 64, lane, KP, c_aggs);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
    // row pointers and first ids of the NEXT tile: requested here, consumed behind layer 0
    int ipn_last = 0;
    int ipn = tile64_rowptr(a, next_tile, lane, ipn_last);
    int srcn = 0;
    float wn = 0.0f;
    tile64_first_ids(a, ipn, ipn_last, lane, srcn, wn);

  for (;;) {
    const int64_t i0 = (int64_t)tile * 64;
    // fresh, compiler-opaque copies of the pointers for every tile (see k_fused)
    GnnFusedArgs a = a0;
    asm volatile("" : "+s"(a.bias[0]), "+s"(a.bias[1]), "+s"(a.bias[2]), "+s"(a.bn_scale), "+s"(a.bn_shift));
    asm volatile("" : "+s"(a.state_cur), "+s"(a.state_nxt), "+s"(a.inv), "+s"(a.adj_src), "+s"(a.adj_w));
    const bool have_next = (int64_t)next_tile * 64 < a.n_rows;               // wave-uniform
#ifndef GNN_DIAG
#define GNN_STAMP64(slot) do { } while (0)
#else      // diagnostic build, GNN_FUSED_STAMPS=<file>: s_memtime per tile at the phase boundaries (tools/stamps64.py)
    unsigned long long *stamp = a.stamps ? a.stamps + ((size_t)tile << 3) : nullptr;
#define GNN_STAMP64(slot)                                                                    \
    do {                                                                                     \
        if (stamp) {                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
            if (lane == 0) stamp[slot] = t_;                                                 \
            __builtin_amdgcn_sched_barrier(0);                                               \
        }                                                                                    \
    } while (0)
#endif
    GNN_STAMP64(0);

    // ---- layer 0 of both halves: own-state chunks from registers, the rest from the image ---------------------------------------------
    const float *xrA = Xv + (lane & 31) * KP + 8 * half, *xrB = xrA + 32 * KP;
    f32x16 h1A[NT], h1B[NT];
    layer0_split2<NT>(rows, xrA, xrB, wrs, wv, a.ws_off[0], a.chunks0, h1A, h1B, hb, half);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    GNN_STAMP64(1);
    // ---- the image is dead: the next tile's gather starts -----------------------------------------------------------------------------
    if (have_next) {
        ipt[lane] = ipn;
        if (lane == 0) ipt[64] = ipn_last;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        g.start(ipt, lane, srcn, wn);
    }
    bool gathering = have_next && g.more();           // wave-uniform
    // Gather events, called right behind the weight requests of a unit (the fragments of the units in flight are then older than the
    // rows requested here and never wait for them).  The first unit of the first dense layer requests the first batch; after that a
    // batch is consumed and the next requested at every second unit (3-layer nets: 8 + 4 events; 2-layer nets: at every unit, 8 events).
    // `younger` counts the vector-memory instructions issued since the last DMA load: the two id loads of issue() and every weight
    // fragment requested since - the consume waits with s_waitcnt vmcnt(younger), which leaves exactly those in flight.
    int younger = 0;
    auto step = [&]() {
        if (gathering) {
            if (a.variant & 1) __builtin_amdgcn_s_setprio(3);
            g.consume(xo, rg, ipt, KP, younger);
            gathering = g.more();
            if (gathering) { g.issue(a, rsrc, ring, lane); younger = 2; }
            if (a.variant & 1) __builtin_amdgcn_s_setprio(0);
        }
    };
    auto event_first = [&](int u, int n) {            // first dense layer behind layer 0
        younger += n;
        if (u == 0) { if (gathering) { g.issue(a, rsrc, ring, lane); younger = 2; } }
        else if (u > 0 && (LAYERS == 2 || (u & 1) == 1)) step();
    };
    auto event_later = [&](int u, int n) {
        younger += n;
        if (u > 0 && (u & 1) == 1) step();
    };
    f32x16 out[2][NTL];
    if constexpr (LAYERS == 2) {
        layer_split_from_regs2<NT, NTL, ACT>(h1A, h1B, ep, half, out[0], out[1], wrs, wv, a.ws_off[1], event_first);
    } else {
        f32x16 h2A[NT], h2B[NT];
        layer_split_from_regs2<NT, NT, ACT>(h1A, h1B, hb + 32 * NT, half, h2A, h2B, wrs, wv, a.ws_off[1], event_first);
        GNN_STAMP64(2);
        layer_split_from_regs2<NT, NTL, ACT>(h2A, h2B, ep, half, out[0], out[1], wrs, wv, a.ws_off[2], event_later);
    }
    GNN_STAMP64(3);
    // ---- what the fixed schedule has not consumed (tiles with more arcs than average) -------------------------------------------------
    while (gathering) {
        g.consume(xo, rg, ipt, KP, 0);
        gathering = g.more();
        if (gathering) g.issue(a, rsrc, ring, lane);
    }
    if (have_next) g.finish(xo, KP);
    GNN_STAMP64(4);
    // ---- requests that stay in flight across the epilogue: the ticket of the tile after next, the next tile's own rows / labels, and the
    // old state of this tile's rows for the condition (L2: they were this tile's layer-0 operands) ------------------------------------------
    int next2_tile = 0x1fffffff;
    if (lane == 0 && !a0.single_ticket) next2_tile = atomicAdd(a0.tile_ctr, 1);
    v4f old[2][2][4];
    {
        const float *so = a.state_cur + (a.row_begin + i0 + (lane & 31)) * 64 + 4 * half;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int jt = 0; jt < 2; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q) old[hf][jt][q] = gload4(so + hf * 32 * 64 + 32 * jt + 8 * q);
    }
    if (have_next) rows.load(a, (int64_t)next_tile * 64, lane);
    GNN_STAMP64(5);
    // ---- epilogue of both halves: BatchNormalization, condition, row stores ---------------------------------------------------------------
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
        for (int jt = 0; jt < NTL; ++jt) {
            if (a.bn_scale) tile_epilogue<ACT, true, true, true, true>(out[hf][jt], ep, ep + 32 * NTL, ep + 64 * NTL, jt, half);
            else tile_epilogue<ACT, false, true, true, true>(out[hf][jt], ep, nullptr, nullptr, jt, half);
        }
        finish_half64(a, ring, out[hf], old[hf], i0 + 32 * hf, lane);
    }
    GNN_STAMP64(6);
    if (!have_next) break;
    rows.store_labels(a, Xv, (int64_t)next_tile * 64, lane, KP, c_aggs);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    GNN_STAMP64(7);
    tile = next_tile;
    next_tile = __builtin_amdgcn_readfirstlane(next2_tile);
    ipn = tile64_rowptr(a, next_tile, lane, ipn_last);
    tile64_first_ids(a, ipn, ipn_last, lane, srcn, wn);
  }
}

template <int LAYERS, int ACT>
inline void launch64_one(const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    static bool raised[64] = {false};   // dynamic LDS above 64 KiB has to be requested once per kernel AND device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused64<LAYERS, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) raised[dev] = true;
    }
    hipLaunchKernelGGL((k_fused64<LAYERS, ACT>), grid, GNN_F64_THREADS, lds_bytes, st, a);
}

// instantiated activations: the ones the reference's starter and the parity suites use at this size (a fully unrolled instantiation
// takes minutes to compile); anything else stays with k_fused (the host falls back when this returns false)
template <int LAYERS>
inline bool launch64_act(int act, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    switch (act) {
    case GNN_ACT_SELU: launch64_one<LAYERS, GNN_ACT_SELU>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_TANH: launch64_one<LAYERS, GNN_ACT_TANH>(a, grid, lds_bytes, st); return true;
    default: return false;
    }
}

}   // namespace gnn_fused_dev
