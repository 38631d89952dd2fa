// Fused iteration kernel for gfx950 (device code).  Included by gnn_fused_l{1,2,3}.hip, one translation unit per layer
// count so that the instantiations compile in parallel.
//
// One launch = one iteration of GNN.Loop (reference GNN/GNN.py:223-242 + :202-220):
// CSR neighbour gather -> net_state (all Dense layers + BatchNormalization) -> convergence test.
//
// Design (DESIGN.md "Fused kernel"):
//   * one wavefront owns a tile of 32 destination nodes from gather to store; the 4 waves of a workgroup never
//     synchronise with each other (no s_barrier in the kernel);
//   * gather: groups of `lpr` lanes walk the CSR rows of 64/lpr nodes at a time, 16 B per lane per neighbour row, up to
//     4 rows in flight per lane, fmaf chain in stored order (bit-identical to the oracle); source ids / weights of a row
//     are fetched coalesced by the group and broadcast with ds_bpermute;
//   * the concat [state | labels | aggregated state | aggregated labels] of the 32 nodes lives only in LDS
//     (32 x KP floats per wave, KP odd => conflict-free column reads), never in HBM;
//   * layers run on v_mfma_f32_32x32x2_f32 in the transposed form H^T = W^T . X^T: weights are the A operand (streamed
//     from L2 in a pre-packed lane order, 16 B per lane per K step for 4 feature tiles, software-pipelined), node
//     activations the B operand.  The accumulator of layer l (feature on the register, node on the lane) becomes the B
//     operand of layer l+1 after 8 v_permlane32_swap per 32x32 tile, so hidden activations never leave registers.
//     MFMA f32 evaluates the same k-ordered fmaf chain as the oracle, hence bit-identical results;
//   * epilogue: BatchNormalization, new state to LDS, per-node relative-L2 test in the oracle's summation order,
//     coalesced 256 B row stores, one slotted atomicOr per wave that still moves.
//
// Template parameters: LAYERS Dense layers; NT 32-wide feature tiles of every hidden layer; NTL tiles of the last layer;
// ACT the activation shared by all layers (gnn_activation).
#pragma once
#include <utility>

#include "gnn_common.h"
#include "gnn_fused.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace gnn_fused_dev {

// The tile loop launders its pointers through empty asm statements (see k_fused); that hides from the compiler that they
// are global-memory pointers, and it would fall back to flat_load/flat_store (which also tie up the LDS counter).  All
// memory accesses below therefore go through explicit address_space(1) pointers.
#define GNN_GLOBAL __attribute__((address_space(1)))
template <class T>
__device__ __forceinline__ const GNN_GLOBAL T *gptr(const T *p) { return (const GNN_GLOBAL T *)p; }
template <class T>
__device__ __forceinline__ GNN_GLOBAL T *gptr_w(T *p) { return (GNN_GLOBAL T *)p; }
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef int v4i_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f gload4(const float *p) { return *(const GNN_GLOBAL v4f *)p; }
// Cache policy of the new-state row stores: 0 default, 2 nt, 16 sc1, 1 sc0 (bits of the buffer instructions' aux operand).  Round 5
// (profiles/r05_ab_store_policy.txt, BASELINE size, ms per launch): default 0.681, nt 0.666, sc1 0.678, sc0 sc1 0.679, nt sc1 0.664, sc0 nt 0.661,
// all three 0.663 - every variant with nt gains 2 - 3 %: the rows written for the NEXT iteration no longer displace the table the gather of
// THIS iteration re-reads ten times.  (The once-read CSR ids / weights as nt loads: +1 %, not adopted.)
#ifndef GNN_STORE_AUX
#define GNN_STORE_AUX 2
#endif
// 16 bytes of a new-state row at base (wave-uniform) + off floats (per lane)
__device__ __forceinline__ void gstore_row4(float *base, int64_t off, v4f v)
{
#if GNN_STORE_AUX == 0
    *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(base) + off) = v;
#elif GNN_STORE_AUX == 2
    __builtin_nontemporal_store(v, reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(base) + off));
#else
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(base, 0, 0x7ffffff0, 0x00020000);
    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4i_, v), rs, (int)(off * 4), 0, GNN_STORE_AUX);
#endif
}
__device__ __forceinline__ void gstore_row1(float *p, float v)       // one float of a new-state row (exact path: lane = feature)
{
#if GNN_STORE_AUX & 2
    __builtin_nontemporal_store(v, gptr_w(p));
#else
    *gptr_w(p) = v;
#endif
}
// the once-read streams of a tile (CSR ids / weights, row pointers); as non-temporal loads they measured 1 % slower (round 5)
__device__ __forceinline__ float gstream1(const float *p) { return *(const GNN_GLOBAL float *)p; }
__device__ __forceinline__ int gstream1(const int *p) { return *(const GNN_GLOBAL int *)p; }
__device__ __forceinline__ v2f gload2(const float *p) { return *(const GNN_GLOBAL v2f *)p; }
__device__ __forceinline__ float gload1(const float *p) { return *(const GNN_GLOBAL float *)p; }
__device__ __forceinline__ int gload1(const int *p) { return *(const GNN_GLOBAL int *)p; }

// State rows that ANOTHER workgroup of the same launch has written (the persistent small-graph loop, k_small_loop): loads and
// stores that bypass this CU's L1 / write through the XCD's L2 (global_load / global_store ... sc1), the form under which the
// hand-off needs no cache fences (cdna_hip_programming.md Guideline 16, R1; MI355X_MICROARCH.md hand-off table, row 1).
template <bool COH>
__device__ __forceinline__ float sload1(const float *p)
{
    if constexpr (COH) return __hip_atomic_load(const_cast<GNN_GLOBAL float *>((const GNN_GLOBAL float *)p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else return gload1(p);
}
template <bool COH>
__device__ __forceinline__ void sstore1(float *p, float v)
{
    if constexpr (COH) __hip_atomic_store((GNN_GLOBAL float *)p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    else *(GNN_GLOBAL float *)p = v;
}

__device__ __forceinline__ float shfl_f(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int shfl_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

// lane J of every 16-lane row broadcast to the whole row: one VALU move (DPP row_newbcast), no LDS crossbar, no wait
template <int J>
__device__ __forceinline__ int row_bcast_i(int v)
{
    return __builtin_amdgcn_update_dpp(0, v, 0x150 + J, 0xf, 0xf, true);   // bound_ctrl: no "old" value to materialise
}
template <int J>
__device__ __forceinline__ float row_bcast_f(float v) { return __int_as_float(row_bcast_i<J>(__float_as_int(v))); }

// after this, for q = 0..3: registers {4q, 4q+2, 4q+1, 4q+3} hold, in that order, the k pairs (8q, 8q+1), (8q+2, 8q+3),
// (8q+4, 8q+5), (8q+6, 8q+7) of the tile: lower half-wave the even k, upper half-wave the odd k (MFMA B-operand order)
__device__ __forceinline__ void acc_to_operand(f32x16 &h)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int r0 = 4 * q + 2 * t, r1 = r0 + 1;
            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(h[r0]), __float_as_uint(h[r1]), false, false);
            h[r0] = __uint_as_float(sw[0]);
            h[r1] = __uint_as_float(sw[1]);
        }
    }
}

template <int N>
__device__ __forceinline__ void load_w(const float *p, float (&w)[N])
{
    if constexpr (N == 4) {
        const v4f t = gload4(p);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    } else if constexpr (N == 2) {
        const v2f t = gload2(p);
        w[0] = t.x; w[1] = t.y;
    } else {
        w[0] = gload1(p);
    }
}

template <int ACT>
__device__ __forceinline__ float act_t(float v)
{
    if constexpr (ACT == GNN_ACT_SELU) {
        // gnn_act(v, SELU) with the parts of gnn_expf that cannot influence the result removed: exp is only used for
        // v <= 0 (no overflow side), NaN propagates through the arithmetic, and below the underflow threshold both forms
        // give e - 1 == -1.  On every input this returns the same bits as the generic form.
        const float u = v * 1.44269504088896341f;
        const float n = __builtin_rintf(u);
        const float f = u - n;
        float p = 0.0013218672247603536f;
        p = __builtin_fmaf(p, f, 0.009671698324382305f);
        p = __builtin_fmaf(p, f, 0.05550893023610115f);
        p = __builtin_fmaf(p, f, 0.24022237956523895f);
        p = __builtin_fmaf(p, f, 0.6931468844413757f);
        p = __builtin_fmaf(p, f, 1.0f);
        const float e = __builtin_ldexpf(p, (int)n);
        const float neg = 1.6732632423543772f * (e - 1.0f);
        return 1.0507009873554805f * (v > 0.0f ? v : neg);
    } else {
        return gnn_act(v, ACT);     // ACT is a compile-time constant: the switch folds
    }
}

// act_t on a PAIR of elements with the arithmetic on packed float2 operands (v_pk_mul / add / fma_f32: two IEEE results per issue
// slot, the same bits as the scalar instructions in the same order)
template <int ACT>
__device__ __forceinline__ v2f act_t2(v2f v)
{
    if constexpr (ACT == GNN_ACT_SELU) {
        auto splat = [](float c) { return v2f{c, c}; };
        const v2f u = v * splat(1.44269504088896341f);
        const v2f n = v2f{__builtin_rintf(u.x), __builtin_rintf(u.y)};
        const v2f f = u - n;
        v2f p = __builtin_elementwise_fma(splat(0.0013218672247603536f), f, splat(0.009671698324382305f));
        p = __builtin_elementwise_fma(p, f, splat(0.05550893023610115f));
        p = __builtin_elementwise_fma(p, f, splat(0.24022237956523895f));
        p = __builtin_elementwise_fma(p, f, splat(0.6931468844413757f));
        p = __builtin_elementwise_fma(p, f, splat(1.0f));
        v2f e = v2f{__builtin_ldexpf(p.x, (int)n.x), __builtin_ldexpf(p.y, (int)n.y)};
        e = splat(1.6732632423543772f) * (e - splat(1.0f));
        return splat(1.0507009873554805f) * v2f{v.x > 0.0f ? v.x : e.x, v.y > 0.0f ? v.y : e.y};
    } else {
        return v2f{act_t<ACT>(v.x), act_t<ACT>(v.y)};
    }
}

// Activations of the split-arithmetic path (tolerance-based parity, so the hardware transcendentals are admissible):
// v_exp_f32 / v_rcp_f32 are accurate to 1 ulp; the formulas are those of gnn_act.
template <int ACT>
__device__ __forceinline__ float act_fast(float v)
{
    constexpr float LOG2E = 1.44269504088896341f, SCALE = 1.0507009873554805f, ALPHA = 1.6732632423543772f;
    if constexpr (ACT == GNN_ACT_SELU) {
        const float e = __builtin_amdgcn_exp2f(v * LOG2E);
        return v > 0.0f ? SCALE * v : __builtin_fmaf(e, SCALE * ALPHA, -(SCALE * ALPHA));
    } else if constexpr (ACT == GNN_ACT_ELU) {
        return v > 0.0f ? v : __builtin_amdgcn_exp2f(v * LOG2E) - 1.0f;
    } else if constexpr (ACT == GNN_ACT_SIGMOID) {
        return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -LOG2E));
    } else if constexpr (ACT == GNN_ACT_TANH) {
        const float t = __builtin_amdgcn_exp2f(__builtin_fabsf(v) * (-2.0f * LOG2E));
        const float q = (1.0f - t) * __builtin_amdgcn_rcpf(1.0f + t);
        return v < 0.0f ? -q : q;
    } else {
        return gnn_act(v, ACT);
    }
}

// bias + activation (+ BatchNormalization when BN) on one accumulator tile; feature of register r on this lane:
// 32 jt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
// LDS: the three vectors were staged in LDS at kernel start (last layer of the tile loop: no global round trips per tile)
// NOBIAS: the accumulator was started from the bias (split arithmetic, bias_tile)
// fmax: features >= fmax are padding of the tile (never stored): groups of registers that hold only such features are skipped
template <int ACT, bool BN, bool FAST = false, bool LDS = false, bool NOBIAS = false>
__device__ __forceinline__ void tile_epilogue(f32x16 &a, const float *bias, const float *bn_scale, const float *bn_shift,
                                              int jt, int half, int fmax = 1 << 30)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (32 * jt + 8 * q >= fmax) continue;           // (wave-uniform)
        const int f0 = 32 * jt + 8 * q + 4 * half;
        v4f b = {0.f, 0.f, 0.f, 0.f};
        if constexpr (!NOBIAS) b = LDS ? *reinterpret_cast<const v4f *>(bias + f0) : gload4(bias + f0);
        const float bb[4] = {b.x, b.y, b.z, b.w};
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (BN) {
            const v4f s4 = LDS ? *reinterpret_cast<const v4f *>(bn_scale + f0) : gload4(bn_scale + f0);
            const v4f h4 = LDS ? *reinterpret_cast<const v4f *>(bn_shift + f0) : gload4(bn_shift + f0);
            sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
            sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
        }
        if constexpr (FAST) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                float v = NOBIAS ? a[4 * q + t] : a[4 * q + t] + bb[t];
                v = act_fast<ACT>(v);
                if (BN) { const float m = v * sc[t]; v = m + sh[t]; }
                a[4 * q + t] = v;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 4; t += 2) {             // pairs: packed arithmetic, the same bits
                v2f v = v2f{a[4 * q + t], a[4 * q + t + 1]};
                if constexpr (!NOBIAS) v = v + v2f{bb[t], bb[t + 1]};
                v = act_t2<ACT>(v);
                if (BN) { const v2f m = v * v2f{sc[t], sc[t + 1]}; v = m + v2f{sh[t], sh[t + 1]}; }
                a[4 * q + t] = v.x;
                a[4 * q + t + 1] = v.y;
            }
        }
    }
}

// Dense layer whose input is the LDS tile X (layer 0).  KK (multiple of 12) K-steps in groups of 4 with three register
// sets: the weight loads / LDS reads of group g+2 are issued before the MFMAs of group g, i.e. 8 K-steps (2048 matrix
// cycles) ahead.  Reads past the end land in the zero slack of the packed image / LDS (see gnn_fused.hip) and are unused.
template <int NO>
__device__ __forceinline__ void layer_from_lds(const float *xb, const float *wp, int kk_total, f32x16 (&acc)[NO], int wstride)
{
    constexpr int PF = 4;
    float wa[PF][NO], wb[PF][NO], wc[PF][NO], ba[PF], bb[PF], bc[PF];
#define GNN_L0_LOAD(W, B, K0)                                                                       \
    _Pragma("unroll") for (int u = 0; u < PF; ++u) {                                                \
        load_w<NO>(wp + (size_t)((K0) + u) * 64 * NO * wstride, W[u]);                                        \
        B[u] = xb[2 * ((K0) + u)];                                                                  \
    }                                                                                               \
    __builtin_amdgcn_sched_barrier(0);
#define GNN_L0_MFMA(W, B)                                                                           \
    _Pragma("unroll") for (int u = 0; u < PF; ++u)                                                  \
        _Pragma("unroll") for (int jt = 0; jt < NO; ++jt)                                           \
            acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(W[u][jt], B[u], acc[jt], 0, 0, 0);       \
    __builtin_amdgcn_sched_barrier(0);
    GNN_L0_LOAD(wa, ba, 0)
    GNN_L0_LOAD(wb, bb, PF)
    for (int kk = 0; kk < kk_total; kk += 3 * PF) {
        GNN_L0_LOAD(wc, bc, kk + 2 * PF)
        GNN_L0_MFMA(wa, ba)
        GNN_L0_LOAD(wa, ba, kk + 3 * PF)
        GNN_L0_MFMA(wb, bb)
        GNN_L0_LOAD(wb, bb, kk + 4 * PF)
        GNN_L0_MFMA(wc, bc)
    }
#undef GNN_L0_LOAD
#undef GNN_L0_MFMA
}

// Activation of a PAIR of accumulator elements, split into eight stages so that it can be issued in the shadow of
// consecutive MFMAs (the wave issues in order, so VALU work has to sit BETWEEN the MFMAs to overlap with anything).  The
// arithmetic runs on packed float2 operands (v_pk_mul/add/fma_f32: two IEEE results per issue slot, the same bits as the
// scalar instructions).  The concatenation of the stages is exactly act_t<ACT>(v + bias) on both elements.
template <int ACT>
struct ActPipe2 {
    v2f t, u, n, f, p, e;
    static __device__ __forceinline__ v2f splat(float c) { return v2f{c, c}; }
    __device__ __forceinline__ void stage(int st, float v0, float v1, float b0, float b1, float &o0, float &o1)
    {
        if constexpr (ACT == GNN_ACT_SELU) {
            switch (st) {
            case 0: t = v2f{v0, v1} + v2f{b0, b1}; u = t * splat(1.44269504088896341f); break;
            case 1: n = v2f{__builtin_rintf(u.x), __builtin_rintf(u.y)}; f = u - n; break;
            case 2: p = __builtin_elementwise_fma(splat(0.0013218672247603536f), f, splat(0.009671698324382305f));
                    p = __builtin_elementwise_fma(p, f, splat(0.05550893023610115f)); break;
            case 3: p = __builtin_elementwise_fma(p, f, splat(0.24022237956523895f));
                    p = __builtin_elementwise_fma(p, f, splat(0.6931468844413757f)); break;
            case 4: p = __builtin_elementwise_fma(p, f, splat(1.0f)); break;
            case 5: e = v2f{__builtin_ldexpf(p.x, (int)n.x), __builtin_ldexpf(p.y, (int)n.y)}; e = e - splat(1.0f); break;
            case 6: e = e * splat(1.6732632423543772f); break;
            case 7: {
                const v2f r = v2f{t.x > 0.0f ? t.x : e.x, t.y > 0.0f ? t.y : e.y} * splat(1.0507009873554805f);
                o0 = r.x; o1 = r.y;
            } break;
            }
        } else {
            if (st == 0) t = v2f{v0, v1} + v2f{b0, b1};
            if (st == 7) { o0 = gnn_act(t.x, ACT); o1 = gnn_act(t.y, ACT); }
        }
    }
};

// Dense layer whose input is the RAW accumulator of the previous layer (hin).  The previous layer's epilogue (bias +
// activation, then the half-wave exchange that turns an accumulator tile into B operands) is applied here and
// software-pipelined against this layer's MFMAs in plain program order: K-step ss of input tile ti issues NO MFMAs, and
// BETWEEN them (the wave issues in order, so that is the only place where VALU work overlaps the matrix pipe) sit the
// stages of the epilogue of element ss of input tile ti + 1 (ActPipe; on odd ss also the exchange of the pair).  Only the
// epilogue of tile 0 is exposed.  16 NI K-steps, fully unrolled; weight loads are issued DEPTH steps ahead of their MFMAs; a sched_barrier after
// every step pins this order.
template <int NI, int NO, int ACT>
__device__ __forceinline__ void layer_from_regs(f32x16 (&hin)[NI], const float *bias_prev, int half, f32x16 (&acc)[NO],
                                                const float *wp, int wstride)
{
    constexpr int STEPS = 16 * NI, DEPTH = 8;
    float w[STEPS][NO];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load_w<NO>(wp + (size_t)s * 64 * NO * wstride, w[s]);
    tile_epilogue<ACT, false>(hin[0], bias_prev, nullptr, nullptr, 0, half);
    acc_to_operand(hin[0]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ti = 0; ti < NI; ++ti) {
        float nb[16];                                   // biases of the 16 features this lane holds of tile ti + 1
        if (ti + 1 < NI) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f b4 = gload4(bias_prev + 32 * (ti + 1) + 8 * q + 4 * half);
                nb[4 * q] = b4.x; nb[4 * q + 1] = b4.y; nb[4 * q + 2] = b4.z; nb[4 * q + 3] = b4.w;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        ActPipe2<ACT> ap;
#pragma unroll
        for (int ss = 0; ss < 16; ++ss) {
            const int s = 16 * ti + ss;
            if (s + DEPTH < STEPS) load_w<NO>(wp + (size_t)(s + DEPTH) * 64 * NO * wstride, w[s + DEPTH]);
            const int reg = 4 * (ss >> 2) + ((ss & 3) == 1 ? 2 : (ss & 3) == 2 ? 1 : (ss & 3));
            const float b = hin[ti][reg];
            const bool epi = ti + 1 < NI;
            const int pe = ss & ~1;                                  // the pair (pe, pe + 1) spans steps pe and pe + 1
#pragma unroll
            for (int jt = 0; jt < NO; ++jt) {
                acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][jt], b, acc[jt], 0, 0, 0);
                if (epi) {      // stages of the epilogue of elements (pe, pe + 1) of tile ti + 1, in the shadow of this MFMA
                    constexpr int SPM = 4 / NO;                      // stages per MFMA: NO = 4 -> 1, 2 -> 2, 1 -> 4
#pragma unroll
                    for (int q = jt * SPM; q < (jt + 1) * SPM; ++q) {
                        const int st = (ss & 1) * 4 + q;             // even step: stages 0..3, odd step: stages 4..7
                        float o0 = 0.0f, o1 = 0.0f;
                        ap.stage(st, hin[ti + 1][pe], hin[ti + 1][pe + 1], nb[pe], nb[pe + 1], o0, o1);
                        if (st == 7) {
                            auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(o0), __float_as_uint(o1), false, false);
                            hin[ti + 1][pe] = __uint_as_float(sw[0]);
                            hin[ti + 1][pe + 1] = __uint_as_float(sw[1]);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// SPLIT arithmetic (impl 2): fp32 operands as three bf16 pieces on v_mfma_f32_32x32x16_bf16.
//
// The f32 MFMA of gfx950 issues at the f32 VALU rate and shares its pipe; the bf16 MFMA is 13x faster per K and overlaps
// the partner wave's VALU work.  Every fp32 value v is cut, by truncation, into p0 + p1 + p2 == v EXACTLY (8 + 8 + 8
// mantissa bits; the remainders v - p0 and v - p0 - p1 are exact in fp32), products of pieces are exact in the fp32
// accumulator, and of the nine piece products of x * w the six of relative weight >= 2^-16 are accumulated
// (p0q0, p0q1, p1q0, p1q1, p0q2, p2q0); the dropped ones are <= 3 * 2^-24 |x w|, the size of one fp32 rounding.  What
// differs from the oracle's fmaf chain is therefore only the accumulation order inside the matrix unit: results agree
// to fp32 rounding noise (tests: 1e-5 against the float64 oracle), not bit for bit.  The exact f32-MFMA path stays
// available as impl 1.
//
// Operand layout of the 32x32x16 MFMA: lane l carries 8 consecutive k of row / column l & 31, k group l >> 5.  Which k
// sits where is free as long as weights (A) and activations (B) agree, so the k order of a hidden layer is simply the
// order in which the previous layer's accumulator already holds the features (no cross-lane exchange at all):
//   layer 0, chunk c:                  k(h, i) = 16 c + 8 h + i                      (LDS tile column)
//   hidden, chunk c = 2 ti + q:        k(h, i) = 32 ti + (r & 3) + 8 (r >> 2) + 4 h,  r = 8 q + i   (accumulator register r)
// gnn_fused.hip packs the weight pieces in the same order: [chunk][out tile][piece][lane][8 bf16].
// ---------------------------------------------------------------------------------------------------------------------
typedef int v4i __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ v4i gload4i(const int *p) { return *(const GNN_GLOBAL v4i *)p; }

// weight pieces through the image's buffer descriptor: vector offset = 16 * lane (the same register for every load), everything
// else in the scalar offset, so an unrolled layer issues no vector address arithmetic at all
__device__ __forceinline__ v4i bload4i(__amdgpu_buffer_rsrc_t r, int voff, int soff)
{
    return __builtin_bit_cast(v4i, __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0));
}
// Running scalar offset of a weight stream that is read strictly in image order, 1 KiB per load: four loads share one
// value of the offset (immediate offsets 0 / 1 / 2 / 3 KiB), then ONE scalar add moves it on.  The add is opaque to the compiler
// on purpose: as plain constants the ~200 offsets of an unrolled layer are hoisted out of the tile loop, spilled and fetched
// back with a v_readlane each.
struct WStream {
    int soff, n;
    __device__ __forceinline__ WStream(int start) : soff(start), n(0) {}
    __device__ __forceinline__ v4i next(__amdgpu_buffer_rsrc_t r, int voff)
    {
        // (s_add_u32 writes SCC: without the clobber the compiler kept a loop's s_cmp result live across this statement and the loop of
        // layer0_split16 ran once - found in round 3; the 32-node layers happened to have no SCC value live here)
        if (n == 4) { asm volatile("s_add_u32 %0, %0, 0x1000" : "+s"(soff) : : "scc"); n = 0; }
        return bload4i(r, voff + 1024 * n++, soff);
    }
};

// bias of the 16 features this lane holds of output tile jt, laid out as an accumulator tile: the split-arithmetic layers START
// their accumulators from the bias (4 LDS reads per tile) instead of adding it to every element afterwards (16 VALU adds)
__device__ __forceinline__ f32x16 bias_tile(const float *bias_lds, int jt, int half)
{
    f32x16 t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const v4f b = *reinterpret_cast<const v4f *>(bias_lds + 32 * jt + 8 * q + 4 * half);
        t[4 * q] = b.x; t[4 * q + 1] = b.y; t[4 * q + 2] = b.z; t[4 * q + 3] = b.w;
    }
    return t;
}

__device__ __forceinline__ void split8(const float (&v)[8], v4i &p0, v4i &p1, v4i &p2)
{
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const unsigned a = __float_as_uint(v[2 * j]), b = __float_as_uint(v[2 * j + 1]);
        const float ar = v[2 * j] - __uint_as_float(a & 0xffff0000u), br = v[2 * j + 1] - __uint_as_float(b & 0xffff0000u);
        const unsigned aru = __float_as_uint(ar), bru = __float_as_uint(br);
        const float ar2 = ar - __uint_as_float(aru & 0xffff0000u), br2 = br - __uint_as_float(bru & 0xffff0000u);
        // high halves of (even, odd) element into the (low, high) half of one dword
        p0[j] = (int)__builtin_amdgcn_perm(b, a, 0x07060302u);
        p1[j] = (int)__builtin_amdgcn_perm(bru, aru, 0x07060302u);
        p2[j] = (int)__builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302u);
    }
}

__device__ __forceinline__ f32x16 mfma_bf16(v4i a, v4i b, f32x16 c)
{
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}

// one (even, odd) element pair of split8: pieces of v0, v1 into dword j of the three operands
__device__ __forceinline__ void split_pair(float v0, float v1, int &p0, int &p1, int &p2)
{
    const unsigned a = __float_as_uint(v0), b = __float_as_uint(v1);
    const float ar = v0 - __uint_as_float(a & 0xffff0000u), br = v1 - __uint_as_float(b & 0xffff0000u);
    const unsigned aru = __float_as_uint(ar), bru = __float_as_uint(br);
    const float ar2 = ar - __uint_as_float(aru & 0xffff0000u), br2 = br - __uint_as_float(bru & 0xffff0000u);
    p0 = (int)__builtin_amdgcn_perm(b, a, 0x07060302u);
    p1 = (int)__builtin_amdgcn_perm(bru, aru, 0x07060302u);
    p2 = (int)__builtin_amdgcn_perm(__float_as_uint(br2), __float_as_uint(ar2), 0x07060302u);
}

// the six piece products of one K = 16 chunk on T output tiles; consecutive MFMAs go to different accumulators
template <int T>
__device__ __forceinline__ void mfma_split(const v4i (&w)[T][3], v4i b0, v4i b1, v4i b2, f32x16 *acc)
{
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = mfma_bf16(w[t][0], b2, acc[t]);      // smallest terms first
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = mfma_bf16(w[t][2], b0, acc[t]);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = mfma_bf16(w[t][1], b1, acc[t]);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = mfma_bf16(w[t][0], b1, acc[t]);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = mfma_bf16(w[t][1], b0, acc[t]);
#pragma unroll
    for (int t = 0; t < T; ++t) acc[t] = mfma_bf16(w[t][0], b0, acc[t]);
}

// layer 0: input = LDS tile.  xr = X + (lane & 31) * KP + 8 * (lane >> 5); wl = split image of the layer + 4 * lane.
// Two register sets: the weights / tile values of chunk c + 1 are requested before the MFMAs of chunk c.  The image has
// two zero chunks of slack and the LDS allocation 128 B, so the look-ahead never leaves them; it is never consumed.
template <int NO, bool AL16>
__device__ __forceinline__ void layer0_split(const float *xr, __amdgpu_buffer_rsrc_t wrs, int voff, int soff, int n_chunks, f32x16 (&acc)[NO],
                                             const float *bias_lds, int half)
{
    v4i wa[NO][3], wb[NO][3];
    float xa[8], xb[8];
    v4i pa[3], pb[3];                                  // operand pieces of the chunk whose weights sit in wa / wb
    WStream ws(soff);                                  // chunks are requested in ascending order: 0, 1, 2, ...
#define GNN_S0_LOAD(W, XV, C)                                                                       \
    _Pragma("unroll") for (int jt = 0; jt < NO; ++jt)                                               \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc)                                            \
            W[jt][pc] = ws.next(wrs, voff);                                                         \
    if constexpr (AL16) {                                                                           \
        const v4f lo_ = *reinterpret_cast<const v4f *>(xr + 16 * (C)), hi_ = *reinterpret_cast<const v4f *>(xr + 16 * (C) + 4);   \
        XV[0] = lo_.x; XV[1] = lo_.y; XV[2] = lo_.z; XV[3] = lo_.w; XV[4] = hi_.x; XV[5] = hi_.y; XV[6] = hi_.z; XV[7] = hi_.w;    \
    } else {                                                                                        \
        _Pragma("unroll") for (int i = 0; i < 8; ++i) XV[i] = xr[16 * (C) + i];                     \
    }                                                                                               \
    __builtin_amdgcn_sched_barrier(0);
    // the six piece products of the chunk in W / P; after the first NO MFMAs the pieces of the NEXT chunk (values XN, already
    // in registers) are cut in the shadow of the matrix pipe, one element pair per following MFMA
#define GNN_S0_MFMA(W, P, XN, PN, Z)                                                                \
    {                                                                                               \
        constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};                       \
        _Pragma("unroll") for (int term = 0; term < 6; ++term)                                      \
            _Pragma("unroll") for (int t = 0; t < NO; ++t) {                                        \
                acc[t] = mfma_bf16(W[t][PA[term]], P[PB[term]], (Z && term == 0) ? bias_tile(bias_lds, t, half) : acc[t]);       \
                const int m = term * NO + t;                                                        \
                if (m >= NO && m < NO + 4) {                                                        \
                    const int j = m - NO;                                                           \
                    int q0, q1, q2;                                                                 \
                    split_pair(XN[2 * j], XN[2 * j + 1], q0, q1, q2);                               \
                    PN[0][j] = q0; PN[1][j] = q1; PN[2][j] = q2;                                    \
                }                                                                                   \
                __builtin_amdgcn_sched_barrier(0);                                                  \
            }                                                                                       \
    }
    GNN_S0_LOAD(wa, xa, 0)
    split8(xa, pa[0], pa[1], pa[2]);
    GNN_S0_LOAD(wb, xb, 1)
    GNN_S0_MFMA(wa, pa, xb, pb, true)                  // chunk 0 starts the accumulators from the layer's bias
    for (int c = 1; c < n_chunks; c += 2) {
        GNN_S0_LOAD(wa, xa, c + 1)
        GNN_S0_MFMA(wb, pb, xa, pa, false)
        GNN_S0_LOAD(wb, xb, c + 2)
        if (c + 1 < n_chunks) GNN_S0_MFMA(wa, pa, xb, pb, false)
    }
#undef GNN_S0_LOAD
#undef GNN_S0_MFMA
}

// hidden / last layer: input = accumulator tiles of the previous layer; its epilogue (bias + activation; bias_prev points to the
// copy staged in LDS at kernel start) is applied here,
// software-pipelined in program order against this layer's MFMAs (the wave issues in order; VALU work placed right after
// an MFMA runs while the matrix pipe executes it): while the 6 NO MFMAs of chunk c are issued, the 8 elements of chunk
// c + 2 get bias + activation (E) and the elements of chunk c + 1 are cut into bf16 pieces (S), one task per few MFMAs.
// Units of (chunk, pair of output tiles), fully unrolled, weights requested DEPTH units ahead.
template <int NI, int NO, int ACT>
__device__ __forceinline__ void layer_split_from_regs(f32x16 (&hin)[NI], const float *bias_lds, int half, f32x16 (&acc)[NO],
                                                      __amdgpu_buffer_rsrc_t wrs, int voff, int soff)
{
    constexpr int TPU = NO >= 2 ? 2 : 1, UPC = NO / TPU, CH = 2 * NI, U = CH * UPC;
    constexpr int DEPTH = (NI + NO >= 8) ? 1 : 3;       // 24 VGPRs per unit in flight next to 16 (NI + NO) of activations (128 -> 128: depth 2 spills ~40 VGPRs and is slower)
    constexpr int NM = 6 * NO, NTASK = 12;              // MFMAs per chunk; VALU tasks per chunk: 8 E elements, then 4 S pairs (late:
                                                        // the pieces of chunk c + 1 become live when b2 / b1 of chunk c are dead)
    v4i w[U][TPU][3];
    int bp[2][3][4];                                    // operand pieces of chunk c (bp[c & 1]) and c + 1
    WStream ws(soff);                                   // units are requested in ascending order = image order
#define GNN_S1_LOAD(UU)                                                                             \
    _Pragma("unroll") for (int t = 0; t < TPU; ++t)                                                 \
        _Pragma("unroll") for (int pc = 0; pc < 3; ++pc) w[UU][t][pc] = ws.next(wrs, voff);
    // (the previous layer's accumulators already contain its bias: bias_tile)
#define GNN_S1_H(C, I) hin[(C) >> 1][8 * ((C) & 1) + (I)]
    // SELU between dense layers, folded (gnn_fused_pack scales the split image to match): the accumulator holds v' = log2(e) v, the
    // operand handed to the next layer is x' = v' (v > 0) or log2(e) alpha (2^v' - 1), and the next layer's weights carry the factor
    // scale / log2(e).  Four instructions per element (exp2, compare, fma, select) instead of six.
#define GNN_S1_E(C, I)                                                                              \
    if constexpr (ACT == GNN_ACT_SELU) {                                                            \
        constexpr float AL2_ = 1.6732632423543772f * 1.44269504088896341f;                          \
        const float v_ = GNN_S1_H(C, I);                                                            \
        GNN_S1_H(C, I) = v_ > 0.0f ? v_ : __builtin_fmaf(__builtin_amdgcn_exp2f(v_), AL2_, -AL2_);  \
    } else GNN_S1_H(C, I) = act_fast<ACT>(GNN_S1_H(C, I));
#define GNN_S1_S(C, J, DST) split_pair(GNN_S1_H(C, 2 * (J)), GNN_S1_H(C, 2 * (J) + 1), DST[0][J], DST[1][J], DST[2][J]);
    // (Halving the tasks - one half per MFMA gap instead of a whole task after every second MFMA - was measured: 5 % slower.)
#pragma unroll
    for (int u = 0; u < DEPTH && u < U; ++u) { GNN_S1_LOAD(u) }
    {   // prologue: E(0), S(0), E(1)
#pragma unroll
        for (int i = 0; i < 8; ++i) { GNN_S1_E(0, i) }
#pragma unroll
        for (int j = 0; j < 4; ++j) { GNN_S1_S(0, j, bp[0]) }
#pragma unroll
        for (int i = 0; i < 8; ++i) { GNN_S1_E(1, i) }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int c = 0; c < CH; ++c) {
#pragma unroll
        for (int up = 0; up < UPC; ++up) {
            const int u = c * UPC + up;
            if (u + DEPTH < U) { GNN_S1_LOAD(u + DEPTH) }
            // term order as mfma_split: smallest products first
            constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int term = 0; term < 6; ++term) {
#pragma unroll
                for (int t = 0; t < TPU; ++t) {
                    const int *bq = bp[c & 1][PB[term]];
                    // the first MFMA of an accumulator takes the layer's bias as C (no zeroed register tile, no bias add later)
                    acc[up * TPU + t] = mfma_bf16(w[u][t][PA[term]], v4i{bq[0], bq[1], bq[2], bq[3]},
                                                  (c == 0 && term == 0) ? bias_tile(bias_lds, up * TPU + t, half) : acc[up * TPU + t]);
                    // VALU tasks due after MFMA number m of the chunk: [(m - 1) NTASK / NM, m NTASK / NM).  (Grouping the tasks of 2 or 4
                    // consecutive MFMAs behind the last of them was measured in round 4: no effect, DESIGN.md appendix.)
                    const int m = (up * 6 + term) * TPU + t + 1;
                    const int k0 = (m - 1) * NTASK / NM, k1 = m * NTASK / NM;
#pragma unroll
                    for (int k = 0; k < NTASK; ++k) {
                        if (k >= k0 && k < k1) {
                            if (k < 8) { if (c + 2 < CH) { GNN_S1_E(c + 2, k) } }
                            else { if (c + 1 < CH) { GNN_S1_S(c + 1, k - 8, bp[(c + 1) & 1]) } }
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
    }
#undef GNN_S1_LOAD
#undef GNN_S1_E
#undef GNN_S1_S
#undef GNN_S1_H
}

// ---------------------------------------------------------------------------------------------------------------------
// tile load + gather.  X[32][KP] columns: [state | nodes? | aggregated state | aggregated nodes? | aggregated arcs | 0 pad]
// ---------------------------------------------------------------------------------------------------------------------

// label columns [nodes | agg nodes | agg arcs] of inv[n_rows][IW] -> columns Ds.., 2Ds+NLc.., 2Ds+2NLc..
__device__ __forceinline__ int label_col(int c, int Ds, int NLc, int c_aggs)
{
    return c < NLc ? Ds + c : c_aggs + Ds + (c - NLc);      // [nodes] behind the state, [agg nodes | agg arcs] behind the aggregated state
}

// (row, column) of the flat index lane, lane + 64, lane + 128, ... of a [rows, width] block: one division per tile instead
// of one per element (the generic paths are latency chains of small graphs; an integer division is ~40 instructions)
struct RowCol {
    int i, c, qi, qc, width;
    __device__ __forceinline__ RowCol(int lane, int width_) : width(width_)
    {
        i = lane / width; c = lane - i * width;
        qi = 64 / width; qc = 64 - qi * width;
    }
    __device__ __forceinline__ void next()
    {
        i += qi; c += qc;
        if (c >= width) { c -= width; ++i; }
    }
};

__device__ __forceinline__ void zero_pad_columns(const GnnFusedArgs &a, float *X, int lane, int KP)
{
    const int padw = KP - a.in_s;
    RowCol rc(lane, padw);
    for (int t = lane; t < 32 * padw; t += 64, rc.next()) X[rc.i * KP + a.in_s + rc.c] = 0.0f;
    const int hole0 = a.Ds + a.NLc, holew = a.c_aggs - hole0;      // alignment hole in front of the aggregated-state block (0 - 3 columns)
    if (holew > 0 && lane < 32)
        for (int c = 0; c < holew; ++c) X[lane * KP + hole0 + c] = 0.0f;
}

// Generic shapes (any Ds, partial tiles).  Correct for everything, tuned for nothing: small graphs are launch-bound.
// COH: the state rows are read with L1-bypassing loads (see sload1).
// own_from_lds (k_small_loop, bodies > 0): the tile is still in LDS from the previous body - zero padding and label columns are
// unchanged, the new own state sits in columns c_aggs.. and only has to move to columns 0..
// RND: entries per round of the narrow gather (ids / weights, then rows of RND entries requested together)
// ecache_src / ecache_w (k_small_loop): the tile's arc ids / weights, entries [ecache_base, ...), already in LDS - the narrow gather then
// pays one memory round trip per round (the rows) instead of two (ids, then rows)
template <bool COH = false, int RND = 4>
__device__ __forceinline__ void load_tile_generic(const GnnFusedArgs &a, float *X, const int *ipt, int64_t i0, int lane,
                                               int nvalid, int KP, int c_aggs, bool own_from_lds = false, const int *ecache_src = nullptr,
                                               const float *ecache_w = nullptr, int ecache_base = 0, bool skip_gather = false)
{
    const int Ds = a.Ds, NLc = a.NLc;
    if (!own_from_lds) {
        zero_pad_columns(a, X, lane, KP);
        if (nvalid < 32) {
            RowCol rc(lane, a.in_s);
            for (int t = lane; t < (32 - nvalid) * a.in_s; t += 64, rc.next()) X[(nvalid + rc.i) * KP + rc.c] = 0.0f;
        }
    }
    if (own_from_lds) {
        const int total = nvalid * Ds;
        RowCol rc(lane, Ds);
        for (int t = lane; t < total; t += 64, rc.next()) X[rc.i * KP + rc.c] = X[rc.i * KP + c_aggs + rc.c];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // before the aggregated state overwrites those columns
    } else {   // own state rows (contiguous in HBM) into columns [0, Ds)
        const float *src = a.state_cur + (a.row_begin + i0) * Ds;
        const int total = nvalid * Ds;
        RowCol rc(lane, Ds);
        for (int t = lane; t < total; t += 64, rc.next()) X[rc.i * KP + rc.c] = sload1<COH>(src + t);
    }
    if (a.IW > 0 && !own_from_lds) {
        const float *src = a.inv + i0 * a.IW;
        const int total = nvalid * a.IW;
        RowCol rc(lane, a.IW);
        for (int t = lane; t < total; t += 64, rc.next()) X[rc.i * KP + label_col(rc.c, Ds, NLc, c_aggs)] = gload1(src + t);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (skip_gather) return;       // k_small_loop: the aggregate comes from the padded exchange rows (small_gather_padded)
    if (a.agg_in) {        // feature-sliced exchange: the aggregate of the owned rows was computed outside (contiguous rows, no gather)
        const float *src = a.agg_in + i0 * Ds;
        const int total = nvalid * Ds;
        RowCol rc(lane, Ds);
        for (int t = lane; t < total; t += 64, rc.next()) X[rc.i * KP + c_aggs + rc.c] = gload1(src + t);
        return;
    }
    if (Ds <= 32) {
        // narrow rows (small label-sized states): lane = (node, column half); all 32 nodes walk their entries at once, so
        // the latency chain is max-degree long instead of 32 / groups passes long
        const int node = lane & 31, hf = lane >> 5;
        const int cbeg = hf ? (Ds + 1) / 2 : 0, cend = hf ? Ds : (Ds + 1) / 2;
        const int beg = ipt[node], end = ipt[node + 1];
        float acc[16];
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = 0.0f;
        // RND entries per round: their ids / weights are requested together, then their rows, then the fmaf chain in stored
        // order - two memory latencies per round instead of two per entry
        for (int e = beg; e < end; e += RND) {
            float w[RND];
            const float *xp[RND];
#pragma unroll
            for (int u = 0; u < RND; ++u) {
                const int ee = e + u < end ? e + u : e;        // clamp: a real entry, result unused
                w[u] = ecache_w ? ecache_w[ee - ecache_base] : gload1(a.adj_w + ee);
                xp[u] = a.state_cur + (int64_t)(ecache_src ? ecache_src[ee - ecache_base] : gload1(a.adj_src + ee)) * Ds + cbeg;
            }
            float x[RND][16];
#pragma unroll
            for (int u = 0; u < RND; ++u)
#pragma unroll
                for (int c = 0; c < 16; ++c) x[u][c] = (cbeg + c < cend) ? sload1<COH>(xp[u] + c) : 0.0f;
#pragma unroll
            for (int u = 0; u < RND; ++u)
                if (e + u < end) {
#pragma unroll
                    for (int c = 0; c < 16; ++c)
                        if (cbeg + c < cend) acc[c] = __builtin_fmaf(w[u], x[u][c], acc[c]);
                }
        }
        if (node < nvalid) {
#pragma unroll
            for (int c = 0; c < 16; ++c)
                if (cbeg + c < cend) X[node * KP + c_aggs + cbeg + c] = acc[c];
        }
        return;
    }
    // gather: groups of lpr lanes, one node per group per pass, one column chunk per lane (lpr * vec >= Ds); the entries
    // of a row are consumed four at a time (ids, weights and rows of the four requested before the first fmaf)
    const int lpr = a.lpr, gl = lane & (lpr - 1), grp = lane >> a.lpr_log2, groups = 64 >> a.lpr_log2;
    const bool colok = gl * a.vec < Ds;
    const int c0 = colok ? gl * a.vec : 0;
    for (int pass = 0; pass * groups < 32; ++pass) {
        const int i = pass * groups + grp;                 // < 32 because groups divides 32 (lpr >= 2)
        const int beg = ipt[i], end = ipt[i + 1];          // rows past nvalid: beg == end
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int e = beg; e < end; e += 4) {
            float w[4];
            v4f x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = e + u < end ? e + u : e;    // clamp: a real entry, result unused
                w[u] = gload1(a.adj_w + ee);
                const float *xp = a.state_cur + (int64_t)gload1(a.adj_src + ee) * Ds + c0;
                if (a.vec == 4) {
                    if constexpr (COH) x[u] = v4f{sload1<true>(xp), sload1<true>(xp + 1), sload1<true>(xp + 2), sload1<true>(xp + 3)};
                    else x[u] = gload4(xp);
                } else x[u] = v4f{sload1<COH>(xp), 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (e + u < end) {
                    acc[0] = __builtin_fmaf(w[u], x[u].x, acc[0]);
                    if (a.vec == 4) {
                        acc[1] = __builtin_fmaf(w[u], x[u].y, acc[1]); acc[2] = __builtin_fmaf(w[u], x[u].z, acc[2]);
                        acc[3] = __builtin_fmaf(w[u], x[u].w, acc[3]);
                    }
                }
            }
        }
        if (i < nvalid && colok) {
            float *x = X + i * KP + c_aggs + c0;
            x[0] = acc[0];
            if (a.vec == 4) { x[1] = acc[1]; x[2] = acc[2]; x[3] = acc[3]; }
        }
    }
}

// one entry of the Ds == 64 gather: acc += w * row piece (4 floats per lane), the oracle's fmaf chain.  Four v_fma_f32, as inline asm so that
// the compiler does not pair them into two v_pk_fma_f32 (same bits): MI355X_MICROARCH.md prices a packed f32 operation beside the SIMD partner's
// MFMAs at +22 cycles over the scalar pair; A/B of round 5 at BASELINE size: 0.683 against 0.687 ms per launch (profiles/r05_ab_gather_fma.txt).
__device__ __forceinline__ void gather_fma(v2f &acc01, v2f &acc23, float w, v4f x)
{
    float a0 = acc01.x, a1 = acc01.y, a2 = acc23.x, a3 = acc23.y;
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a0) : "v"(w), "v"(x.x));
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a1) : "v"(w), "v"(x.y));
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a2) : "v"(w), "v"(x.z));
    asm("v_fma_f32 %0, %1, %2, %0" : "+v"(a3) : "v"(w), "v"(x.w));
    acc01 = v2f{a0, a1}; acc23 = v2f{a2, a3};
}

// one batch of the Ds == 64 gather: entry J of the group's batch (held by lane J of the 16-lane row) is broadcast to the
// row, and the 16 lanes request the 256-byte neighbour row, 16 B each
template <int GB, int... J>
__device__ __forceinline__ void gather_batch(int my_src, float my_w, __amdgpu_buffer_rsrc_t rsrc, int voff0, float (&w)[GB],
                                             v4f (&x)[GB], std::integer_sequence<int, J...>)
{
    ((w[J] = row_bcast_f<J>(my_w),
      x[J] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rsrc, (row_bcast_i<J>(my_src) << 8) + voff0, 0, 0))), ...);
}

// Ds == 64, full tile.  Lane group g (16 lanes, 16 B per lane = one 256 B state row per group and instruction) owns the
// 8 consecutive nodes 8g..8g+7 and walks their contiguous CSR entries in batches of GB: ids/weights of a batch are fetched
// coalesced by the group (and those of the next batch prefetched), the GB neighbour rows are all requested before the
// first is consumed (GB x 4 groups x 256 B = 16 KiB in flight per wave), and the fmaf chain runs in stored order,
// flushing to LDS at every row boundary.
// AL16: rows of the tile and the aggregated-state block are 16-byte aligned (split arithmetic): one ds_write_b128 per row piece
// PADDED: the zero padding of the tile (columns behind the concat, alignment hole) is already in place: nothing in a tile's life
// writes those columns, so the full-tile kernel zeroes them once per wave instead of once per tile
// (Two batches of GB rows in flight per lane group were measured in round 2: 4 % slower, DESIGN.md appendix.)
template <bool AL16, bool PADDED = false>
__device__ __forceinline__ void load_tile_fast64(const GnnFusedArgs &a, float *X, const int *ipt, int64_t i0, int lane,
                                                 int KP, int c_aggs, int my_src, float my_w)
{
    constexpr int GB = 16, Ds = 64;
    const int gl = lane & 15, grp = lane >> 4;
    // own state rows: 8 x 16 B per lane requested now, written to LDS after the gather (their latency hides behind it)
    v4f own[8];
    {
        const float *src = a.state_cur + (a.row_begin + i0) * Ds + lane * 4;
#pragma unroll
        for (int u = 0; u < 8; ++u) own[u] = gload4(src + u * 256);      // (as non-temporal loads: +2.5 %, round 5 - these rows are other tiles' neighbour rows too)
    }
    float lab[4];
    const int IW = a.IW, nlab = 32 * IW;
    {
        const float *src = a.inv + i0 * IW;
#pragma unroll
        for (int u = 0; u < 4; ++u) lab[u] = (lane + 64 * u < nlab) ? gload1(src + lane + 64 * u) : 0.0f;
    }
    if constexpr (!PADDED) zero_pad_columns(a, X, lane, KP);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");                   // ipt visible to the whole wave

    int node = grp * 8;
    const int node_end = node + 8;
    const int e_begin = ipt[node], e_end = ipt[node_end];
    int next_end = ipt[node + 1];
    v2f acc01 = {0.f, 0.f}, acc23 = {0.f, 0.f};
    // neighbour rows through a buffer descriptor: 32-bit byte offsets (src * 256 + 16 * lane-in-row) instead of 64-bit
    // pointer arithmetic per row; the state replica is < 4 GiB by the fused path's precondition
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(a.state_cur), 0, (int)a.state_bytes, 0x00020000);
    const int voff0 = gl * 16;
    float *xo = X + c_aggs + gl * 4;
#define GNN_ROW_BOUNDARY(e)                                                                     \
    while ((e) >= next_end) {                                                                   \
        float *xr = xo + node * KP;                                                             \
        if constexpr (AL16) *reinterpret_cast<v4f *>(xr) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};   \
        else { xr[0] = acc01.x; xr[1] = acc01.y; xr[2] = acc23.x; xr[3] = acc23.y; }            \
        acc01 = v2f{0.f, 0.f}; acc23 = v2f{0.f, 0.f};                                           \
        ++node;                                                                                 \
        next_end = ipt[node + 1];                                                               \
    }
    // (my_src, my_w): ids / weights of the group's first batch, requested during the previous tile (tile_first_ids)
    int base = e_begin;
    for (; base + GB <= e_end; base += GB) {                                 // full batches: no guards
        float w[GB];
        v4f x[GB];
        gather_batch<GB>(my_src, my_w, rsrc, voff0, w, x, std::make_integer_sequence<int, GB>{});
        const int nb = base + GB + gl;                                       // ids / weights of the next batch
        my_src = 0; my_w = 0.0f;
        if (nb < e_end) { my_src = gstream1(a.adj_src + nb); my_w = gstream1(a.adj_w + nb); }
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            GNN_ROW_BOUNDARY(base + u)
            gather_fma(acc01, acc23, w[u], x[u]);
        }
    }
    {                                                                        // tail batch: cnt in [0, GB)
        const int cnt = e_end - base;
        float w[GB];
        v4f x[GB];
        // all GB slots are requested (slots >= cnt re-read whatever id the lane holds: entry of this batch or 0 = row 0,
        // always a valid row) and only the first cnt are consumed
        gather_batch<GB>(my_src, my_w, rsrc, voff0, w, x, std::make_integer_sequence<int, GB>{});
#pragma unroll
        for (int u = 0; u < GB; ++u) {
            if (u < cnt) {
                GNN_ROW_BOUNDARY(base + u)
                gather_fma(acc01, acc23, w[u], x[u]);
            }
        }
    }
#undef GNN_ROW_BOUNDARY
    for (; node < node_end; ++node) {                                        // last row with entries, then empty rows
        float *xr = xo + node * KP;
        if constexpr (AL16) *reinterpret_cast<v4f *>(xr) = v4f{acc01.x, acc01.y, acc23.x, acc23.y};
        else { xr[0] = acc01.x; xr[1] = acc01.y; xr[2] = acc23.x; xr[3] = acc23.y; }
        acc01 = v2f{0.f, 0.f}; acc23 = v2f{0.f, 0.f};
    }
    // own state and label columns into the tile
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float *x = X + (4 * u + (lane >> 4)) * KP + (lane & 15) * 4;         // flat element 256 u + 4 lane = row 4u + lane/16
        if constexpr (AL16) *reinterpret_cast<v4f *>(x) = own[u];
        else { x[0] = own[u].x; x[1] = own[u].y; x[2] = own[u].z; x[3] = own[u].w; }
    }
    const float inv_iw = 1.0f / (float)(IW > 0 ? IW : 1);                    // t / IW without an integer division: exact for t < 2^16
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int t = lane + 64 * u;
        if (t < nlab) {
            const int i = (int)(((float)t + 0.5f) * inv_iw), c = t - i * IW;
            X[i * KP + label_col(c, Ds, a.NLc, c_aggs)] = lab[u];
        }
    }
    if (nlab > 256)                                                          // wide label blocks: the rest, plainly
        for (int t = 256 + lane; t < nlab; t += 64) {
            const int i = t / IW, c = t - i * IW;
            X[i * KP + label_col(c, Ds, a.NLc, c_aggs)] = gload1(a.inv + i0 * IW + t);
        }
}

// Ds == 64, full tile, aggregate GIVEN (a.agg_in: rows of the aggregated state computed by another kernel): three coalesced row
// copies - own state, aggregate, label columns - instead of the gather.
template <bool AL16>
__device__ __forceinline__ void load_tile_given64(const GnnFusedArgs &a, float *X, int64_t i0, int lane, int KP, int c_aggs)
{
    constexpr int Ds = 64;
    v4f own[8], agg[8];
    {
        const float *src = a.state_cur + (a.row_begin + i0) * Ds + lane * 4;
        const float *sag = a.agg_in + i0 * Ds + lane * 4;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            own[u] = gload4(src + u * 256);
            agg[u] = gload4(sag + u * 256);
        }
    }
    float lab[4];
    const int IW = a.IW, nlab = 32 * IW;
    {
        const float *src = a.inv + i0 * IW;
#pragma unroll
        for (int u = 0; u < 4; ++u) lab[u] = (lane + 64 * u < nlab) ? gload1(src + lane + 64 * u) : 0.0f;
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        float *x = X + (4 * u + (lane >> 4)) * KP + (lane & 15) * 4;         // flat element 256 u + 4 lane = row 4u + lane/16
        if constexpr (AL16) { *reinterpret_cast<v4f *>(x) = own[u]; *reinterpret_cast<v4f *>(x + c_aggs) = agg[u]; }
        else {
            x[0] = own[u].x; x[1] = own[u].y; x[2] = own[u].z; x[3] = own[u].w;
            x[c_aggs] = agg[u].x; x[c_aggs + 1] = agg[u].y; x[c_aggs + 2] = agg[u].z; x[c_aggs + 3] = agg[u].w;
        }
    }
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

// condition() for the next body + coalesced store of the new rows.  New state sits in columns [c_aggs, c_aggs + Ds).
// lanes 0..31 sum (new - old)^2, lanes 32..63 sum old^2, ascending feature order, unfused (oracle order).
// moved_out != nullptr: the verdict "some node of the tile still moves" is returned there instead of raised in a.flag_out
// STORE == false: the condition only (k_small_loop stores the rows itself, into its padded exchange buffer)
template <bool COH = false, bool STORE = true>
__device__ __forceinline__ void check_store_generic(const GnnFusedArgs &a, float *X, int64_t i0, int lane, int nvalid, int KP,
                                                 int c_aggs, int *moved_out = nullptr)
{
    const int Ds = a.Ds, half = lane >> 5;
    if constexpr (STORE) {   // the row stores first: they drain while the condition is evaluated
        float *dst = a.state_nxt + i0 * Ds;
        const int total = nvalid * Ds;
        RowCol rc(lane, Ds);
        for (int t = lane; t < total; t += 64, rc.next()) sstore1<COH>(dst + t, X[rc.i * KP + c_aggs + rc.c]);
    }
    const float *xo = X + (lane & 31) * KP, *xn = xo + c_aggs;
    float s_ = 0.0f;
    for (int f = 0; f < Ds; ++f) {
        const float o = xo[f];
        const float d = half ? o : (xn[f] - o);
        const float dd = d * d;
        s_ = s_ + dd;
    }
    const float root = sqrtf(s_);
    const float nrm = shfl_f(root, (lane & 31) + 32);
    const float rhs = a.thr * nrm;
    const bool voter = (half == 0) && ((lane & 31) < nvalid);
    const int moved = voter && (root > rhs);
    if (moved_out) *moved_out = __any(moved) ? 1 : 0;
    else if (a.certify) {      // split arithmetic: certified gate
        const float band = GNN_BAND_ABS * nrm + GNN_BAND_REL * rhs;
        const int robust = voter && (root > rhs + band), border = voter && gnn_gate_borderline(root, rhs, band);
        const bool am = __any(moved), ar = __any(robust), ab = __any(border);
        if (lane == 0) gnn_flag_raise_certified(a.flag_out, am, ar, ab);
    }
    else if (__any(moved) && lane == 0) gnn_flag_raise(a.flag_out);
}

__device__ __forceinline__ void check_store_fast64(const GnnFusedArgs &a, float *X, int64_t i0, int lane, int KP, int c_aggs)
{
    constexpr int Ds = 64;
    const int half = lane >> 5;
    const float *xo = X + (lane & 31) * KP, *xn = xo + c_aggs;
    float s_ = 0.0f;
#pragma unroll
    for (int f = 0; f < Ds; f += 16) {
        float o[16], nw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { o[u] = xo[f + u]; nw[u] = xn[f + u]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float d = half ? o[u] : (nw[u] - o[u]);
            const float dd = d * d;
            s_ = s_ + dd;
        }
    }
    const float root = sqrtf(s_);
    const float nrm = shfl_f(root, (lane & 31) + 32);
    const float rhs = a.thr * nrm;
    const int moved = (half == 0) && (root > rhs);
    if (__any(moved) && lane == 0) gnn_flag_raise(a.flag_out);
    float *dst = a.state_nxt + i0 * Ds + lane;
    const float *xs = X + c_aggs + lane;                                     // row i, feature lane
#pragma unroll
    for (int i = 0; i < 32; i += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = xs[(i + u) * KP];
#pragma unroll
        for (int u = 0; u < 8; ++u) gstore_row1(dst + (i + u) * Ds, v[u]);
    }
}

// Split arithmetic, Ds == 64, full tile: last-layer epilogue output (already activated / normalised, in registers: lane = (node,
// half), 4 consecutive features per register quad) -> condition() for the next body and coalesced row stores.  Each lane sums
// its 32 features of (new - old)^2 and old^2 (old state: aligned 16-byte reads of the tile), the two halves of a node are added
// across lanes l / l + 32.  The summation order differs from the oracle's ascending-feature chain; on this path the state
// itself already differs from the oracle in the last bits, so that is within the same tolerance (k is compared in the tests).
// peek: the gate words of this wave's slot, requested by lane 0 behind the last weight loads (gnn_flag_peek)
__device__ __forceinline__ void finish_fast64_aligned(const GnnFusedArgs &a, float *X, f32x16 (&out)[2], int64_t i0, int lane, int KP, int c_aggs,
                                                      const GnnFlagPeek &peek)
{
    const int half = lane >> 5;
    float *xrow = X + (lane & 31) * KP;
    float d2 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * jt + 8 * q + 4 * half;
            const v4f o = *reinterpret_cast<const v4f *>(xrow + f0);
            const v4f nw = {out[jt][4 * q], out[jt][4 * q + 1], out[jt][4 * q + 2], out[jt][4 * q + 3]};
            *reinterpret_cast<v4f *>(xrow + c_aggs + f0) = nw;
            const v4f d = nw - o;
            d2 = __builtin_fmaf(d.x, d.x, d2); d2 = __builtin_fmaf(d.y, d.y, d2); d2 = __builtin_fmaf(d.z, d.z, d2); d2 = __builtin_fmaf(d.w, d.w, d2);
            o2 = __builtin_fmaf(o.x, o.x, o2); o2 = __builtin_fmaf(o.y, o.y, o2); o2 = __builtin_fmaf(o.z, o.z, o2); o2 = __builtin_fmaf(o.w, o.w, o2);
        }
    d2 = d2 + shfl_f(d2, lane ^ 32);
    o2 = o2 + shfl_f(o2, lane ^ 32);
    const float root = sqrtf(d2), nrm = sqrtf(o2);
    {   // certified gate (gnn_common.h): both half-lanes of a node hold the same sums
        const float rhs = a.thr * nrm, band = GNN_BAND_ABS * nrm + GNN_BAND_REL * rhs;
        const bool am = __any(root > rhs), ar = __any(root > rhs + band), ab = __any(gnn_gate_borderline(root, rhs, band));
        if (lane == 0) gnn_flag_raise_peeked(a.flag_out, peek, am, ar, ab);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int64_t dst = i0 * 64 + lane * 4;                                  // flat element 256 u + 4 lane = row 4u + lane/16
    const float *xs = X + (lane >> 4) * KP + c_aggs + (lane & 15) * 4;
    v4f v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const v4f *>(xs + 4 * u * KP);
#pragma unroll
    for (int u = 0; u < 8; ++u) gstore_row4(a.state_nxt, dst + 256 * u, v[u]);
}

// The same two for the LAST tile of a range whose row count is not a multiple of 32 (wave-uniform branch in the full-tile kernel: a
// second, one-tile launch of the general kernel for that tile cost 70 - 100 us per iteration).  Rows >= nvalid of the tile exist
// in memory (the buffers are padded to whole tiles) and have no arcs; what they compute stays in LDS: their lanes do not vote in
// condition() and their rows are not stored.
__device__ __forceinline__ void check_store_fast64_partial(const GnnFusedArgs &a, float *X, int64_t i0, int lane, int KP, int c_aggs, int nvalid)
{
    constexpr int Ds = 64;
    const int half = lane >> 5;
    const float *xo = X + (lane & 31) * KP, *xn = xo + c_aggs;
    float s_ = 0.0f;
#pragma unroll
    for (int f = 0; f < Ds; f += 16) {
        float o[16], nw[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) { o[u] = xo[f + u]; nw[u] = xn[f + u]; }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const float d = half ? o[u] : (nw[u] - o[u]);
            const float dd = d * d;
            s_ = s_ + dd;
        }
    }
    const float root = sqrtf(s_);
    const float nrm = shfl_f(root, (lane & 31) + 32);
    const float rhs = a.thr * nrm;
    const int moved = (half == 0) && (lane < nvalid) && (root > rhs);
    if (__any(moved) && lane == 0) gnn_flag_raise(a.flag_out);
    float *dst = a.state_nxt + i0 * Ds + lane;
    const float *xs = X + c_aggs + lane;                                     // row i, feature lane
    for (int i = 0; i < nvalid; ++i) gptr_w(dst)[i * Ds] = xs[i * KP];
}

__device__ __forceinline__ void finish_fast64_partial(const GnnFusedArgs &a, float *X, f32x16 (&out)[2], int64_t i0, int lane, int KP, int c_aggs,
                                                      int nvalid)
{
    const int half = lane >> 5;
    float *xrow = X + (lane & 31) * KP;
    float d2 = 0.0f, o2 = 0.0f;
#pragma unroll
    for (int jt = 0; jt < 2; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = 32 * jt + 8 * q + 4 * half;
            const v4f o = *reinterpret_cast<const v4f *>(xrow + f0);
            const v4f nw = {out[jt][4 * q], out[jt][4 * q + 1], out[jt][4 * q + 2], out[jt][4 * q + 3]};
            *reinterpret_cast<v4f *>(xrow + c_aggs + f0) = nw;
            const v4f d = nw - o;
            d2 = __builtin_fmaf(d.x, d.x, d2); d2 = __builtin_fmaf(d.y, d.y, d2); d2 = __builtin_fmaf(d.z, d.z, d2); d2 = __builtin_fmaf(d.w, d.w, d2);
            o2 = __builtin_fmaf(o.x, o.x, o2); o2 = __builtin_fmaf(o.y, o.y, o2); o2 = __builtin_fmaf(o.z, o.z, o2); o2 = __builtin_fmaf(o.w, o.w, o2);
        }
    d2 = d2 + shfl_f(d2, lane ^ 32);
    o2 = o2 + shfl_f(o2, lane ^ 32);
    const float root = sqrtf(d2), nrm = sqrtf(o2);
    {   // certified gate (gnn_common.h); rows past the end of the range do not vote
        const bool voter = (lane & 31) < nvalid;
        const float rhs = a.thr * nrm, band = GNN_BAND_ABS * nrm + GNN_BAND_REL * rhs;
        const bool am = __any(voter && root > rhs), ar = __any(voter && root > rhs + band), ab = __any(voter && gnn_gate_borderline(root, rhs, band));
        if (lane == 0) gnn_flag_raise_certified(a.flag_out, am, ar, ab);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    float *dst = a.state_nxt + i0 * 64 + lane * 4;                           // flat element 256 u + 4 lane = row 4u + lane/16
    const float *xs = X + (lane >> 4) * KP + c_aggs + (lane & 15) * 4;
    for (int u = 0; u < 8; ++u)
        if (4 * u + (lane >> 4) < nvalid) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(dst) + 256 * u) = *reinterpret_cast<const v4f *>(xs + 4 * u * KP);
}

// Graph readout of the persistent small-graph loops (k_small_loop / k_small16; reference GNN/GNN.py:331-332, arithmetic of k_readout):
// out_graph[g, t] = sum over the (node, w) of graph g, ascending, fmaf(w, out[node, t]), one lane per (g, t), by ONE workgroup after the
// grid barrier behind the output stage.  The entries of a graph are taken eight at a time: their (node, w) pairs are requested
// together, then the eight output values (sc1 loads: other workgroups wrote them in this launch), then the eight fmaf in stored order -
// two round trips per eight nodes instead of two per node (a MUTAG graph has 18: 36 dependent round trips were 36 us of a 128 us Loop).
__device__ __forceinline__ void small_graph_readout(const GnnSmallCtl &c, int lane)
{
    for (int t = lane; t < c.G * c.T; t += 64) {
        const int gi = t / c.T, ci = t - gi * c.T;
        const int e0 = gload1(c.ng_ip + gi), e1 = gload1(c.ng_ip + gi + 1);
        float acc = 0.0f;
        for (int e = e0; e < e1; e += 8) {
            int node[8];
            float w[8], v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int ee = e + u < e1 ? e + u : e;          // clamp: a real entry, result unused
                node[u] = gload1(c.ng_node + ee);
                w[u] = gload1(c.ng_w + ee);
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = sload1<true>(c.out + (int64_t)node[u] * c.T + ci);
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (e + u < e1) acc = __builtin_fmaf(w[u], v[u], acc);
        }
        c.ng_host[t] = acc;
    }
}

// ---- cross-tile prefetch of the dependent loads that open a tile: ticket -> row pointers -> ids of the first gather batch ----
// Each of them is a full memory round trip in front of the first neighbour row; requested one tile ahead they cost three VGPRs.
__device__ __forceinline__ int tile_rowptr_request(const GnnFusedArgs &a, int tile, int lane)
{
    const int64_t i0 = (int64_t)tile * 32;
    if (i0 >= a.n_rows) return 0;
    const int nvalid = (int)((a.n_rows - i0) < 32 ? (a.n_rows - i0) : 32);
    return (lane <= nvalid) ? gstream1(a.indptr + i0 + lane) : 0;
}
// rows past the end of a partial tile get the last pointer (empty rows); lanes 33.. hold it too
__device__ __forceinline__ int tile_rowptr_clamp(const GnnFusedArgs &a, int tile, int lane, int raw)
{
    const int64_t i0 = (int64_t)tile * 32;
    if (i0 >= a.n_rows) return 0;
    const int nvalid = (int)((a.n_rows - i0) < 32 ? (a.n_rows - i0) : 32);
    const int last_ip = shfl_i(raw, nvalid);
    return lane <= nvalid ? raw : last_ip;
}
// lane group g = lane >> 4 owns rows 8g .. 8g+7: ids / weights of its first 16 entries (Ds == 64 gather)
__device__ __forceinline__ void tile_first_ids(const GnnFusedArgs &a, int ip, int lane, int &src, float &w)
{
    const int gl = lane & 15, grp = lane >> 4;
    const int e_begin = shfl_i(ip, grp * 8), e_end = shfl_i(ip, grp * 8 + 8);
    src = 0; w = 0.0f;
    if (e_begin + gl < e_end) { src = gstream1(a.adj_src + e_begin + gl); w = gstream1(a.adj_w + e_begin + gl); }
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[N])
{
#pragma unroll
    for (int jt = 0; jt < N; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
}

// FULL: state width 64, the tuned 32-node tile paths only (a partial last tile takes their masked branch): the guarded generic paths
// are not compiled in at all, which frees registers and scalar registers for the tuned ones.
// GIVEN (with FULL): the aggregated states come from a.agg_in (feature-sliced exchange) - row copies instead of the gather.  A
// template parameter, not a branch: a wave-uniform branch in the tile loop of the full-tile kernel cost 3 % (0.700 -> 0.721 ms).
template <int LAYERS, int NT, int NTL, int ACT, bool SPLIT, bool FULL = false, bool GIVEN = false>
__global__ void __launch_bounds__(GNN_FUSED_THREADS, 2) k_fused(const GnnFusedArgs a0)
{
    const GnnFusedArgs &a = a0;      // (shadowed inside the tile loop)
    // Persistent workgroup of 8 waves (one per CU): every wave pulls 32-node tiles from a device-wide counter until none
    // is left.  Waves w and w + 4 share a SIMD; the second half starts late so that, in steady state, one partner streams
    // neighbour rows from HBM while the other runs its MFMA / activation phases (without the offset all waves of a CU
    // gather at once and then compute at once: HBM idles during compute, the matrix pipe during the gather).
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!gnn_gate_open(a.gate, a.world)) return;
#ifdef GNN_DIAG      // GNN_POISON=1: NaN over the whole LDS allocation (tiles, alignment holes, look-ahead slack, staged vectors) before anything is staged
    if (a.lds_floats) {
        for (int t = threadIdx.x; t < a.lds_floats; t += blockDim.x) lds[t] = __builtin_nanf("");
        __syncthreads();
    }
#endif
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = a.KP, Ds = a.Ds;
    float *X = lds + (size_t)wave * 32 * KP;
    const int c_aggs = a.c_aggs;                      // column of the aggregated state block (Ds + NLc + alignment hole)
    // Tiles: the first TWO rounds are assigned statically - wave w of the launch (wave-major over the workgroups, so that a partial round
    // spreads over all CUs) takes tiles w and w + W, W = waves of the launch - and only from the third round on does a wave draw tickets
    // from the iteration's counter, one tile ahead (at the end of the previous tile's dense layers), early enough to have that tile's row
    // pointers and first gather ids requested.  (Until round 4 every wave drew its first two tickets at kernel start: 4,096 atomic adds on
    // ONE word, which the memory side serves at about 88 per microsecond - up to 46 us before the last wave knew its first tile, a third
    // of an iteration at 125 k nodes.  Serving the tickets heaviest-tile-first was measured in round 2: 1 % slower on the BASELINE graph.)
    const int W_launch = (int)gridDim.x * (int)(blockDim.x >> 6);
    const int w_launch = wave * (int)gridDim.x + (int)blockIdx.x;
    const bool third_round = (int64_t)2 * W_launch * 32 < a.n_rows;          // wave-uniform: tickets are only drawn when tiles beyond 2 W exist
    int tile = w_launch + a.tile_base, next_tile = w_launch + W_launch + a.tile_base;
    const int ip_first_raw = tile_rowptr_request(a, tile, lane);      // on its way while the workgroup stages its vectors below
    // last-layer bias and BatchNormalization scale / shift: staged once per workgroup behind the row-pointer slots
    float *ep = lds + (size_t)GNN_FUSED_WAVES * 32 * KP + 32 + GNN_FUSED_WAVES * 36;
    for (int t = threadIdx.x; t < 3 * 32 * NTL; t += blockDim.x) {
        const int which = t / (32 * NTL), f = t - which * 32 * NTL;
        ep[t] = which == 0 ? a.bias[LAYERS - 1][f] : (a.bn_scale ? (which == 1 ? a.bn_scale[f] : a.bn_shift[f]) : 0.0f);
    }
    float *hb = ep + 3 * 32 * NTL;                    // hidden-layer biases (split path): [LAYERS - 1][32 NT]
    if constexpr (SPLIT && LAYERS > 1)
        for (int t = threadIdx.x; t < (LAYERS - 1) * 32 * NT; t += blockDim.x)
            hb[t] = a.bias[t / (32 * NT)][t % (32 * NT)] * (ACT == GNN_ACT_SELU ? 1.44269504088896341f : 1.0f);      // folded SELU: see GNN_S1_E
    __syncthreads();
    // Start-up spread.  All waves of the chip run the same phases on tiles of similar cost: started together they gather together
    // (HBM saturated, 3-4 us per round trip) and compute together (HBM idle).  Every wave therefore waits a different fraction of
    // one tile period before its first tile; the dynamic tickets keep the load balanced.  variant bit 2: the old two-cluster
    // stagger (waves 4-7 delayed by a fixed amount).
    if (a.stagger > 0) {
        int rounds = 0;
        if (a.variant & 4) rounds = wave >= GNN_FUSED_WAVES / 2 ? a.stagger : 0;
        else rounds = (int)((((unsigned)blockIdx.x * GNN_FUSED_WAVES + (unsigned)wave) * 0x9E3779B1u) >> 16) % (unsigned)(a.stagger + 1);
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
    int ip_cur = tile_rowptr_clamp(a, tile, lane, ip_first_raw);
    int src_cur = 0;
    float w_cur = 0.0f;
    if (FULL || Ds == 64) tile_first_ids(a, ip_cur, lane, src_cur, w_cur);
    if constexpr (FULL) zero_pad_columns(a, X, lane, KP);             // once: no tile ever writes the padding columns
  for (;;) {
    const int64_t i0 = (int64_t)tile * 32;
    if (i0 >= a.n_rows) break;                        // wave-uniform; no workgroup barrier anywhere in the kernel
    const int nvalid = (int)((a0.n_rows - i0) < 32 ? (a0.n_rows - i0) : 32);
    // Fresh, compiler-opaque copies of the pointers for every tile: without this the loop-invariant address arithmetic of
    // the unrolled layers is hoisted out of the tile loop and spills (256 VGPRs + scratch instead of ~190).
    GnnFusedArgs a = a0;
    asm volatile("" : "+s"(a.Wp[0]), "+s"(a.Wp[1]), "+s"(a.Wp[2]), "+s"(a.bias[0]), "+s"(a.bias[1]), "+s"(a.bias[2]));
    asm volatile("" : "+s"(a.bn_scale), "+s"(a.bn_shift), "+s"(a.state_cur), "+s"(a.state_nxt), "+s"(a.inv), "+s"(a.adj_src), "+s"(a.adj_w));
#ifndef GNN_DIAG
#define GNN_STAMP(slot) do { } while (0)
#else
    unsigned long long *stamp = a.stamps ? a.stamps + ((size_t)tile << 3) : nullptr;
#define GNN_STAMP(slot)                                                                      \
    do {                                                                                     \
        if (stamp) {                                                                         \
            __builtin_amdgcn_sched_barrier(0);                                               \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                      \
            if (lane == 0) stamp[slot] = t_;                                                 \
            __builtin_amdgcn_sched_barrier(0);                                               \
        }                                                                                    \
    } while (0)
#endif
    GNN_STAMP(0);

    // the tile's 33 row pointers go through LDS: the gather re-reads them inside divergent code, where a cross-lane
    // broadcast from lanes of another group would not be safe
    int *ipt = reinterpret_cast<int *>(lds + (size_t)GNN_FUSED_WAVES * 32 * KP + 32) + wave * 36;
    if (lane <= 32) ipt[lane] = ip_cur;               // requested during the previous tile
    const int ip_next_raw = tile_rowptr_request(a, next_tile, lane);      // row pointers of the NEXT tile: on their way during the gather
    const bool fast64 = FULL || ((Ds == 64) && (nvalid == 32) && !a.agg_in);     // wave-uniform: the BASELINE shape takes the unguarded paths
    // the gather is a chain of few instructions and long memory waits: with a raised priority its loads are issued ahead of the
    // SIMD partner's dense VALU / MFMA stream instead of behind it
    if (a.variant & 1) __builtin_amdgcn_s_setprio(3);
    if constexpr (FULL && GIVEN) load_tile_given64<SPLIT>(a, X, i0, lane, KP, c_aggs);      // feature-sliced exchange: no gather (a.agg_in)
    else if constexpr (FULL) load_tile_fast64<SPLIT, true>(a, X, ipt, i0, lane, KP, c_aggs, src_cur, w_cur);
    else {
        if (fast64) load_tile_fast64<SPLIT>(a, X, ipt, i0, lane, KP, c_aggs, src_cur, w_cur);
        else load_tile_generic(a, X, ipt, i0, lane, nvalid, KP, c_aggs);
    }
    if (a.variant & 1) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    GNN_STAMP(2);

    // ---- B: net_state.  H^T[feature][node] = W^T . X^T on MFMA f32 32x32x2 -------------------------------------------
    const int half = lane >> 5;
    const float *xb = X + (lane & 31) * KP + half;
    f32x16 out[NTL];
    if constexpr (SPLIT) {
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.Ws_base), 0, a.ws_bytes, 0x00020000);
        const int wv = lane * 16;
        const float *xr = X + (lane & 31) * KP + 8 * half;
        if constexpr (LAYERS == 1) {
            layer0_split<NTL, false>(xr, wrs, wv, a.ws_off[0], a.chunks0, out, ep, half);
        } else {
            f32x16 h1[NT];
            if constexpr (FULL) layer0_split<NT, true>(xr, wrs, wv, a.ws_off[0], a.chunks0, h1, hb, half);
            else {
                if (Ds == 64) layer0_split<NT, true>(xr, wrs, wv, a.ws_off[0], a.chunks0, h1, hb, half);    // 16-byte aligned tile layout
                else layer0_split<NT, false>(xr, wrs, wv, a.ws_off[0], a.chunks0, h1, hb, half);
            }
            GNN_STAMP(3);
            GNN_STAMP(4);
            if constexpr (LAYERS == 2) {
                layer_split_from_regs<NT, NTL, ACT>(h1, ep, half, out, wrs, wv, a.ws_off[1]);
            } else {
                f32x16 h2[NT];
                layer_split_from_regs<NT, NT, ACT>(h1, hb + 32 * NT, half, h2, wrs, wv, a.ws_off[1]);
                layer_split_from_regs<NT, NTL, ACT>(h2, ep, half, out, wrs, wv, a.ws_off[2]);
            }
        }
    } else if constexpr (LAYERS == 1) {
        zero_acc<NTL>(out);
        layer_from_lds<NTL>(xb, a.Wp[0] + (size_t)lane * NTL, a.kk0, out, a.wstride);
    } else {
        f32x16 h1[NT];
        zero_acc<NT>(h1);
        layer_from_lds<NT>(xb, a.Wp[0] + (size_t)lane * NT, a.kk0, h1, a.wstride);
        GNN_STAMP(3);
        GNN_STAMP(4);
        if constexpr (LAYERS == 2) {
            zero_acc<NTL>(out);
            layer_from_regs<NT, NTL, ACT>(h1, a.bias[0], half, out, a.Wp[1] + (size_t)lane * NTL, a.wstride);
        } else {
            f32x16 h2[NT];
            zero_acc<NT>(h2);
            layer_from_regs<NT, NT, ACT>(h1, a.bias[0], half, h2, a.Wp[1] + (size_t)lane * NT, a.wstride);
            zero_acc<NTL>(out);
            layer_from_regs<NT, NTL, ACT>(h2, a.bias[1], half, out, a.Wp[2] + (size_t)lane * NTL, a.wstride);
        }
    }
    // ---- C: last-layer epilogue, new state to LDS (over the aggregated-state columns, no longer needed) ---------------
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    GNN_STAMP(5);
    // Requests for the tile after next (ticket) and for the next tile (ids / weights of its first gather batch): issued HERE,
    // behind the last weight loads, because vector-memory results return in issue order - in front of the dense layers these
    // HBM-latency loads would hold up every weight wait - and the epilogue / norm / stores below cover their latency.
    // (no draw when the launch has no third round: two static tiles per wave cover it)
    int next2_tile = 0x3fffffff;
    if (third_round) {
        if (lane == 0) next2_tile = atomicAdd(a0.tile_ctr, 1) + 2 * W_launch;
    }
    const int ip_next = tile_rowptr_clamp(a, next_tile, lane, ip_next_raw);
    int src_next = 0;
    float w_next = 0.0f;
    if (FULL || Ds == 64) tile_first_ids(a, ip_next, lane, src_next, w_next);
    GnnFlagPeek peek = {0, 0, 0};                             // the gate words of this wave's slot: on their way across the epilogue arithmetic
    if (SPLIT && NTL == 2 && lane == 0) peek = gnn_flag_peek(a.flag_out);
    bool finished = false;
    if constexpr (SPLIT && NTL == 2) {
        if (fast64) {                                         // registers -> norms, LDS (16-byte pieces), row stores
#pragma unroll
            for (int jt = 0; jt < NTL; ++jt) {
                if (a.bn_scale) tile_epilogue<ACT, true, true, true, true>(out[jt], ep, ep + 32 * NTL, ep + 64 * NTL, jt, half);
                else tile_epilogue<ACT, false, true, true, true>(out[jt], ep, nullptr, nullptr, jt, half);
            }
            GNN_STAMP(6);
            if (nvalid == 32) finish_fast64_aligned(a, X, out, i0, lane, KP, c_aggs, peek);
            else finish_fast64_partial(a, X, out, i0, lane, KP, c_aggs, nvalid);       // (full-tile kernel on the range's last, partial tile)
            finished = true;
        }
    }
    if (!(FULL && SPLIT && NTL == 2) && !finished) {
#pragma unroll
        for (int jt = 0; jt < NTL; ++jt) {
            if (a.bn_scale) tile_epilogue<ACT, true, SPLIT, true, SPLIT>(out[jt], ep, ep + 32 * NTL, ep + 64 * NTL, jt, half);
            else tile_epilogue<ACT, false, SPLIT, true, SPLIT>(out[jt], ep, nullptr, nullptr, jt, half);
            float *x = X + (lane & 31) * KP + c_aggs;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (f < Ds) x[f] = out[jt][r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        GNN_STAMP(6);
        if constexpr (FULL) {
            if (nvalid == 32) check_store_fast64(a, X, i0, lane, KP, c_aggs);
            else check_store_fast64_partial(a, X, i0, lane, KP, c_aggs, nvalid);
        } else {
            if (fast64) check_store_fast64(a, X, i0, lane, KP, c_aggs);
            else check_store_generic(a, X, i0, lane, nvalid, KP, c_aggs);
        }
    }
    GNN_STAMP(7);
#undef GNN_STAMP
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");      // the next tile re-uses this wave's LDS region
    tile = next_tile;
    next_tile = __builtin_amdgcn_readfirstlane(next2_tile) + a0.tile_base;
    ip_cur = ip_next; src_cur = src_next; w_cur = w_next;
  }
}

template <int LAYERS, int NT, int NTL, int ACT, bool SPLIT, bool FULL, bool GIVEN = false>
inline void launch_one(const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    static bool raised[64] = {false};   // dynamic LDS above 64 KiB has to be requested once per kernel AND device
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused<LAYERS, NT, NTL, ACT, SPLIT, FULL, GIVEN>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) raised[dev] = true;
    }
    hipLaunchKernelGGL((k_fused<LAYERS, NT, NTL, ACT, SPLIT, FULL, GIVEN>), grid, a.threads ? a.threads : GNN_FUSED_THREADS, lds_bytes, st, a);
}

// a.full_tiles: the host asks for the full-tile specialisation (state width 64); it exists for NTL == 2
template <int LAYERS, int NT, int NTL, int ACT, bool SPLIT>
inline void launch(const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    if constexpr (NTL == 2) {
        if (a.full_tiles && a.agg_in) { launch_one<LAYERS, NT, NTL, ACT, SPLIT, true, true>(a, grid, lds_bytes, st); return; }
        if (a.full_tiles) { launch_one<LAYERS, NT, NTL, ACT, SPLIT, true>(a, grid, lds_bytes, st); return; }
    }
    launch_one<LAYERS, NT, NTL, ACT, SPLIT, false>(a, grid, lds_bytes, st);
}

// (NT, NTL) pairs that are instantiated; gnn_fused.hip rounds every net up to one of them
template <int LAYERS, int ACT, bool SPLIT>
inline bool launch_tiles(int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    if constexpr (LAYERS == 1) {
        if (ntl == 1) launch<1, 1, 1, ACT, SPLIT>(a, grid, lds_bytes, st);
        else if (ntl == 2) launch<1, 2, 2, ACT, SPLIT>(a, grid, lds_bytes, st);
        else if (ntl == 4) launch<1, 4, 4, ACT, SPLIT>(a, grid, lds_bytes, st);
        else return false;
    } else {
        if (nt == 1 && ntl == 1) launch<LAYERS, 1, 1, ACT, SPLIT>(a, grid, lds_bytes, st);
        else if (nt == 2 && ntl == 2) launch<LAYERS, 2, 2, ACT, SPLIT>(a, grid, lds_bytes, st);
        else if (nt == 4 && ntl == 2) launch<LAYERS, 4, 2, ACT, SPLIT>(a, grid, lds_bytes, st);
        else if (nt == 4 && ntl == 4) launch<LAYERS, 4, 4, ACT, SPLIT>(a, grid, lds_bytes, st);
        else return false;
    }
    return true;
}

template <int LAYERS, bool SPLIT>
inline bool launch_act(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    switch (act) {
    case GNN_ACT_LINEAR: return launch_tiles<LAYERS, GNN_ACT_LINEAR, SPLIT>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_RELU: return launch_tiles<LAYERS, GNN_ACT_RELU, SPLIT>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_SELU: return launch_tiles<LAYERS, GNN_ACT_SELU, SPLIT>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_ELU: return launch_tiles<LAYERS, GNN_ACT_ELU, SPLIT>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_TANH: return launch_tiles<LAYERS, GNN_ACT_TANH, SPLIT>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_SIGMOID: return launch_tiles<LAYERS, GNN_ACT_SIGMOID, SPLIT>(nt, ntl, a, grid, lds_bytes, st);
    default: return false;
    }
}

}   // namespace gnn_fused_dev
