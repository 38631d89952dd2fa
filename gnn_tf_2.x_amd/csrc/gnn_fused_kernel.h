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
#include "gnn_common.h"
#include "gnn_fused.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace gnn_fused_dev {

__device__ __forceinline__ float shfl_f(float v, int src_lane)
{
    return __int_as_float(__builtin_amdgcn_ds_bpermute(src_lane << 2, __float_as_int(v)));
}
__device__ __forceinline__ int shfl_i(int v, int src_lane) { return __builtin_amdgcn_ds_bpermute(src_lane << 2, v); }

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
        const float4 t = *reinterpret_cast<const float4 *>(p);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    } else if constexpr (N == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        w[0] = t.x; w[1] = t.y;
    } else {
        w[0] = p[0];
    }
}

template <int ACT>
__device__ __forceinline__ float act_t(float v)
{
    return gnn_act(v, ACT);     // ACT is a compile-time constant: the switch folds
}

// bias + activation (+ BatchNormalization when BN) on one accumulator tile; feature of register r on this lane:
// 32 jt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
template <int ACT, bool BN>
__device__ __forceinline__ void tile_epilogue(f32x16 &a, const float *bias, const float *bn_scale, const float *bn_shift,
                                              int jt, int half)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * jt + 8 * q + 4 * half;
        const float4 b = *reinterpret_cast<const float4 *>(bias + f0);
        const float bb[4] = {b.x, b.y, b.z, b.w};
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (BN) {
            const float4 s4 = *reinterpret_cast<const float4 *>(bn_scale + f0), h4 = *reinterpret_cast<const float4 *>(bn_shift + f0);
            sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
            sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = a[4 * q + t] + bb[t];
            v = act_t<ACT>(v);
            if (BN) { const float m = v * sc[t]; v = m + sh[t]; }
            a[4 * q + t] = v;
        }
    }
}

// Dense layer whose input is the LDS tile X (layer 0).  KK (multiple of 8) K-steps, software pipelined in groups of 4:
// the weight loads and LDS reads of group g+1 are issued before the MFMAs of group g.
template <int NO>
__device__ __forceinline__ void layer_from_lds(const float *xb, const float *wp, int kk_total, f32x16 (&acc)[NO])
{
    constexpr int PF = 4;
    float wa[PF][NO], wb[PF][NO], ba[PF], bb[PF];
#pragma unroll
    for (int u = 0; u < PF; ++u) { load_w<NO>(wp + (size_t)u * 64 * NO, wa[u]); ba[u] = xb[2 * u]; }
    for (int kk = 0; kk < kk_total; kk += 2 * PF) {
#pragma unroll
        for (int u = 0; u < PF; ++u) { load_w<NO>(wp + (size_t)(kk + PF + u) * 64 * NO, wb[u]); bb[u] = xb[2 * (kk + PF + u)]; }
        __builtin_amdgcn_sched_barrier(0);      // keep the prefetch ABOVE the MFMAs it overlaps with
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int jt = 0; jt < NO; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[u][jt], ba[u], acc[jt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        // group kk + 2 PF (reads past the end land in the zero padding of the packed image / LDS tile: see gnn_fused.hip)
#pragma unroll
        for (int u = 0; u < PF; ++u) { load_w<NO>(wp + (size_t)(kk + 2 * PF + u) * 64 * NO, wa[u]); ba[u] = xb[2 * (kk + 2 * PF + u)]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int u = 0; u < PF; ++u)
#pragma unroll
            for (int jt = 0; jt < NO; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[u][jt], bb[u], acc[jt], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
}

// Dense layer whose input sits in registers as B operands (hin, after acc_to_operand): 16 NI K-steps, fully unrolled,
// weight loads issued DEPTH steps ahead of their MFMAs.
template <int NI, int NO>
__device__ __forceinline__ void layer_from_regs(const f32x16 (&hin)[NI], f32x16 (&acc)[NO], const float *wp)
{
    constexpr int STEPS = 16 * NI, DEPTH = 8;
    float w[STEPS][NO];
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) load_w<NO>(wp + (size_t)s * 64 * NO, w[s]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + DEPTH < STEPS) load_w<NO>(wp + (size_t)(s + DEPTH) * 64 * NO, w[s + DEPTH]);
        __builtin_amdgcn_sched_barrier(0);      // the load stays DEPTH steps ahead of its MFMAs
        const int ti = s >> 4, ss = s & 15;
        const int reg = 4 * (ss >> 2) + ((ss & 3) == 1 ? 2 : (ss & 3) == 2 ? 1 : (ss & 3));
        const float b = hin[ti][reg];
#pragma unroll
        for (int jt = 0; jt < NO; ++jt) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[s][jt], b, acc[jt], 0, 0, 0);
    }
}

template <int N>
__device__ __forceinline__ void zero_acc(f32x16 (&acc)[N])
{
#pragma unroll
    for (int jt = 0; jt < N; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
}

template <int LAYERS, int NT, int NTL, int ACT>
__global__ void __launch_bounds__(256, 2) k_fused(const GnnFusedArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!gnn_gate_open(a.gate, a.world)) return;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + wave) * 32;
    if (i0 >= a.n_rows) return;                       // wave-uniform; no workgroup barrier anywhere below
    const int nvalid = (int)((a.n_rows - i0) < 32 ? (a.n_rows - i0) : 32);
    const int KP = a.KP, Ds = a.Ds, NLc = a.NLc;
    float *X = lds + (size_t)wave * 32 * KP;
    const int c_aggs = Ds + NLc;                      // column of the aggregated state block

    // ---- A0: zero what no phase below writes: pad columns [in_s, KP) and the rows of a partial last tile -------------
    {
        const int padw = KP - a.in_s;
        for (int t = lane; t < 32 * padw; t += 64) {
            const int i = t / padw, c = t - i * padw;
            X[i * KP + a.in_s + c] = 0.0f;
        }
        if (nvalid < 32)
            for (int t = lane; t < (32 - nvalid) * a.in_s; t += 64) {
                const int i = nvalid + t / a.in_s, c = t % a.in_s;
                X[i * KP + c] = 0.0f;
            }
    }
    // ---- A1: own state rows (contiguous in HBM) into columns [0, Ds) ------------------------------------------------
    {
        const float *src = a.state_cur + (a.row_begin + i0) * Ds;
        const int total = nvalid * Ds;
        if (a.vec == 4) {
            int i = (lane * 4) / Ds, f = (lane * 4) - i * Ds;
            for (int t = lane * 4; t < total; t += 256) {
                const float4 v = *reinterpret_cast<const float4 *>(src + t);
                float *x = X + i * KP + f;
                x[0] = v.x; x[1] = v.y; x[2] = v.z; x[3] = v.w;
                f += 256;
                while (f >= Ds) { f -= Ds; ++i; }
            }
        } else {
            int i = lane / Ds, f = lane - i * Ds;
            for (int t = lane; t < total; t += 64) {
                X[i * KP + f] = src[t];
                f += 64;
                while (f >= Ds) { f -= Ds; ++i; }
            }
        }
    }
    // ---- A2: loop-invariant label columns --------------------------------------------------------------------------
    if (a.IW > 0) {
        const float *src = a.inv + i0 * a.IW;
        const int total = nvalid * a.IW;
        for (int t = lane; t < total; t += 64) {
            const int i = t / a.IW, c = t - i * a.IW;
            // [nodes | agg nodes | agg arcs] -> columns Ds.., 2Ds+NLc.., 2Ds+2NLc..
            const int col = c < NLc ? Ds + c : (c < 2 * NLc ? c_aggs + Ds + (c - NLc) : 2 * Ds + 2 * NLc + (c - 2 * NLc));
            X[i * KP + col] = src[t];
        }
    }
    // ---- A3: gather: aggregated_states = Adjacency^T . state (GNN.py:234) --------------------------------------------
    {
        const int lpr = a.lpr, gl = lane & (lpr - 1), grp = lane >> a.lpr_log2, groups = 64 >> a.lpr_log2;
        const int my_ip = (lane <= nvalid) ? a.indptr[i0 + lane] : 0;      // lanes 0..32 hold the tile's row pointers
        // one column chunk per lane (lpr * vec >= Ds is a precondition of the fused path); lanes past the row width still
        // walk the edges (they feed the broadcasts) on column 0 and store nothing
        const bool colok = gl * a.vec < Ds;
        const int c0 = colok ? gl * a.vec : 0;
        for (int pass = 0; pass * groups < 32; ++pass) {
            const int i = pass * groups + grp;
            const int beg = shfl_i(my_ip, i < nvalid ? i : 0), end = shfl_i(my_ip, i < nvalid ? i + 1 : 0);
            float acc[4] = {0.f, 0.f, 0.f, 0.f};
            for (int base = beg; base < end; base += lpr) {
                const int e = base + gl;
                const bool has = e < end;
                const int my_src = has ? a.adj_src[e] : 0;
                const float my_w = has ? a.adj_w[e] : 0.0f;
                const int cnt = (end - base) < lpr ? (end - base) : lpr;
                for (int j = 0; j < cnt; j += 4) {
                    float w[4];
                    float x[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int jj = (j + u < cnt) ? j + u : j;          // clamp: data of a real edge, result unused
                        const int s = shfl_i(my_src, (grp << a.lpr_log2) + jj);
                        w[u] = shfl_f(my_w, (grp << a.lpr_log2) + jj);
                        const float *xp = a.state_cur + (int64_t)s * Ds + c0;
                        if (a.vec == 4) {
                            const float4 v = *reinterpret_cast<const float4 *>(xp);
                            x[u][0] = v.x; x[u][1] = v.y; x[u][2] = v.z; x[u][3] = v.w;
                        } else {
                            x[u][0] = xp[0]; x[u][1] = x[u][2] = x[u][3] = 0.0f;
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool on = j + u < cnt;
#pragma unroll
                        for (int v = 0; v < 4; ++v) {
                            const float r = __builtin_fmaf(w[u], x[u][v], acc[v]);
                            acc[v] = on ? r : acc[v];
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
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");

    // ---- B: net_state.  H^T[feature][node] = W^T . X^T on MFMA f32 32x32x2 -------------------------------------------
    const int half = lane >> 5;
    const float *xb = X + (lane & 31) * KP + half;
    f32x16 out[NTL];
    if constexpr (LAYERS == 1) {
        zero_acc<NTL>(out);
        layer_from_lds<NTL>(xb, a.Wp[0] + (size_t)lane * NTL, a.kk0, out);
    } else {
        f32x16 h1[NT];
        zero_acc<NT>(h1);
        layer_from_lds<NT>(xb, a.Wp[0] + (size_t)lane * NT, a.kk0, h1);
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            tile_epilogue<ACT, false>(h1[jt], a.bias[0], nullptr, nullptr, jt, half);
            acc_to_operand(h1[jt]);
        }
        if constexpr (LAYERS == 2) {
            zero_acc<NTL>(out);
            layer_from_regs<NT, NTL>(h1, out, a.Wp[1] + (size_t)lane * NTL);
        } else {
            f32x16 h2[NT];
            zero_acc<NT>(h2);
            layer_from_regs<NT, NT>(h1, h2, a.Wp[1] + (size_t)lane * NT);
#pragma unroll
            for (int jt = 0; jt < NT; ++jt) {
                tile_epilogue<ACT, false>(h2[jt], a.bias[1], nullptr, nullptr, jt, half);
                acc_to_operand(h2[jt]);
            }
            zero_acc<NTL>(out);
            layer_from_regs<NT, NTL>(h2, out, a.Wp[2] + (size_t)lane * NTL);
        }
    }
    // ---- C: last-layer epilogue, new state to LDS (over the aggregated-state columns, no longer needed) ---------------
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int jt = 0; jt < NTL; ++jt) {
        if (a.bn_scale) tile_epilogue<ACT, true>(out[jt], a.bias[LAYERS - 1], a.bn_scale, a.bn_shift, jt, half);
        else tile_epilogue<ACT, false>(out[jt], a.bias[LAYERS - 1], nullptr, nullptr, jt, half);
        float *x = X + (lane & 31) * KP + c_aggs;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int f = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (f < Ds) x[f] = out[jt][r];
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    // condition() for the next body: lanes 0..31 sum (new - old)^2, lanes 32..63 sum old^2, ascending feature order
    {
        const float *xo = X + (lane & 31) * KP, *xn = xo + c_aggs;
        float s = 0.0f;
        for (int f = 0; f < Ds; ++f) {
            const float o = xo[f];
            const float d = half ? o : (xn[f] - o);
            const float dd = d * d;
            s = s + dd;
        }
        const float root = __fsqrt_rn(s);
        const float nrm = shfl_f(root, (lane & 31) + 32);
        const float rhs = a.thr * nrm;
        const int moved = (half == 0) && ((lane & 31) < nvalid) && (root > rhs);
        if (__any(moved) && lane == 0) gnn_flag_raise(a.flag_out);
    }
    // coalesced store of the 32 new state rows (one contiguous block of HBM)
    {
        float *dst = a.state_nxt + i0 * Ds;
        const int total = nvalid * Ds;
        int i = lane / Ds, f = lane - i * Ds;
        for (int t = lane; t < total; t += 64) {
            dst[t] = X[i * KP + c_aggs + f];
            f += 64;
            while (f >= Ds) { f -= Ds; ++i; }
        }
    }
}

template <int LAYERS, int NT, int NTL, int ACT>
inline void launch(const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    static bool raised = false;   // dynamic LDS above 64 KiB has to be requested once per kernel
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused<LAYERS, NT, NTL, ACT>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    hipLaunchKernelGGL((k_fused<LAYERS, NT, NTL, ACT>), grid, 256, lds_bytes, st, a);
}

// (NT, NTL) pairs that are instantiated; gnn_fused.hip rounds every net up to one of them
template <int LAYERS, int ACT>
inline bool launch_tiles(int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    if constexpr (LAYERS == 1) {
        if (ntl == 1) launch<1, 1, 1, ACT>(a, grid, lds_bytes, st);
        else if (ntl == 2) launch<1, 2, 2, ACT>(a, grid, lds_bytes, st);
        else if (ntl == 4) launch<1, 4, 4, ACT>(a, grid, lds_bytes, st);
        else return false;
    } else {
        if (nt == 1 && ntl == 1) launch<LAYERS, 1, 1, ACT>(a, grid, lds_bytes, st);
        else if (nt == 2 && ntl == 2) launch<LAYERS, 2, 2, ACT>(a, grid, lds_bytes, st);
        else if (nt == 4 && ntl == 2) launch<LAYERS, 4, 2, ACT>(a, grid, lds_bytes, st);
        else if (nt == 4 && ntl == 4) launch<LAYERS, 4, 4, ACT>(a, grid, lds_bytes, st);
        else return false;
    }
    return true;
}

template <int LAYERS>
inline bool launch_act(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    switch (act) {
    case GNN_ACT_LINEAR: return launch_tiles<LAYERS, GNN_ACT_LINEAR>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_RELU: return launch_tiles<LAYERS, GNN_ACT_RELU>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_SELU: return launch_tiles<LAYERS, GNN_ACT_SELU>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_ELU: return launch_tiles<LAYERS, GNN_ACT_ELU>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_TANH: return launch_tiles<LAYERS, GNN_ACT_TANH>(nt, ntl, a, grid, lds_bytes, st);
    case GNN_ACT_SIGMOID: return launch_tiles<LAYERS, GNN_ACT_SIGMOID>(nt, ntl, a, grid, lds_bytes, st);
    default: return false;
    }
}

}   // namespace gnn_fused_dev
