// EXPERIMENT, NOT BUILT, NOT SHIPPED (round 2).  Result on MI355X, BASELINE metric configuration (1 M nodes, 10 M arcs), tools/sweep.sh, DIAG build, same box:
//   k_fused (shipped)                                   0.700 ms / launch
//   this kernel (producer / consumer pairs per SIMD)    0.776 ms   correct: the 125 parity / default-path / sharded tests pass with it
//   ... producer with two gather batches in flight      0.786 ms
//   ... without the raised producer priority            0.793 ms
//   ... consumers alone (no gather: timing only)        0.535 ms   = what the dense layers of one wave per SIMD take, 35 k cycles per tile
// The dense phase alone would carry the kernel to 0.74 of the HBM roof, but the four producers of a CU do not deliver a tile per
// 35 k cycles next to the consumers' weight stream (the same L2 / TA path): the pair runs at a tile per 50 k cycles, k_fused's two
// symmetric waves per SIMD at a tile per 46 k.  Kept as the record of the attempt; to try it again: add the file to
// SRC, declare gnn_fused_launch_pair3 in gnn_fused.h and call it from gnn_fused_iteration in place of the full-tile launch.
//
// Producer / consumer form of the fused iteration kernel for the BASELINE shape (state width 64, split arithmetic, two output
// tiles): one launch = one iteration of GNN.Loop (reference GNN/GNN.py:223-242 + :202-220), same arithmetic, same LDS tile, same
// device functions as k_fused<.., SPLIT = true, FULL = true> (gnn_fused_kernel.h) - only WHO does what differs.
//
// k_fused: each of the eight waves of a CU does everything for its tile (gather, three dense layers, epilogue), one phase after
// the other; two waves share a SIMD and a start-up spread makes it LIKELY that one of them waits for HBM while the other computes.
// Here that pairing is made explicit.  Waves w and w + 4 share SIMD w: wave w + 4 is the PRODUCER of the pair - it draws the
// tickets, requests row pointers / ids / neighbour rows and builds the 32-node tile [own state | labels | aggregated states | ...]
// in one of the pair's two LDS buffers; wave w is the CONSUMER - dense layers on the bf16 MFMA, epilogue, condition, row stores
// from the other buffer.  The two never run the same phase, a buffer is handed over through one LDS word:
//     full[b] == 0      the producer may fill buffer b
//     full[b] == t + 1  tile t is complete in buffer b (the consumer may read it); -1: no more tiles
// LDS: the same eight tile buffers as k_fused (two per pair), the flags live in the consumer's unused row-pointer slots.
#include "gnn_fused_kernel.h"

namespace gnn_fused_dev {

__device__ __forceinline__ int lds_flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
// the data written / read before the hand-over are LDS operations of this wave: they are complete once lgkmcnt is zero
__device__ __forceinline__ void lds_flag_store(int *p, int v)
{
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

template <int LAYERS, int NT, int ACT>
__global__ void __launch_bounds__(GNN_FUSED_THREADS, 2) k_fused_pair(const GnnFusedArgs a0)
{
    constexpr int NTL = 2;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if (!gnn_gate_open(a0.gate, a0.world)) return;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = a0.KP, c_aggs = a0.c_aggs;
    const int pair = wave & 3;
    const bool producer = wave >= 4;
    float *ep = lds + (size_t)GNN_FUSED_WAVES * 32 * KP + 32 + GNN_FUSED_WAVES * 36;
    for (int t = threadIdx.x; t < 3 * 32 * NTL; t += GNN_FUSED_THREADS) {
        const int which = t / (32 * NTL), f = t - which * 32 * NTL;
        ep[t] = which == 0 ? a0.bias[LAYERS - 1][f] : (a0.bn_scale ? (which == 1 ? a0.bn_scale[f] : a0.bn_shift[f]) : 0.0f);
    }
    float *hb = ep + 3 * 32 * NTL;
    if constexpr (LAYERS > 1)
        for (int t = threadIdx.x; t < (LAYERS - 1) * 32 * NT; t += GNN_FUSED_THREADS)
            hb[t] = a0.bias[t / (32 * NT)][t % (32 * NT)] * (ACT == GNN_ACT_SELU ? 1.44269504088896341f : 1.0f);
    int *slots = reinterpret_cast<int *>(lds + (size_t)GNN_FUSED_WAVES * 32 * KP + 32);
    int *full = slots + pair * 36;                    // [2]: the consumer's row-pointer slot is free (consumers read no CSR)
    int *ipt = slots + wave * 36;                     // producer: its own slot
    float *Xb[2] = {lds + (size_t)(2 * pair) * 32 * KP, lds + (size_t)(2 * pair + 1) * 32 * KP};
    if (!producer && lane < 2) full[lane] = 0;
    if (producer) { zero_pad_columns(a0, Xb[0], lane, KP); zero_pad_columns(a0, Xb[1], lane, KP); }
    __syncthreads();

    if (producer) {
        // ---- tickets, row pointers, ids, neighbour rows: the tile into the pair's free buffer -------------------------------------
        int tile = 0, next_tile = 0;
        if (lane == 0) { tile = atomicAdd(a0.tile_ctr, 1); next_tile = atomicAdd(a0.tile_ctr, 1); }
        tile = __builtin_amdgcn_readfirstlane(tile) + a0.tile_base;
        next_tile = __builtin_amdgcn_readfirstlane(next_tile) + a0.tile_base;
        int ip_cur = tile_rowptr_clamp(a0, tile, lane, tile_rowptr_request(a0, tile, lane));
        int src_cur = 0;
        float w_cur = 0.0f;
        tile_first_ids(a0, ip_cur, lane, src_cur, w_cur);
        if (!(a0.variant & 1024)) __builtin_amdgcn_s_setprio(3);                // few instructions, long waits: issue ahead of the partner's dense stream
        int b = 0;
        for (;;) {
            const int64_t i0 = (int64_t)tile * 32;
            if (i0 >= a0.n_rows) break;
            GnnFusedArgs a = a0;
            asm volatile("" : "+s"(a.state_cur), "+s"(a.inv), "+s"(a.adj_src), "+s"(a.adj_w));
            while (lds_flag_load(full + b) != 0) __builtin_amdgcn_s_sleep(2);
            if (lane <= 32) ipt[lane] = ip_cur;
            const int ip_next_raw = tile_rowptr_request(a, next_tile, lane);
#ifdef GNN_DIAG
            if (a0.variant & 256) { }                                  // timing experiment: no gather (results meaningless)
            else if (a0.variant & 128) load_tile_fast64<true, true, true>(a, Xb[b], ipt, i0, lane, KP, c_aggs, src_cur, w_cur);
            else
#endif
            load_tile_fast64<true, true, false>(a, Xb[b], ipt, i0, lane, KP, c_aggs, src_cur, w_cur);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            int next2_tile = 0;
            if (lane == 0) next2_tile = atomicAdd(a0.tile_ctr, 1);
            const int ip_next = tile_rowptr_clamp(a, next_tile, lane, ip_next_raw);
            int src_next = 0;
            float w_next = 0.0f;
            tile_first_ids(a, ip_next, lane, src_next, w_next);
            if (lane == 0) lds_flag_store(full + b, tile + 1);
            b ^= 1;
            tile = next_tile;
            next_tile = __builtin_amdgcn_readfirstlane(next2_tile) + a0.tile_base;
            ip_cur = ip_next; src_cur = src_next; w_cur = w_next;
        }
        while (lds_flag_load(full + b) != 0) __builtin_amdgcn_s_sleep(2);
        if (lane == 0) lds_flag_store(full + b, -1);
        return;
    }

    // ---- consumer: net_state on the tile, condition, row stores -----------------------------------------------------------------
    const int half = lane >> 5;
    int b = 0;
    for (;;) {
        int t;
        while ((t = lds_flag_load(full + b)) == 0) __builtin_amdgcn_s_sleep(1);
        t = __builtin_amdgcn_readfirstlane(t);
        if (t < 0) break;
        const int64_t i0 = (int64_t)(t - 1) * 32;
        const int nvalid = (int)((a0.n_rows - i0) < 32 ? (a0.n_rows - i0) : 32);
        GnnFusedArgs a = a0;
        asm volatile("" : "+s"(a.bn_scale), "+s"(a.bn_shift), "+s"(a.state_nxt), "+s"(a.Ws_base));
        float *X = Xb[b];
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#ifdef GNN_DIAG
        if (a0.variant & 512) {                                        // timing experiment: no dense layers (results meaningless)
            if (lane == 0) lds_flag_store(full + b, 0);
            b ^= 1;
            continue;
        }
#endif
        const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int *>(a.Ws_base), 0, a.ws_bytes, 0x00020000);
        const int wv = lane * 16;
        const float *xr = X + (lane & 31) * KP + 8 * half;
        f32x16 out[NTL];
        if constexpr (LAYERS == 1) {
            layer0_split<NTL, false>(xr, wrs, wv, a.ws_off[0], a.chunks0, out, ep, half);
        } else {
            f32x16 h1[NT];
            layer0_split<NT, true>(xr, wrs, wv, a.ws_off[0], a.chunks0, h1, hb, half);
            if constexpr (LAYERS == 2) {
                layer_split_from_regs<NT, NTL, ACT>(h1, ep, half, out, wrs, wv, a.ws_off[1]);
            } else {
                f32x16 h2[NT];
                layer_split_from_regs<NT, NT, ACT>(h1, hb + 32 * NT, half, h2, wrs, wv, a.ws_off[1]);
                layer_split_from_regs<NT, NTL, ACT>(h2, ep, half, out, wrs, wv, a.ws_off[2]);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
        for (int jt = 0; jt < NTL; ++jt) {
            if (a.bn_scale) tile_epilogue<ACT, true, true, true, true>(out[jt], ep, ep + 32 * NTL, ep + 64 * NTL, jt, half);
            else tile_epilogue<ACT, false, true, true, true>(out[jt], ep, nullptr, nullptr, jt, half);
        }
        if (nvalid == 32) { GnnFlagPeek pk = {0, 0, 0}; if (lane == 0) pk = gnn_flag_peek(a.flag_out); finish_fast64_aligned(a, X, out, i0, lane, KP, c_aggs, pk); }
        else finish_fast64_partial(a, X, out, i0, lane, KP, c_aggs, nvalid);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (lane == 0) lds_flag_store(full + b, 0);          // every LDS read of the buffer has completed (lgkmcnt 0); the row stores fly on
        b ^= 1;
    }
}

template <int LAYERS, int NT, int ACT>
static void launch_pair(const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev < 0 || dev >= 64 || !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused_pair<LAYERS, NT, ACT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (dev >= 0 && dev < 64) raised[dev] = true;
    }
    hipLaunchKernelGGL((k_fused_pair<LAYERS, NT, ACT>), grid, GNN_FUSED_THREADS, lds_bytes, st, a);
}

}   // namespace gnn_fused_dev

// net_state with 3 Dense layers, hidden width <= 128 (NT = 4), state width 64 (NTL = 2), split arithmetic
bool gnn_fused_launch_pair3(int act, int nt, int ntl, const GnnFusedArgs &a, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    using namespace gnn_fused_dev;
    if (nt != 4 || ntl != 2) return false;
    switch (act) {
    case GNN_ACT_LINEAR: launch_pair<3, 4, GNN_ACT_LINEAR>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_RELU: launch_pair<3, 4, GNN_ACT_RELU>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_SELU: launch_pair<3, 4, GNN_ACT_SELU>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_ELU: launch_pair<3, 4, GNN_ACT_ELU>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_TANH: launch_pair<3, 4, GNN_ACT_TANH>(a, grid, lds_bytes, st); return true;
    case GNN_ACT_SIGMOID: launch_pair<3, 4, GNN_ACT_SIGMOID>(a, grid, lds_bytes, st); return true;
    default: return false;
    }
}
