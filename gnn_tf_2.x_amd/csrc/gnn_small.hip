// Persistent small-graph loop: ONE launch runs every body of a GNN.Loop (reference GNN/GNN.py:271, tf.while_loop of
// condition :202-220 and convergence :223-242) when the batch is small enough for all of its 32-node tiles to be resident at
// once (BASELINE configs[0] / [1]: a few hundred to a few thousand nodes, nets no wider than 32).  Such loops are latency
// bound: a body is one short chain of dependent loads and narrow MFMAs per tile, and as one launch per body it is mostly
// launch gap, cold caches and host gating.  Here every tile is a one-wave workgroup that keeps its row pointers in LDS, its
// weights warm in L1, and meets the other tiles at a grid barrier after each body:
//   * new state rows are stored write-through (sc1), every wave drains its stores (s_waitcnt vmcnt(0)), one lane adds to the
//     body's barrier word, polls it with L1-bypassing loads, and only then reads state rows, all of them with sc1 loads - the
//     fence-free hand-off of cdna_hip_programming.md Guideline 16 (R1) / MI355X_MICROARCH.md hand-off table, row 1;
//   * the same word carries the convergence verdict (high half: workgroups with a node that still moves), so every workgroup
//     reads the same gate with the poll it does anyway and all of them leave the loop at the same body;
//   * weights stay in registers, row pointers, label columns and the tile's own new state in LDS from body to body;
//   * every spin is bounded: on a timeout (e.g. the grid could not become resident beside another stream's work) the kernel
//     sets a status word and the host repeats the Loop with one launch per body.
// Arithmetic: the exact f32-MFMA chain of k_fused (bit-identical to oracle/gnn_oracle.c) for both fused modes; at these sizes
// the matrix work is a few microseconds either way.
#include "gnn_fused_kernel.h"

namespace gnn_fused_dev {

constexpr int GNN_SMALL_ECACHE = 1024;             // arcs of a tile whose ids / weights are kept in LDS (8 KB)

// Dense layers with the packed A operands (gnn_fused_pack, exact image: [K-step][lane][tile]) held in REGISTERS for the whole
// launch: the same v_mfma_f32_32x32x2_f32 chains as layer_from_lds / layer_from_regs with one 32-feature tile (NT == 1), i.e.
// the oracle's k-ordered fmaf chains, without a weight load per body.
template <int KK>
__device__ __forceinline__ void small_layer0(const float *xb, const float (&w)[KK], f32x16 &acc)
{
    float b[KK];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) b[kk] = xb[2 * kk];
#pragma unroll
    for (int kk = 0; kk < KK; ++kk) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[kk], b[kk], acc, 0, 0, 0);
}

template <int ACT>
__device__ __forceinline__ void small_layer(f32x16 &hin, const float *bias_prev, int half, const float (&w)[16], f32x16 &acc)
{
    tile_epilogue<ACT, false, false, true>(hin, bias_prev, nullptr, nullptr, 0, half);     // bias_prev: staged in LDS at kernel start
    acc_to_operand(hin);
#pragma unroll
    for (int ss = 0; ss < 16; ++ss) {
        const int reg = 4 * (ss >> 2) + ((ss & 3) == 1 ? 2 : (ss & 3) == 2 ? 1 : (ss & 3));       // K-step ss <-> register (acc_to_operand)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(w[ss], hin[reg], acc, 0, 0, 0);
    }
}

// KK0: K-steps of layer 0 (a multiple of 12, gnn_fused.hip make_plan); RND: entries per gather round
template <int LAYERS, int ACT, int KK0, int RND>
__global__ void __launch_bounds__(64) k_small_loop(const GnnFusedArgs a0, const GnnSmallCtl c)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x;
#ifdef GNN_DIAG      // diagnostic build only: s_memtime of workgroup 0 at every phase boundary (GNN_SMALL_STAMPS=<file>)
    int stamp_n = 0;
#define SMALL_STAMP()                                                                                   \
    do {                                                                                                \
        if (a0.stamps && blockIdx.x == 0) {                                                             \
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                 \
            const unsigned long long t_ = __builtin_amdgcn_s_memtime();                                 \
            if (lane == 0 && stamp_n < 250) a0.stamps[stamp_n] = t_;                                    \
            ++stamp_n;                                                                                  \
        }                                                                                               \
    } while (0)
#else
#define SMALL_STAMP() do { } while (0)
#endif
    SMALL_STAMP();
    const int KP = a0.KP, Ds = a0.Ds, c_aggs = a0.c_aggs, half = lane >> 5;
    float *X = lds;
    int *ipt = reinterpret_cast<int *>(lds + 32 * KP + 32);
    float *ep = lds + 32 * KP + 32 + 36;                          // last-layer bias, BatchNormalization scale / shift
    float *hb = ep + 96;                                          // biases of the hidden layers [2][32]
    float *hw = ep + 160;                                         // net_output head: W [wf * T <= 512], then b | BN scale | BN shift [3][8]
    float *scr = hw + 544;                                        // scratch [2048]: the tile's final state rows and label rows
    int *ec_src = reinterpret_cast<int *>(scr + 2048);            // the tile's arc ids / weights [GNN_SMALL_ECACHE], kept for every body
    float *ec_w = scr + 2048 + GNN_SMALL_ECACHE;
    for (int t = lane; t < 3 * 32; t += 64) {
        const int which = t >> 5, f = t & 31;
        ep[t] = which == 0 ? a0.bias[LAYERS - 1][f] : (a0.bn_scale ? (which == 1 ? a0.bn_scale[f] : a0.bn_shift[f]) : 0.0f);
    }
    if constexpr (LAYERS >= 2) {
        const int t = lane;                                       // 64 lanes = 2 x 32 features
        if (t < 32 * (LAYERS - 1)) hb[t] = a0.bias[t >> 5][t & 31];
    }
    if (c.out) {
        const int nw = (a0.Ds + c.NLc) * c.T;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (lane + 64 * u < nw) hw[lane + 64 * u] = c.ow[lane + 64 * u];
        if (lane < 24) {
            const int which = lane >> 3, q = lane & 7;
            float v = which == 1 ? 1.0f : 0.0f;
            if (q < c.T) v = which == 0 ? c.ob[q] : (c.obn_scale ? (which == 1 ? c.obn_scale[q] : c.obn_shift[q]) : v);
            hw[512 + lane] = v;
        }
    }
    const int64_t i0 = (int64_t)blockIdx.x * 32;
    const int nvalid = (int)((a0.n_rows - i0) < 32 ? (a0.n_rows - i0) : 32);
    {   // the tile's row pointers: read once, kept in LDS for every body
        const int my_ip = (lane <= nvalid) ? gload1(a0.indptr + i0 + lane) : 0;
        const int last_ip = shfl_i(my_ip, nvalid);
        if (lane <= 32) ipt[lane] = lane <= nvalid ? my_ip : last_ip;
    }
    // the tile's arcs (contiguous CSR entries of its 32 rows): ids and weights once into LDS when they fit - every body's gather then
    // needs one memory round trip per round (the neighbour rows) instead of two
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    const int e_base = ipt[0], e_cnt = ipt[32] - e_base;
    const bool ecached = e_cnt <= c.ecache;
    if (ecached)
        for (int t = lane; t < e_cnt; t += 64) { ec_src[t] = gload1(a0.adj_src + e_base + t); ec_w[t] = gload1(a0.adj_w + e_base + t); }
    // weights: once, into registers
    float w0[KK0], w1[16], w2[16];
#pragma unroll
    for (int kk = 0; kk < KK0; ++kk) w0[kk] = gload1(a0.Wp[0] + (size_t)kk * 64 + lane);
    if constexpr (LAYERS >= 2) {
#pragma unroll
        for (int ss = 0; ss < 16; ++ss) w1[ss] = gload1(a0.Wp[1] + (size_t)ss * 64 + lane);
    }
    if constexpr (LAYERS >= 3) {
#pragma unroll
        for (int ss = 0; ss < 16; ++ss) w2[ss] = gload1(a0.Wp[2] + (size_t)ss * 64 + lane);
    }
    // the gate words of the NEXT run (the other half of the double buffer): nobody reads them during this launch
    if (blockIdx.x == 0)
        for (int t = lane; t < c.n_words; t += 64) c.zero_words[t] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    SMALL_STAMP();                                                   // 1: set-up loads issued (weights, row pointers)
    const unsigned n_wg = gridDim.x;
    // Grid barrier + gate in ONE word per body: after its write-through stores have drained, every workgroup adds
    // 1 (+ 0x10000 when one of its nodes still moves) to word[b]; the word is complete when its low half reaches the number of
    // workgroups, and body b runs iff its high half is non-zero (GNN.py:218-220: reduce_any over all nodes).  One atomic and one
    // bounded poll per body.  Returns 1 = run body b, 0 = converged, -1 = gave up.
    auto arrive_and_gate = [&](int b, int moved) -> int {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned seen = 0;
        if (lane == 0) {
            GNN_GLOBAL unsigned *word = (GNN_GLOBAL unsigned *)(c.flags + b);
            __hip_atomic_fetch_add(word, 1u + (moved ? 0x10000u : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (unsigned spins = 0;; ++spins) {
                seen = __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((seen & 0xffffu) >= n_wg) break;
                if (spins > (1u << 22)) { seen = 0xffffffffu; break; }       // give up: the host falls back to per-body launches
                __builtin_amdgcn_s_sleep(1);
            }
        }
        seen = (unsigned)__builtin_amdgcn_readfirstlane((int)seen);
        if (seen == 0xffffffffu) {
            // STICKY failure: the kernel only ever SETS the status words (the host clears them before the launch, gnn_small_run).  A
            // workgroup that gives up has already added itself to the barrier word, so a late arrival can still complete that barrier
            // for the others; if it is the last one they finish normally - and must not overwrite this 1 with a 0.
            if (lane == 0) {
                __hip_atomic_store((GNN_GLOBAL int *)c.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (c.host_result) c.host_result[1] = 1;
            }
            return -1;
        }
        return (seen >> 16) ? 1 : 0;
    };
    // ---- state <- initial state (GNN.py:262 / :265) for the owned rows, first condition against ones (GNN.py:266, :271) in the
    // oracle's order (k_check: ascending feature, unfused, one lane per row) ----------------------------------------------------
    int go;
    {
        const float *init = c.init + i0 * Ds;
        float *own0 = c.state0 + (a0.row_begin + i0) * Ds;
        const int total = nvalid * Ds;                          // <= 32 x 32: sixteen loads per lane at most, all in flight at once
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = (lane + 64 * u < total) ? gload1(init + lane + 64 * u) : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lane + 64 * u < total) { sstore1<true>(own0 + lane + 64 * u, v[u]); X[lane + 64 * u] = v[u]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int moved = 0;
        if (lane < nvalid) {
            float dist = 0.0f, nrm = 0.0f;
            for (int f = 0; f < Ds; ++f) {                      // the rows just staged in LDS (the tile itself is built by body 0)
                const float df = X[lane * Ds + f] - 1.0f;
                const float dd = df * df;
                dist = dist + dd;
                nrm = nrm + 1.0f;
            }
            moved = sqrtf(dist) > a0.thr * sqrtf(nrm);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // 2: initial state copied, first condition
        go = arrive_and_gate(0, __any(moved));
        SMALL_STAMP();                                               // 3: gate 0
    }
    int k = 0;
    for (; k < c.max_iter && go == 1; ++k) {
        GnnFusedArgs a = a0;
        // body 0 gathers from the read-only initial state (no tile has to wait for the others' copies of it)
        a.state_cur = k == 0 ? c.init - a0.row_begin * Ds : ((k & 1) ? c.state1 : c.state0);
        a.state_nxt = ((k & 1) ? c.state0 : c.state1) + a0.row_begin * Ds;
        load_tile_generic<true, RND>(a, X, ipt, i0, lane, nvalid, KP, c_aggs, k > 0, ecached ? ec_src : nullptr, ecached ? ec_w : nullptr, e_base);     // k > 0: the tile skeleton is still in LDS
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // body + 0: tile loaded (gather)
        const float *xb = X + (lane & 31) * KP + half;
        f32x16 out;
        if constexpr (LAYERS == 1) {
            out = f32x16{};
            small_layer0<KK0>(xb, w0, out);
        } else {
            f32x16 h1 = {};
            small_layer0<KK0>(xb, w0, h1);
            if constexpr (LAYERS == 2) {
                out = f32x16{};
                small_layer<ACT>(h1, hb, half, w1, out);
            } else {
                f32x16 h2 = {};
                small_layer<ACT>(h1, hb, half, w1, h2);
                out = f32x16{};
                small_layer<ACT>(h2, hb + 32, half, w2, out);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (a.bn_scale) tile_epilogue<ACT, true, false, true>(out, ep, ep + 32, ep + 64, 0, half);
        else tile_epilogue<ACT, false, false, true>(out, ep, nullptr, nullptr, 0, half);
        {
            float *x = X + (lane & 31) * KP + c_aggs;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = (r & 3) + 8 * (r >> 2) + 4 * half;
                if (f < Ds) x[f] = out[r];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // body + 1: dense layers, new state in LDS
        int moved = 0;
        check_store_generic<true>(a, X, i0, lane, nvalid, KP, c_aggs, &moved);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // body + 2: condition, row stores drained
        go = arrive_and_gate(k + 1, moved);
        SMALL_STAMP();                                               // body + 3: barrier + gate
    }
    if (go < 0) return;                      // (status words already set, see arrive_and_gate; this tile's output rows stay stale: the host repeats the Loop)
    if (blockIdx.x == 0 && lane == 0) {      // executed bodies (GNN.py:267; every workgroup agrees).  The status words are NOT touched here.
        c.kfinal[0] = k;
        if (c.host_result) c.host_result[0] = k;
    }
    // ---- apply_filters + one-layer net_output on the tile's masked rows (GNN.py:275-279), arithmetic as k_out1: k-ordered fmaf
    // chain per output, bias, softmax / activation, BatchNormalization ------------------------------------------------------
    // The tile's final state rows (its own write-through stores) and label rows are contiguous in memory: all lanes copy them into
    // LDS with every load in flight at once (one round trip, not one per k-step); the head's weights were staged at kernel start.
    if (c.out) {
        const int wf = Ds + c.NLc, T = c.T, NL = c.NL, NLc = c.NLc;
        const float *sfin = ((k & 1) ? c.state1 : c.state0) + (a0.row_begin + i0) * Ds;
        const float *nod = c.nodes_own + i0 * NL;
        const int ns = nvalid * Ds, nl = NLc ? nvalid * NL : 0;          // <= 1024 each (Ds, NL <= 32)
        float sv[16], lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            sv[u] = (lane + 64 * u < ns) ? sload1<true>(sfin + lane + 64 * u) : 0.0f;
            lv[u] = (lane + 64 * u < nl) ? gload1(nod + lane + 64 * u) : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (lane + 64 * u < ns) scr[lane + 64 * u] = sv[u];
            if (lane + 64 * u < nl) scr[1024 + lane + 64 * u] = lv[u];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        const bool on = lane < nvalid && c.mask[i0 + (lane < nvalid ? lane : 0)];
        if (on) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = 0.0f;
            for (int kk = 0; kk < wf; ++kk) {                            // k-ordered fmaf chain per output, as k_out1
                const float x = kk < Ds ? scr[lane * Ds + kk] : scr[1024 + lane * NL + (kk - Ds)];
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (j < T) y[j] = __builtin_fmaf(x, hw[kk * T + j], y[j]);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < T) y[j] = y[j] + hw[512 + j];
            float v[8];
            if (c.oact == GNN_ACT_SOFTMAX) {
                float mx = y[0];
#pragma unroll
                for (int q = 1; q < 8; ++q)
                    if (q < T) mx = y[q] > mx ? y[q] : mx;
                float sum = 0.0f;
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < T) { v[q] = gnn_expf(y[q] - mx); sum = sum + v[q]; }
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < T) v[q] = __fdiv_rn(v[q], sum);
            } else {
#pragma unroll
                for (int q = 0; q < 8; ++q)
                    if (q < T) v[q] = gnn_act(y[q], c.oact);
            }
            float *o = c.out + (int64_t)c.mask_pos[i0 + lane] * T;
#pragma unroll
            for (int q = 0; q < 8; ++q)
                if (q < T) {
                    float r = v[q];
                    if (c.obn_scale) { const float t2 = r * hw[520 + q]; r = t2 + hw[528 + q]; }
                    sstore1<true>(o + q, r);                         // write-through: workgroup 0 may read it below (graph readout)
                }
        }
    }
    SMALL_STAMP();                                                   // last: output stage
    // ---- graph readout (GNN.py:331-332): out_graph[g, t] = sum over the (node, w) of graph g, ascending, fmaf(w, out[node, t]) - the
    // arithmetic of k_readout - by workgroup 0 after one more grid barrier; the result goes straight to pinned host memory -------------
    if (c.ng_ip) {
        if (arrive_and_gate(c.ro_word, 0) < 0) return;
        if (blockIdx.x == 0)
            for (int t = lane; t < c.G * c.T; t += 64) {
                const int gi = t / c.T, ci = t - gi * c.T;
                float acc = 0.0f;
                for (int e = gload1(c.ng_ip + gi); e < gload1(c.ng_ip + gi + 1); ++e)
                    acc = __builtin_fmaf(gload1(c.ng_w + e), sload1<true>(c.out + (int64_t)gload1(c.ng_node + e) * c.T + ci), acc);
                c.ng_host[t] = acc;
            }
    }
}

template <int LAYERS, int ACT>
static bool small_launch_k(int kk0, int rnd, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st)
{
#define GNN_SMALL_K(K)                                                                                                      \
    if (kk0 == K) {                                                                                                         \
        if (rnd == 8) hipLaunchKernelGGL((k_small_loop<LAYERS, ACT, K, 8>), grid, 64, lds_bytes, st, a, c);                   \
        else hipLaunchKernelGGL((k_small_loop<LAYERS, ACT, K, 4>), grid, 64, lds_bytes, st, a, c);                            \
        return true;                                                                                                        \
    }
    GNN_SMALL_K(12) GNN_SMALL_K(24) GNN_SMALL_K(36) GNN_SMALL_K(48)
#undef GNN_SMALL_K
    return false;
}

template <int LAYERS>
static bool small_launch_act(int act, int kk0, int rnd, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st)
{
#define GNN_SMALL_CASE(A) case A: return small_launch_k<LAYERS, A>(kk0, rnd, a, c, grid, lds_bytes, st);
    switch (act) {
        GNN_SMALL_CASE(GNN_ACT_LINEAR) GNN_SMALL_CASE(GNN_ACT_RELU) GNN_SMALL_CASE(GNN_ACT_SELU) GNN_SMALL_CASE(GNN_ACT_ELU)
        GNN_SMALL_CASE(GNN_ACT_TANH) GNN_SMALL_CASE(GNN_ACT_SIGMOID)
    default: return false;
    }
#undef GNN_SMALL_CASE
}

}   // namespace gnn_fused_dev

bool gnn_small_launch(int layers, int act, int kk0, int rnd, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes,
                      hipStream_t st)
{
    using namespace gnn_fused_dev;
    if (layers == 1) return small_launch_act<1>(act, kk0, rnd, a, c, grid, lds_bytes, st);
    if (layers == 2) return small_launch_act<2>(act, kk0, rnd, a, c, grid, lds_bytes, st);
    if (layers == 3) return small_launch_act<3>(act, kk0, rnd, a, c, grid, lds_bytes, st);
    return false;
}
