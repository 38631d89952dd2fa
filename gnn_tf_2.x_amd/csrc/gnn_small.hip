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

typedef unsigned v4u __attribute__((ext_vector_type(4)));
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


// ---- padded exchange rows ------------------------------------------------------------------------------------------------------
// Between bodies the state travels through a buffer of its own, xs[2][tiles * 32][DP] with DP = 16 or 32 floats per row (a
// 64- or 128-byte line piece per node), not through the [N, Ds] state replicas: a tile publishes its 32 new rows as ONE contiguous
// block of 16-byte write-through stores (2 or 4 store instructions instead of Ds scalar ones - a 4-byte sc1 store is a fabric write of
// its own, MI355X_MICROARCH.md "stores of each flavour") and a neighbour row is fetched with 16-byte sc1 loads (2 per lane and arc for
// Ds <= 16 instead of Ds / 2 scalar ones).  The [N, Ds] replicas get the initial and the final state only.
// HW: floats per half-wave lane (DP / 2); PR: arcs per round.  The fmaf chain per column runs over the arcs in stored order, as before.
template <int HW, int PR>
__device__ __forceinline__ void small_gather_padded(__amdgpu_buffer_rsrc_t rs, float *X, const int *ipt, int lane, int nvalid, int KP, int c_aggs,
                                                    int Ds, const int *adj_src, const float *adj_w, const int *ec_src, const float *ec_w, int ec_base)
{
    const int node = lane & 31, hf = lane >> 5;
    const int beg = ipt[node], end = ipt[node + 1];
    float acc[HW];
#pragma unroll
    for (int c = 0; c < HW; ++c) acc[c] = 0.0f;
    for (int e = beg; e < end; e += PR) {
        float w[PR];
        int off[PR];
#pragma unroll
        for (int u = 0; u < PR; ++u) {
            const int ee = e + u < end ? e + u : e;            // clamp: a real entry, result unused
            w[u] = ec_w ? ec_w[ee - ec_base] : gload1(adj_w + ee);
            const int src = ec_src ? ec_src[ee - ec_base] : gload1(adj_src + ee);
            off[u] = (src * (2 * HW) + hf * HW) * 4;
        }
        v4f x[PR][HW / 4];
#pragma unroll
        for (int u = 0; u < PR; ++u)
#pragma unroll
            for (int j = 0; j < HW / 4; ++j) x[u][j] = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, off[u] + 16 * j, 0, 16));    // aux 16 = sc1
#pragma unroll
        for (int u = 0; u < PR; ++u)
            if (e + u < end) {
#pragma unroll
                for (int j = 0; j < HW / 4; ++j) {
                    acc[4 * j] = __builtin_fmaf(w[u], x[u][j].x, acc[4 * j]);
                    acc[4 * j + 1] = __builtin_fmaf(w[u], x[u][j].y, acc[4 * j + 1]);
                    acc[4 * j + 2] = __builtin_fmaf(w[u], x[u][j].z, acc[4 * j + 2]);
                    acc[4 * j + 3] = __builtin_fmaf(w[u], x[u][j].w, acc[4 * j + 3]);
                }
            }
    }
    if (node < nvalid) {
        float *x = X + node * KP + c_aggs + hf * HW;
#pragma unroll
        for (int c = 0; c < HW; ++c)
            if (hf * HW + c < Ds) x[c] = acc[c];
    }
}

// the tile's 32 rows -> its block of the padded buffer.  Source element (row, col) at src[row * rs_ + col]; rows >= nrows and columns
// >= Ds are stored as zeros (never read back into a result: a gather only keeps columns < Ds of rows that exist)
template <int DP>
__device__ __forceinline__ void small_store_padded(__amdgpu_buffer_rsrc_t rs, int64_t i0, const float *src, int rs_, int nrows, int Ds, int lane)
{
    constexpr int QR = DP / 4;                       // 16-byte pieces per row
#pragma unroll
    for (int u = 0; u < (32 * QR) / 64; ++u) {
        const int q = lane + 64 * u, row = q / QR, c4 = (q % QR) * 4;
        const float *x = src + row * rs_ + c4;
        const bool rok = row < nrows;
        v4f v;
        v.x = (rok && c4 < Ds) ? x[0] : 0.0f;
        v.y = (rok && c4 + 1 < Ds) ? x[1] : 0.0f;
        v.z = (rok && c4 + 2 < Ds) ? x[2] : 0.0f;
        v.w = (rok && c4 + 3 < Ds) ? x[3] : 0.0f;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs, (int)(i0 * DP + 4 * q) * 4, 0, 16);       // aux 16 = sc1: write-through
    }
}

// KK0: K-steps of layer 0 kept in registers (a multiple of 8 that covers the concat width); RND: arcs per gather round for rows of <= 16 floats
template <int LAYERS, int ACT, int KK0>
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
#ifdef GNN_DIAG      // GNN_POISON=1: NaN over the whole LDS allocation before anything is staged (one-wave workgroup: program order is enough)
    if (a0.lds_floats) {
        for (int t = lane; t < a0.lds_floats; t += 64) lds[t] = __builtin_nanf("");
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    }
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
    // Everything the launch reads from read-only memory is requested HERE, at once (one round trip for all of it): row pointers, the
    // tile's initial rows, and what the output stage needs at the very end (label rows, mask, output position)
    const int my_ip = (lane <= nvalid) ? gload1(a0.indptr + i0 + lane) : 0;
    const int Ds0 = a0.Ds;
    float v_init[16];
    {
        const float *init = c.init + i0 * Ds0;
#pragma unroll
        for (int u = 0; u < 16; ++u) v_init[u] = (lane + 64 * u < nvalid * Ds0) ? gload1(init + lane + 64 * u) : 0.0f;
    }
    bool out_on = false;
    int out_pos = 0;
    if (c.out) {
        const int nl = c.NLc ? nvalid * c.NL : 0;                   // <= 1024 (NL <= 32)
        const float *nod = c.nodes_own + i0 * c.NL;
        float lv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) lv[u] = (lane + 64 * u < nl) ? gload1(nod + lane + 64 * u) : 0.0f;
        out_on = lane < nvalid && c.mask[i0 + (lane < nvalid ? lane : 0)];
        out_pos = out_on ? c.mask_pos[i0 + lane] : 0;
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lane + 64 * u < nl) scr[1024 + lane + 64 * u] = lv[u];      // (scr is not touched again before the output stage)
    }
    {   // the tile's row pointers: kept in LDS for every body
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
    // the two padded exchange buffers (see small_gather_padded), as buffer resources: 16-byte sc1 loads / stores
    const int xs_bytes = (int)gridDim.x * 32 * c.DP * 4;
    const __amdgpu_buffer_rsrc_t xs_rs[2] = {__builtin_amdgcn_make_buffer_rsrc(c.xs, 0, xs_bytes, 0x00020000),
                                             __builtin_amdgcn_make_buffer_rsrc(c.xs + (size_t)gridDim.x * 32 * c.DP, 0, xs_bytes, 0x00020000)};
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
            // STICKY failure: the kernel only ever SETS the status word (pinned host memory; the host clears it before the launch, gnn_small_run).  A
            // workgroup that gives up has already added itself to the barrier word, so a late arrival can still complete that barrier
            // for the others; if it is the last one they finish normally - and must not overwrite this 1 with a 0.
            if (lane == 0) c.host_result[1] = 1;
            return -1;
        }
        return (seen >> 16) ? 1 : 0;
    };
    // ---- state <- initial state (GNN.py:262 / :265) for the owned rows, first condition against ones (GNN.py:266, :271) in the
    // oracle's order (k_check: ascending feature, unfused, one lane per row) ----------------------------------------------------
    int go;
    {
        float *own0 = c.state0 + (a0.row_begin + i0) * Ds;
        const int total = nvalid * Ds;                          // <= 32 x 32: sixteen values per lane at most (requested at kernel start)
#pragma unroll
        for (int u = 0; u < 16; ++u)
            if (lane + 64 * u < total) { *gptr_w(own0 + lane + 64 * u) = v_init[u]; X[lane + 64 * u] = v_init[u]; }      // replica 0: read by nobody in this launch (k == 0: the final state)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (c.DP == 16) small_store_padded<16>(xs_rs[0], i0, X, Ds, nvalid, Ds, lane);      // what body 0 gathers from, behind gate 0
        else small_store_padded<32>(xs_rs[0], i0, X, Ds, nvalid, Ds, lane);
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
        a.state_cur = c.init - a0.row_begin * Ds;                    // (body 0 builds the tile skeleton: own rows from the read-only initial state)
        a.state_nxt = nullptr;
        load_tile_generic<false, 4>(a, X, ipt, i0, lane, nvalid, KP, c_aggs, k > 0, nullptr, nullptr, 0, true);     // k > 0: the tile skeleton is still in LDS; no gather here
        {
            const __amdgpu_buffer_rsrc_t rs = xs_rs[k & 1];
            const int *es = ecached ? ec_src : nullptr;
            const float *ew = ecached ? ec_w : nullptr;
            if (c.DP == 32) small_gather_padded<16, 4>(rs, X, ipt, lane, nvalid, KP, c_aggs, Ds, a0.adj_src, a0.adj_w, es, ew, e_base);
            else if (c.rnd == 8) small_gather_padded<8, 8>(rs, X, ipt, lane, nvalid, KP, c_aggs, Ds, a0.adj_src, a0.adj_w, es, ew, e_base);
            else small_gather_padded<8, 4>(rs, X, ipt, lane, nvalid, KP, c_aggs, Ds, a0.adj_src, a0.adj_w, es, ew, e_base);
        }
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
        if (a.bn_scale) tile_epilogue<ACT, true, false, true>(out, ep, ep + 32, ep + 64, 0, half, Ds);      // features >= Ds: padding of the tile
        else tile_epilogue<ACT, false, false, true>(out, ep, nullptr, nullptr, 0, half, Ds);
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
        // the new rows first (they drain while the condition is evaluated), then the condition
        if (c.DP == 16) small_store_padded<16>(xs_rs[(k & 1) ^ 1], i0, X + c_aggs, KP, 32, Ds, lane);
        else small_store_padded<32>(xs_rs[(k & 1) ^ 1], i0, X + c_aggs, KP, 32, Ds, lane);
        check_store_generic<false, false>(a, X, i0, lane, nvalid, KP, c_aggs, &moved);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // body + 2: condition, row stores drained
        go = arrive_and_gate(k + 1, moved);
        SMALL_STAMP();                                               // body + 3: barrier + gate
    }
    if (go < 0) return;                      // (status words already set, see arrive_and_gate; this tile's output rows stay stale: the host repeats the Loop)
    // the final state of the tile's rows into the [N, Ds] replica the host expects it in (k & 1); k == 0: replica 0 holds it already
    if (k > 0) {
        float *dst = ((k & 1) ? c.state1 : c.state0) + (a0.row_begin + i0) * Ds;
        const int total = nvalid * Ds;
        RowCol rc(lane, Ds);
        for (int t = lane; t < total; t += 64, rc.next()) *gptr_w(dst + t) = X[rc.i * KP + c_aggs + rc.c];
    }
    if (blockIdx.x == 0 && lane == 0) {      // executed bodies (GNN.py:267; every workgroup agrees).  The status words are NOT touched here.
        c.kfinal[0] = k;
        c.host_result[0] = k;
    }
    // ---- apply_filters + one-layer net_output on the tile's masked rows (GNN.py:275-279), arithmetic as k_out1: k-ordered fmaf
    // chain per output, bias, softmax / activation, BatchNormalization ------------------------------------------------------
    // The tile's final state rows (its own write-through stores) and label rows are contiguous in memory: all lanes copy them into
    // LDS with every load in flight at once (one round trip, not one per k-step); the head's weights were staged at kernel start.
    if (c.out) {
        const int wf = Ds + c.NLc, T = c.T, NL = c.NL;
        const int ns = nvalid * Ds;                                       // <= 1024 (Ds <= 32)
        {   // the tile's final state rows are still in LDS: the new-state columns of the last body, or (k == 0) the staged initial rows;
            // its label rows were staged at kernel start
            RowCol rc(lane, Ds);
            for (int t = lane; t < ns; t += 64, rc.next()) scr[t] = k > 0 ? X[rc.i * KP + c_aggs + rc.c] : X[t];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (out_on) {
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
            float *o = c.out + (int64_t)out_pos * T;
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
        if (blockIdx.x == 0) small_graph_readout(c, lane);
    }
}

template <int LAYERS, int ACT>
static bool small_launch_k(int kk0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st)
{
#define GNN_SMALL_K(K)                                                                                                      \
    if (kk0 == K) {                                                                                                         \
        hipLaunchKernelGGL((k_small_loop<LAYERS, ACT, K>), grid, 64, lds_bytes, st, a, c);                                    \
        return true;                                                                                                        \
    }
    GNN_SMALL_K(8) GNN_SMALL_K(12) GNN_SMALL_K(16) GNN_SMALL_K(24) GNN_SMALL_K(32) GNN_SMALL_K(36) GNN_SMALL_K(40) GNN_SMALL_K(48)
#undef GNN_SMALL_K
    return false;
}

template <int LAYERS>
static bool small_launch_act(int act, int kk0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st)
{
#define GNN_SMALL_CASE(A) case A: return small_launch_k<LAYERS, A>(kk0, a, c, grid, lds_bytes, st);
    switch (act) {
        GNN_SMALL_CASE(GNN_ACT_LINEAR) GNN_SMALL_CASE(GNN_ACT_RELU) GNN_SMALL_CASE(GNN_ACT_SELU) GNN_SMALL_CASE(GNN_ACT_ELU)
        GNN_SMALL_CASE(GNN_ACT_TANH) GNN_SMALL_CASE(GNN_ACT_SIGMOID)
    default: return false;
    }
#undef GNN_SMALL_CASE
}

}   // namespace gnn_fused_dev

bool gnn_small_launch(int layers, int act, int kk0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes,
                      hipStream_t st)
{
    using namespace gnn_fused_dev;
    if (layers == 1) return small_launch_act<1>(act, kk0, a, c, grid, lds_bytes, st);
    if (layers == 2) return small_launch_act<2>(act, kk0, a, c, grid, lds_bytes, st);
    if (layers == 3) return small_launch_act<3>(act, kk0, a, c, grid, lds_bytes, st);
    return false;
}
