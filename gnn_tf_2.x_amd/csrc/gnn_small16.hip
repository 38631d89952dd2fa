// Persistent small-graph loop on 16-node tiles (see gnn_small.hip for the 32-node form and the hand-off protocol).
//
// A body of the persistent loop is a chain of latencies, not of throughput: one wave per tile walks gather -> three dependent dense
// layers -> condition -> barrier, and with 32 nodes per wave the dense layers alone are 48 dependent v_mfma_f32_32x32x2_f32 (64 cycles
// each) plus 16 activations per lane and layer.  Batches of up to 4,096 nodes therefore run on 16-node tiles: twice the workgroups,
// v_mfma_f32_16x16x4_f32 (32 cycles, four k per instruction, accumulated in k order like the 32x32x2 form: the parity tests compare
// every bit with the C oracle's k-ordered fmaf chain), 8 activations per lane and layer, one 16-byte load per lane and arc.
//   * layer inputs are the B operand (k = 4 s + lane / 16, node = lane % 16): layer 0 reads it from the LDS tile, deeper layers from
//     a small LDS copy of the previous activations (the accumulator holds features 16 j + 4 (lane / 16) + r, so the hop through LDS
//     is the transpose);
//   * weights are the A operand, read once per launch from the Keras-layout kernels (W[k][f], k = 4 s + lane / 16,
//     f = 16 j + lane % 16) and kept in registers.
// Everything else - padded exchange rows, gate words, give-up status, output stage and graph readout inside the launch - is as in
// k_small_loop (reference GNN/GNN.py:202-242, :262-279, :331-332).
#include "gnn_fused_kernel.h"

namespace gnn_fused_dev {

typedef unsigned v4u16 __attribute__((ext_vector_type(4)));
constexpr int GNN_SMALL16_ECACHE = 1024;
constexpr int GNN_SMALL16_HP = 36;                 // row stride of the hidden-activation copy (floats): 16-byte rows, (4 n + g) banks

// aggregated neighbour states of the tile's 16 rows from the padded exchange rows: 4 lanes per node, HW floats per lane
template <int HW, int PR>
__device__ __forceinline__ void small16_gather(__amdgpu_buffer_rsrc_t rs, float *X, const int *ipt, int lane, int nvalid, int KP, int c_aggs, int Ds,
                                               const int *adj_src, const float *adj_w, const int *ec_src, const float *ec_w, int ec_base)
{
    const int node = lane & 15, part = lane >> 4;
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
            off[u] = (src * (4 * HW) + part * HW) * 4;
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
        float *x = X + node * KP + c_aggs + part * HW;
#pragma unroll
        for (int c = 0; c < HW; ++c)
            if (part * HW + c < Ds) x[c] = acc[c];
    }
}

// the tile's 16 rows -> its block of the padded buffer (rows >= nrows, columns >= Ds: zeros)
template <int DP>
__device__ __forceinline__ void small16_store(__amdgpu_buffer_rsrc_t rs, int64_t i0, const float *src, int rs_, int nrows, int Ds, int lane)
{
    constexpr int QR = DP / 4;
#pragma unroll
    for (int u = 0; u < (16 * QR) / 64; ++u) {
        const int q = lane + 64 * u, row = q / QR, c4 = (q % QR) * 4;
        const float *x = src + row * rs_ + c4;
        const bool rok = row < nrows;
        v4f v;
        v.x = (rok && c4 < Ds) ? x[0] : 0.0f;
        v.y = (rok && c4 + 1 < Ds) ? x[1] : 0.0f;
        v.z = (rok && c4 + 2 < Ds) ? x[2] : 0.0f;
        v.w = (rok && c4 + 3 < Ds) ? x[3] : 0.0f;
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u16, v), rs, (int)(i0 * DP + 4 * q) * 4, 0, 16);       // aux 16 = sc1: write-through
    }
}

// A operand of (K-step s, feature tile j) from the Keras-layout kernel W[din][dout]
__device__ __forceinline__ float small16_w(const float *W, int din, int dout, int s, int j, int lane)
{
    const int k = 4 * s + (lane >> 4), f = 16 * j + (lane & 15);
    return (k < din && f < dout) ? gload1(W + (size_t)k * dout + f) : 0.0f;
}

// bias + activation of a hidden layer's accumulators, written to the LDS copy H[node][feature] the next layer reads its B operand from
template <int ACT>
__device__ __forceinline__ void small16_hidden(const v4f (&acc)[2], const float *bias, float *H, int lane)
{
    const int n = lane & 15, g = lane >> 4;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const v4f bb = *reinterpret_cast<const v4f *>(bias + 16 * j + 4 * g);
        const v2f p0 = act_t2<ACT>(v2f{acc[j].x, acc[j].y} + v2f{bb.x, bb.y});
        const v2f p1 = act_t2<ACT>(v2f{acc[j].z, acc[j].w} + v2f{bb.z, bb.w});
        *reinterpret_cast<v4f *>(H + n * GNN_SMALL16_HP + 16 * j + 4 * g) = v4f{p0.x, p0.y, p1.x, p1.y};
    }
}

template <int STEPS>
__device__ __forceinline__ void small16_layer(const float *b_base, const float (&w)[STEPS][2], v4f (&acc)[2], bool second)
{
    float b[STEPS];
#pragma unroll
    for (int s = 0; s < STEPS; ++s) b[s] = b_base[4 * s];
    acc[0] = v4f{0.f, 0.f, 0.f, 0.f};
    acc[1] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int s = 0; s < STEPS; ++s) acc[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s][0], b[s], acc[0], 0, 0, 0);
    if (second) {
#pragma unroll
        for (int s = 0; s < STEPS; ++s) acc[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[s][1], b[s], acc[1], 0, 0, 0);
    } else {
        // The branch around the second chain puts the first chain's last MFMA directly in front of whatever reads acc[0] next, in
        // another basic block - and hipcc (ROCm 7.2) then inserts no wait states between an 8-pass MFMA and the v_accvgpr_read of its
        // last destination register: with the linear activation (the read follows at once) feature 3 of every node came out stale
        // (found by test_persistent_small_graph_loop_random_shapes).  Sixteen wait states here cover the 8-pass result.
        // Guard: every object's rule in the Makefile scans its ISA listing (tools/scan_mfma_hazard.py, -save-temps=obj) and fails the build
        // if such an edge reappears.  Tying the wait to the data instead - acc[0] as an in / out operand of this statement,
        // "+v" or "+a" - was tried in round 4 and RE-CREATES the hazard: the compiler then copies the accumulator for the operand
        // (v_accvgpr_read / v_accvgpr_mov) as the first instruction of this block, in front of the s_nop (108 hits in the scan).
        asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");
    }
}

// S0: K-steps (of 4) of layer 0 kept in registers
template <int LAYERS, int ACT, int S0>
__global__ void __launch_bounds__(64) k_small16(const GnnFusedArgs a0, const GnnSmallCtl c)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x, n = lane & 15, g = lane >> 4;
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
    const int KP = c.KP16, Ds = a0.Ds, c_aggs = a0.c_aggs;
    float *X = lds;                                               // the tile [16][KP]: own state | labels | aggregated state | aggregated labels | zeros
    float *H = X + 16 * KP;                                       // hidden activations [16][GNN_SMALL16_HP]
    int *ipt = reinterpret_cast<int *>(H + 16 * GNN_SMALL16_HP);  // row pointers [17] (+3)
    float *ep = reinterpret_cast<float *>(ipt + 20);              // last-layer bias, BatchNormalization scale / shift [3][32]
    float *hb = ep + 96;                                          // biases of the hidden layers [2][32]
    float *hw = hb + 64;                                          // net_output head: W [wf * T <= 512], then b | BN scale | BN shift [3][8]
    float *scr = hw + 544;                                        // [0, 512): rows of the tile in [row][Ds] order (initial / final state); [512, 1024): its label rows
    int *ec_src = reinterpret_cast<int *>(scr + 1024);            // the tile's arc ids / weights, kept for every body
    float *ec_w = scr + 1024 + GNN_SMALL16_ECACHE;
    for (int t = lane; t < 3 * 32; t += 64) {
        const int which = t >> 5, f = t & 31;
        ep[t] = which == 0 ? a0.bias[LAYERS - 1][f] : (a0.bn_scale ? (which == 1 ? a0.bn_scale[f] : a0.bn_shift[f]) : 0.0f);
    }
    if constexpr (LAYERS >= 2) {
        if (lane < 32 * (LAYERS - 1)) hb[lane] = a0.bias[lane >> 5][lane & 31];
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
    const int64_t i0 = (int64_t)blockIdx.x * 16;
    const int nvalid = (int)((a0.n_rows - i0) < 16 ? (a0.n_rows - i0) : 16);
    // everything the launch reads from read-only memory is requested here, at once: row pointers, the tile's initial rows, label
    // columns, and what the output stage needs at the very end (label rows, mask, output position)
    const int my_ip = (lane <= nvalid) ? gload1(a0.indptr + i0 + lane) : 0;
    float v_init[8];
    {
        const float *init = c.init + i0 * Ds;
#pragma unroll
        for (int u = 0; u < 8; ++u) v_init[u] = (lane + 64 * u < nvalid * Ds) ? gload1(init + lane + 64 * u) : 0.0f;
    }
    bool out_on = false;
    int out_pos = 0;
    if (c.out) {
        const int nl = c.NLc ? nvalid * c.NL : 0;                   // <= 512 (NL <= 32)
        const float *nod = c.nodes_own + i0 * c.NL;
        float lv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) lv[u] = (lane + 64 * u < nl) ? gload1(nod + lane + 64 * u) : 0.0f;
        out_on = lane < nvalid && c.mask[i0 + (lane < nvalid ? lane : 0)];
        out_pos = out_on ? c.mask_pos[i0 + lane] : 0;
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (lane + 64 * u < nl) scr[512 + lane + 64 * u] = lv[u];
    }
    // weights: once, into registers (A operands)
    float w0[S0][2], w1[8][2], w2[8][2];
#pragma unroll
    for (int s = 0; s < S0; ++s)
#pragma unroll
        for (int j = 0; j < 2; ++j) w0[s][j] = small16_w(c.Wraw[0], c.din[0], c.dout[0], s, j, lane);
    if constexpr (LAYERS >= 2) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) w1[s][j] = small16_w(c.Wraw[1], c.din[1], c.dout[1], s, j, lane);
    }
    if constexpr (LAYERS >= 3) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int j = 0; j < 2; ++j) w2[s][j] = small16_w(c.Wraw[2], c.din[2], c.dout[2], s, j, lane);
    }
    // the tile skeleton: zeros everywhere, then the label columns (they never change)
    for (int t = lane; t < 16 * KP; t += 64) X[t] = 0.0f;
    {
        const int last_ip = shfl_i(my_ip, nvalid);
        if (lane <= 16) ipt[lane] = lane <= nvalid ? my_ip : last_ip;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    if (a0.IW > 0) {
        const float *src = a0.inv + i0 * a0.IW;
        const int total = nvalid * a0.IW;
        RowCol rc(lane, a0.IW);
        for (int t = lane; t < total; t += 64, rc.next()) X[rc.i * KP + label_col(rc.c, Ds, a0.NLc, c_aggs)] = gload1(src + t);
    }
    const int e_base = ipt[0], e_cnt = ipt[16] - e_base;
    const bool ecached = e_cnt <= c.ecache;
    if (ecached)
        for (int t = lane; t < e_cnt; t += 64) { ec_src[t] = gload1(a0.adj_src + e_base + t); ec_w[t] = gload1(a0.adj_w + e_base + t); }
    // the gate words of the NEXT run (the other half of the double buffer): nobody reads them during this launch
    if (blockIdx.x == 0)
        for (int t = lane; t < c.n_words; t += 64) c.zero_words[t] = 0;
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    SMALL_STAMP();                                                   // 1: set-up
    const unsigned n_wg = gridDim.x;
    const int xs_bytes = (int)gridDim.x * 16 * c.DP * 4;
    const __amdgpu_buffer_rsrc_t xs_rs[2] = {__builtin_amdgcn_make_buffer_rsrc(c.xs, 0, xs_bytes, 0x00020000),
                                             __builtin_amdgcn_make_buffer_rsrc(c.xs + (size_t)gridDim.x * 16 * c.DP, 0, xs_bytes, 0x00020000)};
    // grid barrier + gate in one word per body (see k_small_loop): 1 = run body b, 0 = converged, -1 = gave up
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
            if (lane == 0) c.host_result[1] = 1;                             // sticky: only ever set here, cleared by the host before the launch
            return -1;
        }
        return (seen >> 16) ? 1 : 0;
    };
    // ---- state <- initial state (GNN.py:262 / :265), first condition against ones (GNN.py:266, :271; k_check's order) --------------
    int go;
    {
        float *own0 = c.state0 + (a0.row_begin + i0) * Ds;
        const int total = nvalid * Ds;
        RowCol rc(lane, Ds);
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (lane + 64 * u < total) {
                *gptr_w(own0 + lane + 64 * u) = v_init[u];              // replica 0: read by nobody in this launch (k == 0: the final state)
                scr[lane + 64 * u] = v_init[u];
                X[rc.i * KP + rc.c] = v_init[u];                        // the tile's own-state columns
            }
            rc.next();
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (c.DP == 16) small16_store<16>(xs_rs[0], i0, scr, Ds, nvalid, Ds, lane);      // what body 0 gathers from, behind gate 0
        else small16_store<32>(xs_rs[0], i0, scr, Ds, nvalid, Ds, lane);
        int moved = 0;
        if (lane < nvalid) {
            float dist = 0.0f, nrm = 0.0f;
            for (int f = 0; f < Ds; ++f) {
                const float df = scr[lane * Ds + f] - 1.0f;
                const float dd = df * df;
                dist = dist + dd;
                nrm = nrm + 1.0f;
            }
            moved = sqrtf(dist) > a0.thr * sqrtf(nrm);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // 2: initial state, first condition
        go = arrive_and_gate(0, __any(moved));
        SMALL_STAMP();                                               // 3: gate 0
    }
    const bool last_two = Ds > 16;                                   // the last layer needs its second feature tile
    int k = 0;
    for (; k < c.max_iter && go == 1; ++k) {
        if (k > 0) {                                                 // the new state of the last body becomes the own state
            const int total = nvalid * Ds;
            RowCol rc(lane, Ds);
            for (int t = lane; t < total; t += 64, rc.next()) X[rc.i * KP + rc.c] = X[rc.i * KP + c_aggs + rc.c];
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // before the aggregated state overwrites those columns
        }
        {
            const __amdgpu_buffer_rsrc_t rs = xs_rs[k & 1];
            const int *es = ecached ? ec_src : nullptr;
            const float *ew = ecached ? ec_w : nullptr;
            if (c.DP == 32) {
                if (c.rnd == 8) small16_gather<8, 8>(rs, X, ipt, lane, nvalid, KP, c_aggs, Ds, a0.adj_src, a0.adj_w, es, ew, e_base);
                else small16_gather<8, 4>(rs, X, ipt, lane, nvalid, KP, c_aggs, Ds, a0.adj_src, a0.adj_w, es, ew, e_base);
            } else {
                if (c.rnd == 8) small16_gather<4, 8>(rs, X, ipt, lane, nvalid, KP, c_aggs, Ds, a0.adj_src, a0.adj_w, es, ew, e_base);
                else small16_gather<4, 4>(rs, X, ipt, lane, nvalid, KP, c_aggs, Ds, a0.adj_src, a0.adj_w, es, ew, e_base);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // body + 0: gather
        v4f out[2];
        const float *xb = X + n * KP + g, *hbp = H + n * GNN_SMALL16_HP + g;
        if constexpr (LAYERS == 1) {
            small16_layer<S0>(xb, w0, out, last_two);
        } else {
            v4f h[2];
            small16_layer<S0>(xb, w0, h, true);
            small16_hidden<ACT>(h, hb, H, lane);
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if constexpr (LAYERS == 2) {
                small16_layer<8>(hbp, w1, out, last_two);
            } else {
                small16_layer<8>(hbp, w1, h, true);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // (the B operands were read before H is rewritten)
                small16_hidden<ACT>(h, hb + 32, H, lane);
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                small16_layer<8>(hbp, w2, out, last_two);
            }
        }
        // last layer: bias, activation, BatchNormalization; the new state into the aggregated-state columns (no longer needed)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !last_two) break;
            const int f0 = 16 * j + 4 * g;
            const v4f bb = *reinterpret_cast<const v4f *>(ep + f0);
            v2f p0 = act_t2<ACT>(v2f{out[j].x, out[j].y} + v2f{bb.x, bb.y});
            v2f p1 = act_t2<ACT>(v2f{out[j].z, out[j].w} + v2f{bb.z, bb.w});
            if (a0.bn_scale) {
                const v4f sc = *reinterpret_cast<const v4f *>(ep + 32 + f0), sh = *reinterpret_cast<const v4f *>(ep + 64 + f0);
                const v2f m0 = p0 * v2f{sc.x, sc.y}, m1 = p1 * v2f{sc.z, sc.w};
                p0 = m0 + v2f{sh.x, sh.y};
                p1 = m1 + v2f{sh.z, sh.w};
            }
            float *x = X + n * KP + c_aggs + f0;
            if (f0 < Ds) x[0] = p0.x;
            if (f0 + 1 < Ds) x[1] = p0.y;
            if (f0 + 2 < Ds) x[2] = p1.x;
            if (f0 + 3 < Ds) x[3] = p1.y;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // body + 1: dense layers, new state in LDS
        // the new rows first (they drain while the condition is evaluated), then the condition of GNN.py:202-220 in k_check's order:
        // lanes 0-15 sum (new - old)^2, lanes 16-31 sum old^2, ascending feature, unfused
        if (c.DP == 16) small16_store<16>(xs_rs[(k & 1) ^ 1], i0, X + c_aggs, KP, 16, Ds, lane);
        else small16_store<32>(xs_rs[(k & 1) ^ 1], i0, X + c_aggs, KP, 16, Ds, lane);
        int moved;
        {
            const float *xo = X + n * KP, *xn = xo + c_aggs;
            float s_ = 0.0f;
            if (g < 2)
                for (int f = 0; f < Ds; ++f) {
                    const float o = xo[f];
                    const float d = g ? o : (xn[f] - o);
                    const float dd = d * d;
                    s_ = s_ + dd;
                }
            const float root = sqrtf(s_);
            const float nrm = shfl_f(root, n + 16);
            const float rhs = a0.thr * nrm;
            moved = __any((g == 0) && (n < nvalid) && (root > rhs)) ? 1 : 0;
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        SMALL_STAMP();                                               // body + 2: row stores issued, condition
        go = arrive_and_gate(k + 1, moved);
        SMALL_STAMP();                                               // body + 3: barrier + gate
    }
    if (go < 0) return;                      // (status word set; the host repeats the Loop with one launch per body)
    if (k > 0) {                             // the final state of the tile's rows into the [N, Ds] replica the host expects it in (k & 1)
        float *dst = ((k & 1) ? c.state1 : c.state0) + (a0.row_begin + i0) * Ds;
        const int total = nvalid * Ds;
        RowCol rc(lane, Ds);
        for (int t = lane; t < total; t += 64, rc.next()) {
            const float v = X[rc.i * KP + c_aggs + rc.c];
            *gptr_w(dst + t) = v;
            scr[t] = v;                      // [row][Ds] order for the output stage (k == 0: the initial rows are there already)
        }
    }
    if (blockIdx.x == 0 && lane == 0) {      // executed bodies (GNN.py:267; every workgroup agrees)
        c.kfinal[0] = k;
        c.host_result[0] = k;
    }
    // ---- apply_filters + one-layer net_output on the tile's masked rows (GNN.py:275-279), arithmetic as k_out1 ------------------------
    if (c.out) {
        const int wf = Ds + c.NLc, T = c.T, NL = c.NL;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (out_on) {
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = 0.0f;
            for (int kk = 0; kk < wf; ++kk) {                            // k-ordered fmaf chain per output
                const float x = kk < Ds ? scr[lane * Ds + kk] : scr[512 + lane * NL + (kk - Ds)];
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
    // ---- graph readout (GNN.py:331-332) by workgroup 0 after one more grid barrier; the result goes straight to pinned host memory -----
    if (c.ng_ip) {
        if (arrive_and_gate(c.ro_word, 0) < 0) return;
        if (blockIdx.x == 0) small_graph_readout(c, lane);
    }
}

template <int LAYERS, int ACT>
static bool small16_launch_s(int s0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st)
{
#define GNN_SMALL16_S(S)                                                                                                    \
    if (s0 == S) {                                                                                                          \
        hipLaunchKernelGGL((k_small16<LAYERS, ACT, S>), grid, 64, lds_bytes, st, a, c);                                       \
        return true;                                                                                                        \
    }
    GNN_SMALL16_S(4) GNN_SMALL16_S(8) GNN_SMALL16_S(12) GNN_SMALL16_S(16) GNN_SMALL16_S(20) GNN_SMALL16_S(24)
#undef GNN_SMALL16_S
    return false;
}

template <int LAYERS>
static bool small16_launch_act(int act, int s0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st)
{
#define GNN_SMALL16_CASE(A) case A: return small16_launch_s<LAYERS, A>(s0, a, c, grid, lds_bytes, st);
    switch (act) {
        GNN_SMALL16_CASE(GNN_ACT_LINEAR) GNN_SMALL16_CASE(GNN_ACT_RELU) GNN_SMALL16_CASE(GNN_ACT_SELU) GNN_SMALL16_CASE(GNN_ACT_ELU)
        GNN_SMALL16_CASE(GNN_ACT_TANH) GNN_SMALL16_CASE(GNN_ACT_SIGMOID)
    default: return false;
    }
#undef GNN_SMALL16_CASE
}

}   // namespace gnn_fused_dev

size_t gnn_small16_lds_bytes(int kp16)
{
    using namespace gnn_fused_dev;
    return sizeof(float) * ((size_t)16 * kp16 + 16 * GNN_SMALL16_HP + 20 + 96 + 64 + 544 + 1024 + 2 * GNN_SMALL16_ECACHE + 4);
}

bool gnn_small16_launch(int layers, int act, int s0, const GnnFusedArgs &a, const GnnSmallCtl &c, unsigned grid, size_t lds_bytes, hipStream_t st)
{
    using namespace gnn_fused_dev;
    if (layers == 1) return small16_launch_act<1>(act, s0, a, c, grid, lds_bytes, st);
    if (layers == 2) return small16_launch_act<2>(act, s0, a, c, grid, lds_bytes, st);
    if (layers == 3) return small16_launch_act<3>(act, s0, a, c, grid, lds_bytes, st);
    return false;
}
