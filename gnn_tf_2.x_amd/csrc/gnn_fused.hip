// Fused iteration kernel for gfx950: CSR neighbour gather -> net_state (all Dense layers + BatchNormalization) ->
// convergence test, one launch per iteration of GNN.Loop (reference GNN/GNN.py:223-242 + :202-220).
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
//     from L2 in a pre-packed lane order, 16 B per lane per K step for 4 feature tiles), node activations the B operand.
//     The accumulator of layer l (feature on the register, node on the lane) becomes the B operand of layer l+1 after 8
//     v_permlane32_swap per 32x32 tile, so hidden activations never leave registers.  MFMA f32 evaluates the same
//     k-ordered fmaf chain as the oracle, hence bit-identical results;
//   * epilogue: BatchNormalization, new state to LDS, per-node relative-L2 test in the oracle's summation order,
//     coalesced 256 B row stores, one slotted atomicOr per wave that still moves.
#include <algorithm>
#include <vector>

#include "gnn_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int MAXL = 3;

struct FusedArgs {
    // graph
    int64_t n_rows, row_begin;
    const int32_t *indptr, *adj_src;
    const float *adj_w;
    const float *inv;        // [n_rows, IW] = [nodes | aggregated nodes | aggregated arcs] (label columns of the concat)
    // state
    const float *state_cur;  // [N_pad, Ds] all nodes
    float *state_nxt;        // owned rows
    // shapes
    int Ds, NLc, AL, IW, in_s, KP, lpr, lpr_log2, vec;
    // layers
    const float *Wp[MAXL];   // packed weights: [kk][lane][NT]
    const float *bias[MAXL]; // padded to 32 * NT
    const float *bn_scale, *bn_shift;   // padded to 32 * NT, or nullptr
    int kk[MAXL], nt[MAXL], act[MAXL];
    // control
    float thr;
    const int *gate;
    int *flag_out;
    int world;
};

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

template <int NT>
__device__ __forceinline__ void load_w(const float *p, float (&w)[NT])
{
    if constexpr (NT == 4) {
        const float4 t = *reinterpret_cast<const float4 *>(p);
        w[0] = t.x; w[1] = t.y; w[2] = t.z; w[3] = t.w;
    } else if constexpr (NT == 2) {
        const float2 t = *reinterpret_cast<const float2 *>(p);
        w[0] = t.x; w[1] = t.y;
    } else {
        w[0] = p[0];
    }
}

// bias + activation (+ BatchNormalization on the last layer) on one accumulator tile; feature of register r on this
// lane: 32 jt + (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
__device__ __forceinline__ void tile_epilogue(f32x16 &a, const float *bias, int act, const float *bn_scale,
                                              const float *bn_shift, int jt, int half)
{
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int f0 = 32 * jt + 8 * q + 4 * half;
        const float4 b = *reinterpret_cast<const float4 *>(bias + f0);
        const float bb[4] = {b.x, b.y, b.z, b.w};
        float sc[4] = {1.f, 1.f, 1.f, 1.f}, sh[4] = {0.f, 0.f, 0.f, 0.f};
        if (bn_scale) {
            const float4 s4 = *reinterpret_cast<const float4 *>(bn_scale + f0), h4 = *reinterpret_cast<const float4 *>(bn_shift + f0);
            sc[0] = s4.x; sc[1] = s4.y; sc[2] = s4.z; sc[3] = s4.w;
            sh[0] = h4.x; sh[1] = h4.y; sh[2] = h4.z; sh[3] = h4.w;
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            float v = a[4 * q + t] + bb[t];
            v = gnn_act(v, act);
            if (bn_scale) { const float m = v * sc[t]; v = m + sh[t]; }
            a[4 * q + t] = v;
        }
    }
}

// one Dense layer whose input already sits in registers as B operands (hin, after acc_to_operand)
template <int NT>
__device__ __forceinline__ void layer_from_regs(const f32x16 (&hin)[NT], int nt_in, f32x16 (&acc)[NT], int nt_out,
                                                const float *wp_lane)
{
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
#pragma unroll
    for (int ti = 0; ti < NT; ++ti) {
        if (ti < nt_in) {
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const int reg = 4 * (s >> 2) + ((s & 3) == 1 ? 2 : (s & 3) == 2 ? 1 : (s & 3));
                float w[NT];
                load_w<NT>(wp_lane + (size_t)(16 * ti + s) * 64 * NT, w);
                const float b = hin[ti][reg];
#pragma unroll
                for (int jt = 0; jt < NT; ++jt)
                    if (jt < nt_out) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[jt], b, acc[jt], 0, 0, 0);
            }
        }
    }
}

template <int LAYERS, int NT>
__global__ void __launch_bounds__(256, 2) k_fused(const FusedArgs a)
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
    for (int t = lane; t < 32 * (KP - a.in_s); t += 64) {
        const int i = t / (KP - a.in_s), c = t - i * (KP - a.in_s);
        X[i * KP + a.in_s + c] = 0.0f;
    }
    if (nvalid < 32)
        for (int t = lane; t < (32 - nvalid) * a.in_s; t += 64) {
            const int i = nvalid + t / a.in_s, c = t % a.in_s;
            X[i * KP + c] = 0.0f;
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
        for (int pass = 0; pass * groups < 32; ++pass) {
            const int i = pass * groups + grp;
            const int beg = shfl_i(my_ip, i < nvalid ? i : 0), end = shfl_i(my_ip, i < nvalid ? i + 1 : 0);
            // one column chunk per lane (lpr * vec >= Ds is a precondition of the fused path); lanes past the row width
            // still walk the edges (they feed the broadcasts) on column 0 and store nothing
            const bool colok = gl * a.vec < Ds;
            const int c0 = colok ? gl * a.vec : 0;
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
    f32x16 acc[NT], hid[NT];
#pragma unroll
    for (int jt = 0; jt < NT; ++jt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[jt][r] = 0.0f;
    {
        const float *xb = X + (lane & 31) * KP + half;
        const float *wp = a.Wp[0] + (size_t)lane * NT;
        const int nt0 = a.nt[0];
#pragma unroll 4
        for (int kk = 0; kk < a.kk[0]; ++kk) {
            float w[NT];
            load_w<NT>(wp + (size_t)kk * 64 * NT, w);
            const float b = xb[2 * kk];
#pragma unroll
            for (int jt = 0; jt < NT; ++jt)
                if (jt < nt0) acc[jt] = __builtin_amdgcn_mfma_f32_32x32x2f32(w[jt], b, acc[jt], 0, 0, 0);
        }
    }
#pragma unroll
    for (int l = 1; l < LAYERS; ++l) {
        // hidden epilogue of layer l-1, then turn its accumulators into operands
#pragma unroll
        for (int jt = 0; jt < NT; ++jt) {
            if (jt < a.nt[l - 1]) {
                tile_epilogue(acc[jt], a.bias[l - 1], a.act[l - 1], nullptr, nullptr, jt, half);
                acc_to_operand(acc[jt]);
            }
            hid[jt] = acc[jt];
        }
        layer_from_regs<NT>(hid, a.nt[l - 1], acc, a.nt[l], a.Wp[l] + (size_t)lane * NT);
    }
    // ---- C: last-layer epilogue, new state to LDS (over the aggregated-state columns, no longer needed) ---------------
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
#pragma unroll
    for (int jt = 0; jt < NT; ++jt) {
        if (jt < a.nt[LAYERS - 1]) {
            tile_epilogue(acc[jt], a.bias[LAYERS - 1], a.act[LAYERS - 1], a.bn_scale, a.bn_shift, jt, half);
            float *x = X + (lane & 31) * KP + c_aggs;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int f = 32 * jt + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (f < Ds) x[f] = acc[jt][r];
            }
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

struct FusedPlan {
    int layers = 0, NT = 0, KP = 0, IW = 0;
    int kk[MAXL] = {0, 0, 0}, nt[MAXL] = {0, 0, 0};
    size_t w_off[MAXL] = {0, 0, 0}, b_off[MAXL] = {0, 0, 0}, bn_off = 0, total = 0;
};

bool make_plan(const gnn_mlp *m, FusedPlan &p)
{
    if (m->n_layers < 1 || m->n_layers > MAXL) return false;
    int maxw = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        if (m->acts[l] == GNN_ACT_SOFTMAX) return false;
        maxw = std::max(maxw, m->dims[l + 1]);
    }
    if (maxw > 128) return false;
    p.layers = m->n_layers;
    p.NT = maxw <= 32 ? 1 : (maxw <= 64 ? 2 : 4);
    const int in_even = (m->dims[0] + 1) & ~1;
    p.KP = in_even + 1;
    size_t off = 0;
    for (int l = 0; l < m->n_layers; ++l) {
        p.nt[l] = (m->dims[l + 1] + 31) / 32;
        p.kk[l] = l == 0 ? in_even / 2 : 16 * p.nt[l - 1];
        p.w_off[l] = off;
        off += (size_t)p.kk[l] * 64 * p.NT;
        off = (off + 3) & ~(size_t)3;
    }
    for (int l = 0; l < m->n_layers; ++l) { p.b_off[l] = off; off += 32 * p.NT; }
    p.bn_off = off;
    off += 2 * 32 * p.NT;
    p.total = off;
    return true;
}

}   // namespace

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
int gnn_fused_pack(gnn_mlp *m)
{
    FusedPlan p;
    if (!make_plan(m, p)) { gnn_fused_release(m); return GNN_OK; }
    std::vector<float> img(p.total, 0.0f);
    std::vector<float> W, b;
    for (int l = 0; l < m->n_layers; ++l) {
        const int n_in = m->dims[l], n_out = m->dims[l + 1];
        W.resize((size_t)n_in * n_out);
        b.resize(n_out);
        HIPCHK(hipMemcpy(W.data(), m->W[l], W.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(b.data(), m->b[l], b.size() * sizeof(float), hipMemcpyDeviceToHost));
        float *wp = img.data() + p.w_off[l];
        for (int kk = 0; kk < p.kk[l]; ++kk)
            for (int lane = 0; lane < 64; ++lane)
                for (int jt = 0; jt < p.NT; ++jt) {
                    const int k = 2 * kk + (lane >> 5), j = 32 * jt + (lane & 31);
                    wp[((size_t)kk * 64 + lane) * p.NT + jt] = (k < n_in && j < n_out) ? W[(size_t)k * n_out + j] : 0.0f;
                }
        for (int j = 0; j < n_out; ++j) img[p.b_off[l] + j] = b[j];
    }
    if (m->has_bn) {
        const int f = m->dims.back();
        HIPCHK(hipMemcpy(img.data() + p.bn_off, m->bn_scale, f * sizeof(float), hipMemcpyDeviceToHost));
        HIPCHK(hipMemcpy(img.data() + p.bn_off + 32 * p.NT, m->bn_shift, f * sizeof(float), hipMemcpyDeviceToHost));
    }
    if (m->packed_floats != p.total) {
        gnn_fused_release(m);
        HIPCHK(hipMalloc((void **)&m->packed, p.total * sizeof(float)));
        m->packed_floats = p.total;
    }
    HIPCHK(hipMemcpy(m->packed, img.data(), p.total * sizeof(float), hipMemcpyHostToDevice));
    return GNN_OK;
}

void gnn_fused_release(gnn_mlp *m)
{
    if (m->packed) (void)hipFree(m->packed);
    m->packed = nullptr;
    m->packed_floats = 0;
}

bool gnn_fused_supported(const gnn_loop *l)
{
    FusedPlan p;
    if (!l->st->packed || !make_plan(l->st, p)) return false;
    if ((size_t)4 * 32 * p.KP * sizeof(float) > 80 * 1024) return false;   // two workgroups per CU
    const int Ds = l->Ds;
    if (!((Ds % 4 == 0 && Ds <= 256) || Ds <= 64)) return false;           // one column chunk per lane in the gather
    return l->g->n_rows > 0;
}

int gnn_fused_prepare(gnn_loop *l)
{
    const gnn_graph *g = l->g;
    const int IW = 2 * l->NLc + g->AL;
    if (!l->inv) {
        HIPCHK(hipMalloc((void **)&l->inv, std::max<size_t>(1, (size_t)g->n_rows * IW) * sizeof(float)));
    }
    if (IW == 0) return GNN_OK;
    // [nodes | Adjacency^T . nodes | ArcNode^T . arc labels]  (GNN.py:263, :259); recomputed per run: labels may have
    // been rewritten by gnn_graph_update_labels
    int rc = gnn_launch_spmm(l->stream, g->n_rows, g->sh->indptr, nullptr, g->sh->arc_w, g->sh->arc_labels, g->AL, g->AL,
                             l->inv + 2 * l->NLc, IW, nullptr, 1);
    if (rc) return rc;
    if (l->NLc) {
        rc = gnn_launch_spmm(l->stream, g->n_rows, g->sh->indptr, g->sh->adj_src, g->sh->adj_w, g->nodes, g->NL, g->NL,
                             l->inv + l->NLc, IW, nullptr, 1);
        if (rc) return rc;
        HIPCHK(hipMemcpy2DAsync(l->inv, sizeof(float) * IW, g->nodes + (size_t)g->row_begin * g->NL, sizeof(float) * g->NL,
                                sizeof(float) * g->NL, (size_t)g->n_rows, hipMemcpyDeviceToDevice, l->stream));
    }
    return GNN_OK;
}

template <int LAYERS, int NT>
static void launch_fused(const FusedArgs &a, unsigned grid, size_t lds, hipStream_t st)
{
    static bool raised = false;   // dynamic LDS above 64 KiB has to be requested once per kernel
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_fused<LAYERS, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    hipLaunchKernelGGL((k_fused<LAYERS, NT>), grid, 256, lds, st, a);
}

int gnn_fused_iteration(gnn_loop *l, int k)
{
    const gnn_graph *g = l->g;
    const gnn_mlp *m = l->st;
    FusedPlan p;
    if (!make_plan(m, p)) return gnn_fail(GNN_ERR_UNSUPPORTED, "fused path does not cover this net_state");
    const int cur = k & 1, nxt = cur ^ 1, P = l->world;
    FusedArgs a{};
    a.n_rows = g->n_rows; a.row_begin = g->row_begin;
    a.indptr = g->sh->indptr; a.adj_src = g->sh->adj_src; a.adj_w = g->sh->adj_w;
    a.inv = l->inv;
    a.state_cur = l->state[cur];
    a.state_nxt = l->state[nxt] + (size_t)g->row_begin * l->Ds;
    a.Ds = l->Ds; a.NLc = l->NLc; a.AL = g->AL; a.IW = 2 * l->NLc + g->AL; a.in_s = l->in_s; a.KP = p.KP;
    a.vec = (l->Ds % 4 == 0) ? 4 : 1;
    int lpr = 1, lg = 0;
    while (lpr * a.vec < l->Ds && lpr < 64) { lpr <<= 1; ++lg; }
    a.lpr = lpr; a.lpr_log2 = lg;
    for (int i = 0; i < p.layers; ++i) {
        a.Wp[i] = m->packed + p.w_off[i];
        a.bias[i] = m->packed + p.b_off[i];
        a.kk[i] = p.kk[i]; a.nt[i] = p.nt[i]; a.act[i] = m->acts[i];
    }
    a.bn_scale = m->has_bn ? m->packed + p.bn_off : nullptr;
    a.bn_shift = m->has_bn ? m->packed + p.bn_off + 32 * p.NT : nullptr;
    a.thr = l->thr;
    a.gate = l->flags + (size_t)k * P * GNN_FLAG_WORDS;
    a.flag_out = l->flags + ((size_t)(k + 1) * P + l->rank) * GNN_FLAG_WORDS;
    a.world = P;
    const unsigned grid = (unsigned)((g->n_rows + 127) / 128);
    const size_t lds = (size_t)4 * 32 * p.KP * sizeof(float);
    switch (p.layers * 10 + p.NT) {
    case 11: launch_fused<1, 1>(a, grid, lds, l->stream); break;
    case 12: launch_fused<1, 2>(a, grid, lds, l->stream); break;
    case 14: launch_fused<1, 4>(a, grid, lds, l->stream); break;
    case 21: launch_fused<2, 1>(a, grid, lds, l->stream); break;
    case 22: launch_fused<2, 2>(a, grid, lds, l->stream); break;
    case 24: launch_fused<2, 4>(a, grid, lds, l->stream); break;
    case 31: launch_fused<3, 1>(a, grid, lds, l->stream); break;
    case 32: launch_fused<3, 2>(a, grid, lds, l->stream); break;
    case 34: launch_fused<3, 4>(a, grid, lds, l->stream); break;
    default: return gnn_fail(GNN_ERR_UNSUPPORTED, "no fused instantiation for %d layers / %d tiles", p.layers, p.NT);
    }
    HIPCHK(hipGetLastError());
    return GNN_OK;
}
