/*
 * CPU oracle (plain C) for the GNN.Loop state-propagation hot path.
 *
 * TEST INFRASTRUCTURE ONLY: the product (gnn_tf_2.x_amd/) never links, loads or calls this file.  It is used by
 * tests/, by __graft_entry__.smoke() as the checker, and by the cpu_baseline leg of bench.py.
 *
 * It restates, op for op, the TF2 sequence of the reference (paths relative to the reference root):
 *   GNN/GNN.py:259,263   loop-invariant aggregates   ArcNode^T . arc_labels, Adjacency^T . nodes      -> orc_spmm
 *   GNN/GNN.py:202-220   condition                   any_i( ||s-s_old|| > thr ||s_old|| ) and k < max  -> orc_not_converged
 *   GNN/GNN.py:223-242   convergence                 SpMM -> concat (materialised, like TF) -> net_state -> orc_loop body
 *   GNN/GNN.py:245-248   apply_filters               boolean_mask of [state | nodes?]                  -> orc_loop tail
 *   GNN/GNN.py:279       net_output                                                                     -> orc_mlp
 *   GNN/MLP.py:11-64     Sequential = Dense(act)... [+ BatchNormalization], Dropout = identity at inference
 *
 * TensorFlow itself is not available: the floating-point evaluation ORDER below is this project's
 * definition, chosen so that a GPU can reproduce it bit for bit:
 *   - SpMM:   acc = fmaf(w_e, x[src_e], acc), entries of a row in stored (row-major reordered) order;
 *   - Dense:  acc = 0; for k ascending: acc = fmaf(x[k], W[k][j], acc); y = act(acc + b[j])
 *             (this is exactly the chain v_mfma_f32_32x32x2_f32 evaluates);
 *   - BatchNormalization (inference): inv = (1/sqrt(var+eps))*gamma; y = x*inv + (beta - mean*inv), unfused;
 *   - norms:  acc = acc + d*d for d ascending, unfused; sqrtf correctly rounded; strict '>';
 *   - expf:   orc_expf below (2^(x log2 e): rint, exact fraction, degree-5 polynomial, ldexp; every step an explicit IEEE op).
 * Compile with -ffp-contract=off so that nothing else is fused.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ACT_LINEAR = 0, ACT_RELU = 1, ACT_SELU = 2, ACT_ELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5, ACT_SOFTMAX = 6 };

/* exp(x) in float, every operation spelled out (the GPU kernels carry an identical sequence):
 * 2^u with u = x * log2(e) rounded once, n = rint(u), f = u - n (exact), degree-5 polynomial for 2^f on [-1/2, 1/2]
 * (least-squares fit on Chebyshev nodes, max relative error 1.6e-7, exp(0) == 1), scaled by 2^n with ldexpf (exact, one rounding on
 * underflow only).  Relative error <= 1.6e-7 + 6e-8 |x|. */
float orc_expf(float x)
{
    if (x != x) return x;
    if (x > 88.72283935546875f) return INFINITY;
    if (x < -87.33654022216797f) return 0.0f;
    float u = x * 1.44269504088896341f;
    float n = rintf(u);
    float f = u - n;
    float p = 0.0013218672247603536f;
    p = fmaf(p, f, 0.009671698324382305f);
    p = fmaf(p, f, 0.05550893023610115f);
    p = fmaf(p, f, 0.24022237956523895f);
    p = fmaf(p, f, 0.6931468844413757f);
    p = fmaf(p, f, 1.0f);
    return ldexpf(p, (int)n);
}

static inline float act_scalar(float v, int act)
{
    switch (act) {
    case ACT_RELU: return v > 0.0f ? v : 0.0f;
    case ACT_SELU: {
        float neg = 1.6732632423543772f * (orc_expf(v) - 1.0f);
        return 1.0507009873554805f * (v > 0.0f ? v : neg);
    }
    case ACT_ELU: return v > 0.0f ? v : (orc_expf(v) - 1.0f);
    case ACT_TANH: {
        float a = fabsf(v);
        float t = orc_expf(-2.0f * a);
        float q = (1.0f - t) / (1.0f + t);
        return v < 0.0f ? -q : q;
    }
    case ACT_SIGMOID: return 1.0f / (1.0f + orc_expf(-v));
    default: return v;
    }
}

/* Timing knob for the CPU baseline of bench.py: TensorFlow's CPU SparseTensorDenseMatMul is believed to be
 * single-threaded, so the baseline is reported a second time with the sparse products on ONE thread (dense layers on all). */
static int g_spmm_single_thread = 0;
void orc_set_spmm_single_thread(int on) { g_spmm_single_thread = on != 0; }

/* out[r] = sum_e val[e] * dense[inner[e]] over the stored entries of row r, in order */
void orc_spmm(int64_t n_rows, const int32_t *indptr, const int32_t *inner, const float *val, const float *dense,
              int width, float *out, int64_t ld_out)
{
#pragma omp parallel for schedule(dynamic, 256) if (!g_spmm_single_thread)
    for (int64_t r = 0; r < n_rows; ++r) {
        float *o = out + r * ld_out;
        for (int c = 0; c < width; ++c) o[c] = 0.0f;
        for (int32_t e = indptr[r]; e < indptr[r + 1]; ++e) {
            const float w = val[e];
            const float *x = dense + (int64_t)inner[e] * width;
            for (int c = 0; c < width; ++c) o[c] = fmaf(w, x[c], o[c]);
        }
    }
}

/* Y[n, n_out] = act(X[n, n_in] . W[n_in, n_out] + b), k-ordered fmaf chain per output element */
void orc_dense(int64_t n, int n_in, int n_out, const float *X, int64_t ldx, const float *W, const float *b, int act,
               float *Y, int64_t ldy)
{
    enum { RB = 4 };
#pragma omp parallel
    {
        float *acc = (float *)malloc(sizeof(float) * RB * (size_t)n_out);
#pragma omp for schedule(static)
        for (int64_t i0 = 0; i0 < n; i0 += RB) {
            const int rb = (int)((n - i0) < RB ? (n - i0) : RB);
            for (int t = 0; t < rb * n_out; ++t) acc[t] = 0.0f;
            for (int k = 0; k < n_in; ++k) {
                const float *w = W + (size_t)k * n_out;
                for (int r = 0; r < rb; ++r) {
                    const float a = X[(i0 + r) * ldx + k];
                    float *ac = acc + r * n_out;
                    for (int j = 0; j < n_out; ++j) ac[j] = fmaf(a, w[j], ac[j]);
                }
            }
            for (int r = 0; r < rb; ++r) {
                float *y = Y + (i0 + r) * ldy;
                float *ac = acc + r * n_out;
                for (int j = 0; j < n_out; ++j) ac[j] = ac[j] + b[j];
                if (act == ACT_SOFTMAX) {
                    float m = ac[0];
                    for (int j = 1; j < n_out; ++j) m = ac[j] > m ? ac[j] : m;
                    float s = 0.0f;
                    for (int j = 0; j < n_out; ++j) { ac[j] = orc_expf(ac[j] - m); s = s + ac[j]; }
                    for (int j = 0; j < n_out; ++j) y[j] = ac[j] / s;
                } else {
                    for (int j = 0; j < n_out; ++j) y[j] = act_scalar(ac[j], act);
                }
            }
        }
        free(acc);
    }
}

/* Keras BatchNormalization, inference; bn = [gamma | beta | mean | var], each n_feat long */
void orc_batchnorm(int64_t n, int n_feat, float *Y, int64_t ldy, const float *bn, float eps)
{
    float *inv = (float *)malloc(sizeof(float) * 2 * (size_t)n_feat), *shift = inv + n_feat;
    for (int j = 0; j < n_feat; ++j) {
        const float g = bn[j], be = bn[n_feat + j], mu = bn[2 * n_feat + j], var = bn[3 * n_feat + j];
        inv[j] = (1.0f / sqrtf(var + eps)) * g;
        float t = mu * inv[j];
        shift[j] = be - t;
    }
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        float *y = Y + i * ldy;
        for (int j = 0; j < n_feat; ++j) { float t = y[j] * inv[j]; y[j] = t + shift[j]; }
    }
    free(inv);
}

/* MLP described by flat arrays: dims[n_layers+1], acts[n_layers], W[l] -> [dims[l], dims[l+1]], b[l], bn or NULL */
void orc_mlp(int64_t n, int n_layers, const int32_t *dims, const int32_t *acts, const float *const *W,
             const float *const *b, const float *bn, float eps, const float *X, int64_t ldx, float *Y, int64_t ldy)
{
    int maxw = 0;
    for (int l = 1; l <= n_layers; ++l) maxw = dims[l] > maxw ? dims[l] : maxw;
    float *t0 = NULL, *t1 = NULL;
    if (n_layers > 1) {
        t0 = (float *)malloc(sizeof(float) * (size_t)n * maxw);
        t1 = (float *)malloc(sizeof(float) * (size_t)n * maxw);
    }
    const float *in = X;
    int64_t ldin = ldx;
    for (int l = 0; l < n_layers; ++l) {
        const int last = (l == n_layers - 1);
        float *out = last ? Y : ((l & 1) ? t1 : t0);
        int64_t ldo = last ? ldy : dims[l + 1];
        orc_dense(n, dims[l], dims[l + 1], in, ldin, W[l], b[l], acts[l], out, ldo);
        in = out;
        ldin = ldo;
    }
    if (bn) orc_batchnorm(n, dims[n_layers], Y, ldy, bn, eps);
    free(t0);
    free(t1);
}

/* GNN.py:206-215 per node: sqrt(sum (s-so)^2) > thr * sqrt(sum so^2); returns OR over nodes; flags optional.
 * so == NULL means "all ones" (GNN.py:266). */
int orc_not_converged(int64_t n, int d, const float *s, const float *so, float thr, uint8_t *flags)
{
    int any = 0;
#pragma omp parallel for schedule(static) reduction(| : any)
    for (int64_t i = 0; i < n; ++i) {
        float dist = 0.0f, nrm = 0.0f;
        for (int c = 0; c < d; ++c) {
            const float o = so ? so[i * d + c] : 1.0f;
            const float df = s[i * d + c] - o;
            const float dd = df * df;
            dist = dist + dd;
            const float oo = o * o;
            nrm = nrm + oo;
        }
        const float lhs = sqrtf(dist);
        const float rn = sqrtf(nrm);
        const float rhs = thr * rn;
        const int f = lhs > rhs;
        if (flags) flags[i] = (uint8_t)f;
        any |= f;
    }
    return any;
}

/*
 * GNNnodeBased.Loop (GNN.py:251-280), inference.  All sparse operands are the TRANSPOSED, row-major reordered
 * matrices of GraphTensor (graph_class.py:355-372) in CSR form sharing one indptr (both are "by destination").
 *   state_dim == 0  ->  state width = NL and state0 = nodes (GNN.py:265); state0 argument ignored.
 * Outputs: *k_out iterations (float, GNN.py:267), state_out [N, Ds], out_out [M, T], *m_out = M.
 * max_iter_override_rows lets bench time a bounded sample; it is NOT used by tests.
 */
int orc_loop(int64_t N, const int32_t *indptr, const int32_t *adj_src, const float *adj_w, const int32_t *arc_id,
             const float *arc_w, const float *nodes, int NL, const float *arc_labels, int AL, const uint8_t *mask,
             int state_dim, int st_layers, const int32_t *st_dims, const int32_t *st_acts, const float *const *st_W,
             const float *const *st_b, const float *st_bn, int out_layers, const int32_t *out_dims,
             const int32_t *out_acts, const float *const *out_W, const float *const *out_b, const float *out_bn,
             float bn_eps, int max_iter, float thr, const float *state0, float *k_out, float *state_out,
             float *out_out, int64_t *m_out, int n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const int Ds = state_dim ? state_dim : NL;
    const int NLc = state_dim ? NL : 0;            /* node labels appear in the concat only when D > 0 (GNN.py:229) */
    const int in_s = Ds + NLc + Ds + NLc + AL;     /* [state | nodes? | agg_states | agg_nodes | agg_arcs] GNN.py:237 */
    if (st_dims[0] != in_s || st_dims[st_layers] != Ds) return -1;
    if (out_dims[0] != Ds + NLc) return -2;
    const int T = out_dims[out_layers];

    float *inp = (float *)malloc(sizeof(float) * (size_t)N * in_s);
    float *s_cur = (float *)malloc(sizeof(float) * (size_t)N * Ds);
    float *s_old = (float *)malloc(sizeof(float) * (size_t)N * Ds);
    if (!inp || !s_cur || !s_old) return -3;

    /* loop-invariant column blocks of the concat are written once */
    const int c_nodes = Ds, c_aggs = Ds + NLc, c_aggn = c_aggs + Ds, c_agga = c_aggn + NLc;
    orc_spmm(N, indptr, arc_id, arc_w, arc_labels, AL, inp + c_agga, in_s);                 /* GNN.py:259 */
    if (state_dim) {
        orc_spmm(N, indptr, adj_src, adj_w, nodes, NL, inp + c_aggn, in_s);                 /* GNN.py:263 */
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < N; ++i)
            for (int c = 0; c < NL; ++c) inp[i * in_s + c_nodes + c] = nodes[i * NL + c];
        memcpy(s_cur, state0, sizeof(float) * (size_t)N * Ds);
    } else {
        memcpy(s_cur, nodes, sizeof(float) * (size_t)N * Ds);                               /* GNN.py:265 */
    }

    int k = 0;
    int go = orc_not_converged(N, Ds, s_cur, NULL, thr, NULL);                              /* vs ones, GNN.py:266 */
    while (go && k < max_iter) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < N; ++i) memcpy(inp + i * in_s, s_cur + i * Ds, sizeof(float) * Ds);
        orc_spmm(N, indptr, adj_src, adj_w, s_cur, Ds, inp + c_aggs, in_s);                 /* GNN.py:234 */
        float *t = s_old; s_old = s_cur; s_cur = t;                                         /* state_old <- state */
        orc_mlp(N, st_layers, st_dims, st_acts, st_W, st_b, st_bn, bn_eps, inp, in_s, s_cur, Ds);   /* GNN.py:240 */
        ++k;
        go = orc_not_converged(N, Ds, s_cur, s_old, thr, NULL);
    }
    *k_out = (float)k;
    memcpy(state_out, s_cur, sizeof(float) * (size_t)N * Ds);

    /* apply_filters + net_output (GNN.py:245-248, 275-279) */
    int64_t M = 0;
    for (int64_t i = 0; i < N; ++i) M += mask[i] ? 1 : 0;
    *m_out = M;
    if (M > 0 && out_out) {
        const int wf = Ds + NLc;
        float *feat = (float *)malloc(sizeof(float) * (size_t)M * wf);
        int64_t m = 0;
        for (int64_t i = 0; i < N; ++i) {
            if (!mask[i]) continue;
            memcpy(feat + m * wf, s_cur + i * Ds, sizeof(float) * Ds);
            if (NLc) memcpy(feat + m * wf + Ds, nodes + i * NL, sizeof(float) * NL);
            ++m;
        }
        orc_mlp(M, out_layers, out_dims, out_acts, out_W, out_b, out_bn, bn_eps, feat, wf, out_out, T);
        free(feat);
    }
    free(inp); free(s_cur); free(s_old);
    return 0;
}

/* GNN.py:331-332 / LGNN.py:278: out_graph[G, T] = NodeGraph^T[G, N] . out_nodes[N, T]; NodeGraph dense row-major [N, G].
 * Accumulation order: n ascending, fmaf. */
void orc_readout(int64_t N, int G, int T, const float *nodegraph, const float *out_nodes, float *out_graph)
{
    for (int g = 0; g < G; ++g)
        for (int t = 0; t < T; ++t) {
            float acc = 0.0f;
            for (int64_t i = 0; i < N; ++i) acc = fmaf(nodegraph[i * G + g], out_nodes[i * T + t], acc);
            out_graph[(int64_t)g * T + t] = acc;
        }
}

void orc_set_threads(int n)
{
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
