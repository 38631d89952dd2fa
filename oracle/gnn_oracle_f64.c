/*
 * Float64 SHADOW of the CPU oracle (plain C): the GNN.Loop state propagation of oracle/gnn_oracle.c evaluated in double.
 *
 * TEST INFRASTRUCTURE ONLY: the product (gnn_tf_2.x_amd/) never links, loads or calls this file.  tests/ use it as the arbiter for
 * rounding questions at sizes where the NumPy float64 shadow (oracle/gnn_oracle.py, dtype=np.float64) takes minutes: at BASELINE
 * size (1,000,000 nodes, 30 bodies) it says how far each float32 evaluation order is from the exact result.
 *
 * Same op sequence and citations as gnn_oracle.c (paths relative to the reference root):
 *   GNN/GNN.py:259,263   loop-invariant aggregates     GNN/GNN.py:202-220  condition (strict '>', state_old = ones first)
 *   GNN/GNN.py:223-242   convergence: SpMM -> concat -> net_state            GNN/GNN.py:245-248, :279  apply_filters + net_output
 *   GNN/MLP.py:11-64     Sequential = Dense(act) ... [+ BatchNormalization], Dropout = identity at inference
 * Inputs are the float32 arrays the float32 paths see (weights, labels, initial state), widened exactly; every operation after that is
 * IEEE double with libm's exp.  The evaluation order is irrelevant at this precision (tests compare at 1e-12 with the NumPy shadow).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

enum { ACT_LINEAR = 0, ACT_RELU = 1, ACT_SELU = 2, ACT_ELU = 3, ACT_TANH = 4, ACT_SIGMOID = 5, ACT_SOFTMAX = 6 };

static inline double act64(double v, int act)
{
    switch (act) {
    case ACT_RELU: return v > 0.0 ? v : 0.0;
    case ACT_SELU: return 1.0507009873554805 * (v > 0.0 ? v : 1.6732632423543772 * (exp(v) - 1.0));
    case ACT_ELU: return v > 0.0 ? v : exp(v) - 1.0;
    case ACT_TANH: return tanh(v);
    case ACT_SIGMOID: return 1.0 / (1.0 + exp(-v));
    default: return v;
    }
}

static void spmm64(int64_t n_rows, const int32_t *indptr, const int32_t *inner, const float *val, const double *dense, int width,
                   double *out, int64_t ld_out)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t r = 0; r < n_rows; ++r) {
        double *o = out + r * ld_out;
        for (int c = 0; c < width; ++c) o[c] = 0.0;
        for (int32_t e = indptr[r]; e < indptr[r + 1]; ++e) {
            const double w = (double)val[e];
            const double *x = dense + (int64_t)inner[e] * width;
            for (int c = 0; c < width; ++c) o[c] += w * x[c];
        }
    }
}

static void dense64(int64_t n, int n_in, int n_out, const double *X, int64_t ldx, const float *W, const float *b, int act, double *Y,
                    int64_t ldy)
{
    enum { RB = 4 };
#pragma omp parallel
    {
        double *acc = (double *)malloc(sizeof(double) * RB * (size_t)n_out);
#pragma omp for schedule(static)
        for (int64_t i0 = 0; i0 < n; i0 += RB) {
            const int rb = (int)((n - i0) < RB ? (n - i0) : RB);
            for (int t = 0; t < rb * n_out; ++t) acc[t] = 0.0;
            for (int k = 0; k < n_in; ++k) {
                const float *w = W + (size_t)k * n_out;
                for (int r = 0; r < rb; ++r) {
                    const double a = X[(i0 + r) * ldx + k];
                    double *ac = acc + r * n_out;
                    for (int j = 0; j < n_out; ++j) ac[j] += a * (double)w[j];
                }
            }
            for (int r = 0; r < rb; ++r) {
                double *y = Y + (i0 + r) * ldy, *ac = acc + r * n_out;
                for (int j = 0; j < n_out; ++j) ac[j] += (double)b[j];
                if (act == ACT_SOFTMAX) {
                    double m = ac[0], s = 0.0;
                    for (int j = 1; j < n_out; ++j) m = ac[j] > m ? ac[j] : m;
                    for (int j = 0; j < n_out; ++j) { ac[j] = exp(ac[j] - m); s += ac[j]; }
                    for (int j = 0; j < n_out; ++j) y[j] = ac[j] / s;
                } else {
                    for (int j = 0; j < n_out; ++j) y[j] = act64(ac[j], act);
                }
            }
        }
        free(acc);
    }
}

static void mlp64(int64_t n, int n_layers, const int32_t *dims, const int32_t *acts, const float *const *W, const float *const *b,
                  const float *bn, double eps, const double *X, int64_t ldx, double *Y, int64_t ldy)
{
    int maxw = 0;
    for (int l = 1; l <= n_layers; ++l) maxw = dims[l] > maxw ? dims[l] : maxw;
    double *t0 = NULL, *t1 = NULL;
    if (n_layers > 1) {
        t0 = (double *)malloc(sizeof(double) * (size_t)n * maxw);
        t1 = (double *)malloc(sizeof(double) * (size_t)n * maxw);
    }
    const double *in = X;
    int64_t ldin = ldx;
    for (int l = 0; l < n_layers; ++l) {
        const int last = (l == n_layers - 1);
        double *out = last ? Y : ((l & 1) ? t1 : t0);
        const int64_t ldo = last ? ldy : dims[l + 1];
        dense64(n, dims[l], dims[l + 1], in, ldin, W[l], b[l], acts[l], out, ldo);
        in = out;
        ldin = ldo;
    }
    if (bn) {     /* Keras BatchNormalization, inference: bn = [gamma | beta | mean | var] */
        const int f = dims[n_layers];
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < n; ++i)
            for (int j = 0; j < f; ++j) {
                const double inv = (1.0 / sqrt((double)bn[3 * f + j] + eps)) * (double)bn[j];
                Y[i * ldy + j] = Y[i * ldy + j] * inv + ((double)bn[f + j] - (double)bn[2 * f + j] * inv);
            }
    }
    free(t0);
    free(t1);
}

static int not_converged64(int64_t n, int d, const double *s, const double *so, double thr)
{
    int any = 0;
#pragma omp parallel for schedule(static) reduction(| : any)
    for (int64_t i = 0; i < n; ++i) {
        double dist = 0.0, nrm = 0.0;
        for (int c = 0; c < d; ++c) {
            const double o = so ? so[i * d + c] : 1.0, df = s[i * d + c] - o;
            dist += df * df;
            nrm += o * o;
        }
        any |= sqrt(dist) > thr * sqrt(nrm);
    }
    return any;
}

/* GNNnodeBased.Loop (GNN.py:251-280), inference, in double.  Arguments as orc_loop of gnn_oracle.c; state_out [N, Ds] and out_out [M, T]
 * are double.  thr is the float32 threshold widened (the reference's threshold is a Python float compared in float32). */
int orc_loop_f64(int64_t N, const int32_t *indptr, const int32_t *adj_src, const float *adj_w, const int32_t *arc_id, const float *arc_w,
                 const float *nodes, int NL, const float *arc_labels, int AL, const uint8_t *mask, int state_dim, int st_layers,
                 const int32_t *st_dims, const int32_t *st_acts, const float *const *st_W, const float *const *st_b, const float *st_bn,
                 int out_layers, const int32_t *out_dims, const int32_t *out_acts, const float *const *out_W, const float *const *out_b,
                 const float *out_bn, double bn_eps, int max_iter, float thr, const float *state0, float *k_out, double *state_out,
                 double *out_out, int64_t *m_out, int n_threads)
{
#ifdef _OPENMP
    if (n_threads > 0) omp_set_num_threads(n_threads);
#endif
    const int Ds = state_dim ? state_dim : NL, NLc = state_dim ? NL : 0, in_s = Ds + NLc + Ds + NLc + AL;
    if (st_dims[0] != in_s || st_dims[st_layers] != Ds) return -1;
    if (out_dims[0] != Ds + NLc) return -2;
    const int T = out_dims[out_layers];
    int64_t E = indptr[N];
    double *inp = (double *)malloc(sizeof(double) * (size_t)N * in_s);
    double *s_cur = (double *)malloc(sizeof(double) * (size_t)N * Ds), *s_old = (double *)malloc(sizeof(double) * (size_t)N * Ds);
    double *nodes64 = (double *)malloc(sizeof(double) * (size_t)N * NL), *arcl64 = (double *)malloc(sizeof(double) * (size_t)(E > 0 ? E : 1) * (AL > 0 ? AL : 1));
    if (!inp || !s_cur || !s_old || !nodes64 || !arcl64) return -3;
    for (int64_t i = 0; i < N * NL; ++i) nodes64[i] = (double)nodes[i];
    for (int64_t i = 0; i < E * AL; ++i) arcl64[i] = (double)arc_labels[i];
    const int c_nodes = Ds, c_aggs = Ds + NLc, c_aggn = c_aggs + Ds, c_agga = c_aggn + NLc;
    spmm64(N, indptr, arc_id, arc_w, arcl64, AL, inp + c_agga, in_s);
    if (state_dim) {
        spmm64(N, indptr, adj_src, adj_w, nodes64, NL, inp + c_aggn, in_s);
        for (int64_t i = 0; i < N; ++i)
            for (int c = 0; c < NL; ++c) inp[i * in_s + c_nodes + c] = nodes64[i * NL + c];
        for (int64_t i = 0; i < N * Ds; ++i) s_cur[i] = (double)state0[i];
    } else {
        for (int64_t i = 0; i < N * Ds; ++i) s_cur[i] = nodes64[i];
    }
    int k = 0;
    int go = not_converged64(N, Ds, s_cur, NULL, (double)thr);
    while (go && k < max_iter) {
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < N; ++i) memcpy(inp + i * in_s, s_cur + i * Ds, sizeof(double) * Ds);
        spmm64(N, indptr, adj_src, adj_w, s_cur, Ds, inp + c_aggs, in_s);
        double *t = s_old; s_old = s_cur; s_cur = t;
        mlp64(N, st_layers, st_dims, st_acts, st_W, st_b, st_bn, bn_eps, inp, in_s, s_cur, Ds);
        ++k;
        go = not_converged64(N, Ds, s_cur, s_old, (double)thr);
    }
    *k_out = (float)k;
    memcpy(state_out, s_cur, sizeof(double) * (size_t)N * Ds);
    int64_t M = 0;
    for (int64_t i = 0; i < N; ++i) M += mask[i] ? 1 : 0;
    *m_out = M;
    if (M > 0 && out_out) {
        const int wf = Ds + NLc;
        double *feat = (double *)malloc(sizeof(double) * (size_t)M * wf);
        int64_t m = 0;
        for (int64_t i = 0; i < N; ++i) {
            if (!mask[i]) continue;
            memcpy(feat + m * wf, s_cur + i * Ds, sizeof(double) * Ds);
            for (int c = 0; c < NLc; ++c) feat[m * wf + Ds + c] = nodes64[i * NL + c];
            ++m;
        }
        mlp64(M, out_layers, out_dims, out_acts, out_W, out_b, out_bn, bn_eps, feat, wf, out_out, T);
        free(feat);
    }
    free(inp); free(s_cur); free(s_old); free(nodes64); free(arcl64);
    return 0;
}
