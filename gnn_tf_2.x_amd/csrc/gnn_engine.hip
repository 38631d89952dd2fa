// gfx950 engine for the GNN fixed-point loop: handles, the per-op ("unfused") kernels and the loop driver.
//
// Every kernel here evaluates floating point in the order pinned by oracle/gnn_oracle.c (sequential fmaf chains,
// unfused BatchNormalization and norms), so results are bit-identical to that restatement.  The fused fast path is in
// gnn_fused.hip; this file is its fallback for shapes it does not cover and the GPU-side cross-check of it.
//
// Reference call sites replaced (paths relative to the reference root):
//   k_spmm        tf.sparse.sparse_dense_matmul   GNN/GNN.py:234, :259, :263
//   k_dense       Keras Dense + activation        GNN/MLP.py:62 (built), GNN/GNN.py:240, :279 (called)
//   k_softmax_bn  softmax activation + BatchNormalization (MLP.py:63)
//   k_check       condition()                     GNN/GNN.py:202-220
//   k_feats       apply_filters()                 GNN/GNN.py:245-248
//   k_readout     tf.matmul(nodegraph, out, transpose_a=True)   GNN/GNN.py:331-332, GNN/LGNN.py:278
//   k_relabel     LGNN.update_graph               GNN/LGNN.py:227-260
#include <dlfcn.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>

#include "gnn_common.h"

// ---------------------------------------------------------------------------------------------------------------------
// errors
// ---------------------------------------------------------------------------------------------------------------------
static thread_local char g_err[1024] = "";

int gnn_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *gnn_last_error(void) { return g_err; }
extern "C" int gnn_version(void) { return 1; }

extern "C" int gnn_device_count(int *count)
{
    ARGCHK(count, "count is NULL");
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return GNN_OK;
}

extern "C" int gnn_device_synchronize(int device)
{
    HIPCHK(hipSetDevice(device));
    HIPCHK(hipDeviceSynchronize());
    return GNN_OK;
}

bool gnn_poison_enabled()
{
#ifdef GNN_DIAG
    static const bool on = [] {
        const bool v = getenv("GNN_POISON") != nullptr && atoi(getenv("GNN_POISON")) != 0;
        if (v) fprintf(stderr, "libgnn_hip (diagnostic build): GNN_POISON=1 - every device allocation and the fused kernels' LDS are filled with NaN before use\n");
        return v;
    }();
    return on;
#else
    return false;
#endif
}

hipError_t gnn_dev_malloc(void **p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
#ifdef GNN_DIAG
    if (e == hipSuccess && gnn_poison_enabled()) {
        // null-stream fill + device-wide wait: complete before ANY stream of the engine can touch the block (diagnostic build only)
        e = hipMemset(*p, 0xFF, bytes);
        if (e == hipSuccess) e = hipDeviceSynchronize();
    }
#endif
    return e;
}

template <typename T>
static int dev_alloc(T **p, size_t count)
{
    *p = nullptr;
    if (count == 0) count = 1;
    HIPCHK(gnn_dev_malloc((void **)p, count * sizeof(T)));
    return GNN_OK;
}

template <typename T>
static int dev_upload(T **p, const T *host, size_t count)
{
    int rc = dev_alloc(p, count);
    if (rc) return rc;
    if (count) HIPCHK(hipMemcpy(*p, host, count * sizeof(T), hipMemcpyHostToDevice));
    return GNN_OK;
}

// Creation-time fills are STREAM-ORDERED (round 3).  hipMemset on device memory is queued on the null stream and returns before
// the fill has run (tools/memset_probe.hip); the loops work on hipStreamNonBlocking streams, which the null stream does not order,
// so a fill queued at creation time may land AFTER data that such a stream wrote later (tools/memset_race_probe.hip reproduces
// it: derive -> relabel).  Hence:
//   * buffers of a handle that has a stream (gnn_loop: state ping-pong, slice aggregate) are zeroed with hipMemsetAsync on THAT
//     stream - every later kernel / copy of the handle is behind the fill by stream order, nothing waits on the host;
//   * buffers of a handle without a stream (derived graphs: labels) are zeroed on the engine's per-device fill stream and the
//     handle keeps an event; every stream that touches the labels first waits for it ON THE DEVICE (gnn_graph_wait_ready:
//     hipStreamWaitEvent), host readers synchronise on the event.  No device-wide synchronisation anywhere.
static hipStream_t g_fill_stream[64] = {nullptr};

static int fill_stream(int device, hipStream_t *st)
{
    if (device < 0 || device >= 64) return gnn_fail(GNN_ERR_ARG, "device %d out of range", device);
    if (!g_fill_stream[device]) HIPCHK(hipStreamCreateWithFlags(&g_fill_stream[device], hipStreamNonBlocking));
    *st = g_fill_stream[device];
    return GNN_OK;
}

static int zero_on_stream(void *p, size_t bytes, hipStream_t st)
{
    if (!bytes) return GNN_OK;
    HIPCHK(hipMemsetAsync(p, 0, bytes, st));
    return GNN_OK;
}

// queue the zero fill of a fresh graph-owned buffer and (re)record the graph's ready event behind it
static int graph_zero_fill(gnn_graph *g, void *p, size_t bytes)
{
    hipStream_t st = nullptr;
    int rc = fill_stream(g->device, &st);
    if (rc) return rc;
#ifdef GNN_DIAG      // diagnostic build only: the creation-time fill as it was before round 3 (null stream, unordered) - exists to show that
    // tests/test_gpu_full_size.py::test_relabelling_is_ordered_behind_the_creation_fill fails without the ordering
    static const bool legacy = getenv("GNN_LEGACY_NULL_MEMSET") != nullptr;
    if (legacy) { HIPCHK(hipMemset(p, 0, bytes)); return GNN_OK; }
#endif
    if (!g->ready) HIPCHK(hipEventCreateWithFlags(&g->ready, hipEventDisableTiming));
    if ((rc = zero_on_stream(p, bytes, st))) return rc;
    HIPCHK(hipEventRecord(g->ready, st));
    return GNN_OK;
}

// device-side wait: work queued on `st` after this call runs after the graph's creation-time fills
int gnn_graph_wait_ready(const gnn_graph *g, hipStream_t st)
{
    if (g && g->ready) HIPCHK(hipStreamWaitEvent(st, g->ready, 0));
    return GNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// kernels
// ---------------------------------------------------------------------------------------------------------------------

// out[r, :] = sum over the stored entries e of row r, in order, of w[e] * X[src(e), :]     (fmaf chain)
// lpr lanes (power of two <= 64) cooperate on a row, each owning VEC consecutive columns per chunk.
template <int VEC, bool INDEXED, int U = 4>
__global__ void __launch_bounds__(256) k_spmm(int64_t n_rows, const int32_t *__restrict__ indptr,
                                              const int32_t *__restrict__ idx, const float *__restrict__ w,
                                              const float *__restrict__ X, int width, int64_t ldx,
                                              float *__restrict__ out, int64_t ldo, int lpr, const int *gate, int world)
{
    if (!gnn_gate_open_block(gate, world)) return;
    const int64_t gtid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t rows_per_pass = ((int64_t)gridDim.x * blockDim.x) / lpr;
    const int lane = (int)(gtid % lpr);
    for (int64_t row = gtid / lpr; row < n_rows; row += rows_per_pass) {
        const int beg = indptr[row], end = indptr[row + 1];
        for (int c0 = lane * VEC; c0 < width; c0 += lpr * VEC) {
            float acc[VEC];
#pragma unroll
            for (int v = 0; v < VEC; ++v) acc[v] = 0.0f;
            if constexpr (U == 4) {
                int e = beg;
                for (; e + 4 <= end; e += 4) {
                    float we[4];
                    float xv[4][VEC];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        we[u] = w[e + u];
                        const int64_t s = INDEXED ? (int64_t)idx[e + u] : (int64_t)(e + u);
                        const float *xp = X + s * ldx + c0;
                        if constexpr (VEC == 4) {
                            const float4 t = *reinterpret_cast<const float4 *>(xp);
                            xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
                        } else {
                            xv[u][0] = xp[0];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(we[u], xv[u][v], acc[v]);
                }
                for (; e < end; ++e) {
                    const float we = w[e];
                    const int64_t s = INDEXED ? (int64_t)idx[e] : (int64_t)e;
                    const float *xp = X + s * ldx + c0;
#pragma unroll
                    for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(we, xp[v], acc[v]);
                }
            } else {
                // narrow slices (feature-sliced exchange, 8 / 16 columns per rank): U entries requested together, the tail as a masked batch
                // (clamped to a real entry, result unused) instead of one dependent round trip per entry
                for (int e = beg; e < end; e += U) {
                    float we[U];
                    float xv[U][VEC];
#pragma unroll
                    for (int u = 0; u < U; ++u) {
                        const int ee = e + u < end ? e + u : e;
                        we[u] = w[ee];
                        const int64_t s = INDEXED ? (int64_t)idx[ee] : (int64_t)ee;
                        const float *xp = X + s * ldx + c0;
                        if constexpr (VEC == 4) {
                            const float4 t = *reinterpret_cast<const float4 *>(xp);
                            xv[u][0] = t.x; xv[u][1] = t.y; xv[u][2] = t.z; xv[u][3] = t.w;
                        } else {
                            xv[u][0] = xp[0];
                        }
                    }
#pragma unroll
                    for (int u = 0; u < U; ++u)
                        if (e + u < end) {
#pragma unroll
                            for (int v = 0; v < VEC; ++v) acc[v] = __builtin_fmaf(we[u], xv[u][v], acc[v]);
                        }
                }
            }
            float *op = out + row * ldo + c0;
#pragma unroll
            for (int v = 0; v < VEC; ++v) op[v] = acc[v];
        }
    }
}

// Y[i, j] = epilogue( sum_k fmaf(X[i,k], W[k,j]) + b[j] );  R rows per block staged in LDS, one thread per column.
template <int R>
__global__ void __launch_bounds__(256) k_dense(int64_t n, int n_in, int n_in_pad, int n_out, const float *__restrict__ X,
                                               int64_t ldx, const float *__restrict__ W, const float *__restrict__ b,
                                               int act, const float *__restrict__ bn_scale,
                                               const float *__restrict__ bn_shift, float *__restrict__ Y, int64_t ldy,
                                               const int *gate, int world)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    if (!gnn_gate_open_block(gate, world)) return;
    const int64_t i0 = (int64_t)blockIdx.x * R;
    for (int t = threadIdx.x; t < R * n_in_pad; t += blockDim.x) {
        const int r = t / n_in_pad, k = t - r * n_in_pad;
        xs[t] = (k < n_in && i0 + r < n) ? X[(i0 + r) * ldx + k] : 0.0f;
    }
    __syncthreads();
    for (int j = threadIdx.x; j < n_out; j += blockDim.x) {
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0f;
        int k = 0;
        for (; k + 4 <= n_in; k += 4) {
            const float w0 = W[(size_t)(k + 0) * n_out + j], w1 = W[(size_t)(k + 1) * n_out + j];
            const float w2 = W[(size_t)(k + 2) * n_out + j], w3 = W[(size_t)(k + 3) * n_out + j];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                const float4 x = *reinterpret_cast<const float4 *>(&xs[r * n_in_pad + k]);
                acc[r] = __builtin_fmaf(x.x, w0, acc[r]);
                acc[r] = __builtin_fmaf(x.y, w1, acc[r]);
                acc[r] = __builtin_fmaf(x.z, w2, acc[r]);
                acc[r] = __builtin_fmaf(x.w, w3, acc[r]);
            }
        }
        for (; k < n_in; ++k) {
            const float wk = W[(size_t)k * n_out + j];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(xs[r * n_in_pad + k], wk, acc[r]);
        }
        const float bj = b[j];
        const float sc = bn_scale ? bn_scale[j] : 1.0f, sh = bn_shift ? bn_shift[j] : 0.0f;
#pragma unroll
        for (int r = 0; r < R; ++r) {
            if (i0 + r >= n) break;
            float v = acc[r] + bj;
            if (act != GNN_ACT_SOFTMAX) {
                v = gnn_act(v, act);
                if (bn_scale) { float t = v * sc; v = t + sh; }
            }
            Y[(i0 + r) * ldy + j] = v;
        }
    }
}

// in-place row softmax (+ trailing BatchNormalization), one thread per row, sequential in j like the oracle
__global__ void k_softmax_bn(int64_t n, int n_out, float *Y, int64_t ldy, const float *bn_scale, const float *bn_shift,
                             const int *gate, int world)
{
    if (!gnn_gate_open_block(gate, world)) return;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float *y = Y + i * ldy;
    float m = y[0];
    for (int j = 1; j < n_out; ++j) m = y[j] > m ? y[j] : m;
    float s = 0.0f;
    for (int j = 0; j < n_out; ++j) { const float e = gnn_expf(y[j] - m); y[j] = e; s = s + e; }
    for (int j = 0; j < n_out; ++j) {
        float v = __fdiv_rn(y[j], s);
        if (bn_scale) { float t = v * bn_scale[j]; v = t + bn_shift[j]; }
        y[j] = v;
    }
}

// condition(): per owned node sqrt(sum (s-so)^2) > thr * sqrt(sum so^2); OR into the rank's flag.  so == nullptr: ones.
// One thread per row keeps the oracle's ascending-feature summation order; the rows of a block are staged through LDS in
// chunks of 32 features so that HBM is read in full 128-byte segments.
__global__ void __launch_bounds__(256) k_check(int64_t n_rows, int d, const float *__restrict__ s,
                                               const float *__restrict__ so, float thr, int *flag_out, const int *gate,
                                               int world)
{
    extern __shared__ float chk_sh[];                      // ts [256 * 33] (+ to [256 * 33] iff so)
    float *ts = chk_sh, *to = chk_sh + 256 * 33;
    if (!gnn_gate_open_block(gate, world)) return;
    const int64_t r0 = (int64_t)blockIdx.x * 256;
    const int64_t i = r0 + threadIdx.x;
    float dist = 0.0f, nrm = 0.0f;
    const bool v4 = (d & 3) == 0;
    for (int c0 = 0; c0 < d; c0 += 32) {
        const int cw = (d - c0) < 32 ? (d - c0) : 32;
        __syncthreads();
        if (v4) {
#pragma unroll
            for (int t = threadIdx.x; t < 256 * 8; t += 256) {
                const int r = t >> 3, c = (t & 7) * 4;
                if (c < cw && r0 + r < n_rows) {
                    const float4 a = *reinterpret_cast<const float4 *>(s + (r0 + r) * d + c0 + c);
                    float *p = ts + r * 33 + c;
                    p[0] = a.x; p[1] = a.y; p[2] = a.z; p[3] = a.w;
                    if (so) {
                        const float4 b = *reinterpret_cast<const float4 *>(so + (r0 + r) * d + c0 + c);
                        float *q = to + r * 33 + c;
                        q[0] = b.x; q[1] = b.y; q[2] = b.z; q[3] = b.w;
                    }
                }
            }
        } else {
            for (int t = threadIdx.x; t < 256 * 32; t += 256) {
                const int r = t >> 5, c = t & 31;
                if (c < cw && r0 + r < n_rows) {
                    ts[r * 33 + c] = s[(r0 + r) * d + c0 + c];
                    if (so) to[r * 33 + c] = so[(r0 + r) * d + c0 + c];
                }
            }
        }
        __syncthreads();
        if (i < n_rows)
            for (int c = 0; c < cw; ++c) {
                const float o = so ? to[threadIdx.x * 33 + c] : 1.0f;
                const float df = ts[threadIdx.x * 33 + c] - o;
                const float dd = df * df;
                dist = dist + dd;
                const float oo = o * o;
                nrm = nrm + oo;
            }
    }
    int f = 0;
    if (i < n_rows) {
        const float lhs = sqrtf(dist);
        const float rn = sqrtf(nrm);
        const float rhs = thr * rn;
        f = lhs > rhs;
    }
    if (__any(f) && (threadIdx.x & 63) == 0) gnn_flag_raise(flag_out);
}

// k_final = number of executed bodies = first k whose gate is closed (or max_iter); one wave, 64 gates per pass
__global__ void k_finalize(const int *flags, int world, int max_iter, int *kfinal)
{
    const int lane = threadIdx.x;
    int k_final = max_iter;
    for (int k0 = 0; k0 < max_iter; k0 += 64) {
        const int k = k0 + lane;
        const bool closed = k < max_iter && !gnn_gate_open(flags + (size_t)k * world * GNN_FLAG_WORDS, world);
        const unsigned long long m = __ballot(closed);
        if (m) { k_final = k0 + __builtin_ctzll(m); break; }
    }
    if (lane == 0) *kfinal = k_final;
    // certified gate of the split-arithmetic path (gnn_common.h, gnn_flag_raise_certified): gates 1 .. k_final that decided this run (the
    // gate of body max_iter is never consulted; gate 0 is the first condition, the same arithmetic on every path).  Not certified: no node
    // moved robustly AND some node was borderline.  The exact paths never raise words 1 / 2, so this stays 0 for them.
    int amb = 0;
    const int last = k_final < max_iter ? k_final : max_iter - 1;
    for (int k = 1 + lane; k <= last; k += 64) {
        const int *gate = flags + (size_t)k * world * GNN_FLAG_WORDS;
        int robust = 0, border = 0;
#pragma unroll 16      // (independent loads: sixteen slots' words in flight per lane instead of one dependent round trip per slot)
        for (int p = 0; p < world * GNN_FLAG_SLOTS; ++p) { robust |= gate[p * GNN_FLAG_STRIDE + 1]; border |= gate[p * GNN_FLAG_STRIDE + 2]; }
        amb |= (!robust && border) ? 1 : 0;
    }
    amb = __any(amb) ? 1 : 0;
    if (lane == 0) kfinal[2] = amb;
}

// apply_filters(): feats[m] = [state_final[row_m] | nodes[row_m] (iff D > 0)]
__global__ void k_feats(int64_t n_masked, const int32_t *__restrict__ masked_rows, const float *s0, const float *s1,
                        const int *kfinal, int Ds, const float *__restrict__ nodes_own, int NL, int NLc,
                        float *__restrict__ feats)
{
    const int wf = Ds + NLc;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_masked * wf) return;
    const float *state = ((*kfinal) & 1) ? s1 : s0;
    const int64_t m = t / wf;
    const int c = (int)(t - m * wf);
    const int64_t row = masked_rows[m];
    feats[t] = c < Ds ? state[row * Ds + c] : nodes_own[row * NL + (c - Ds)];
}

// GNNedgeBased.apply_filters(): feats[m] = [F[dst(e)] | F[src(e)] | arc_labels[e]], e = m-th masked arc, F = [state | nodes?]
__global__ void k_feats_edge(int64_t n_masked, const int32_t *__restrict__ rows, const int32_t *__restrict__ entry_dst,
                             const int32_t *__restrict__ adj_src, const float *s0, const float *s1, const int *kfinal, int Ds,
                             const float *__restrict__ nodes, int NL, int NLc, const float *__restrict__ arc_labels, int AL,
                             float *__restrict__ feats, int64_t own_off)
{
    const int wn = Ds + NLc, wf = 2 * wn + AL;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_masked * wf) return;
    const float *state = ((*kfinal) & 1) ? s1 : s0;
    const int64_t m = t / wf;
    int c = (int)(t - m * wf);
    const int64_t e = rows[m];
    float v;
    if (c < 2 * wn) {
        const int64_t node = c < wn ? own_off + entry_dst[e] : adj_src[e];      // destination: owned row -> row of the replica; source: already in the replica's index space
        if (c >= wn) c -= wn;
        v = c < Ds ? state[node * Ds + c] : nodes[node * NL + (c - Ds)];
    } else {
        v = arc_labels[e * AL + (c - 2 * wn)];
    }
    feats[t] = v;
}

// apply_filters + a ONE-layer net_output with few outputs (the usual classifier head, T <= 8) in one pass over the masked
// rows.  A block stages 64 feature rows [state | labels] through LDS with coalesced reads, then thread (row, j) runs the
// k-ordered fmaf chain of output j (same order as k_dense); softmax / activation / BatchNormalization as k_softmax_bn.
// Saves materialising [M, NL + D] features and two more launches.
#define GNN_OUT1_ROWS 64
__global__ void k_out1(int64_t n_masked, const int32_t *__restrict__ masked_rows, const float *s0, const float *s1,
                       const int *kfinal, int Ds, const float *__restrict__ nodes_own, int NL, int NLc,
                       const float *__restrict__ W, const float *__restrict__ b, int T, int act,
                       const float *__restrict__ bn_scale, const float *__restrict__ bn_shift, float *__restrict__ out)
{
    extern __shared__ float osh[];
    const int wf = Ds + NLc, ldw = wf | 1;
    float *wsh = osh;                                      // W [wf, T] then b [T]
    float *tile = wsh + (wf + 1) * T;                      // [64, ldw]
    float *vsh = tile + GNN_OUT1_ROWS * ldw;               // [64, T]
    const int nthr = blockDim.x, tid = threadIdx.x;
    const float *state = ((*kfinal) & 1) ? s1 : s0;
    const int64_t base = (int64_t)blockIdx.x * GNN_OUT1_ROWS;
    for (int t = tid; t < (wf + 1) * T; t += nthr) wsh[t] = t < wf * T ? W[t] : b[t - wf * T];
    if ((Ds & 3) == 0) {                                   // 16-byte pieces of the state rows, many rows in flight per thread
        const int q = Ds >> 2;
#pragma unroll 4
        for (int idx = tid; idx < GNN_OUT1_ROWS * q; idx += nthr) {
            const int r = idx / q, c = (idx - r * q) * 4;
            if (base + r < n_masked) {
                const float4 v = *reinterpret_cast<const float4 *>(state + (int64_t)masked_rows[base + r] * Ds + c);
                float *t = tile + r * ldw + c;
                t[0] = v.x; t[1] = v.y; t[2] = v.z; t[3] = v.w;
            }
        }
    } else {
        for (int idx = tid; idx < GNN_OUT1_ROWS * Ds; idx += nthr) {
            const int r = idx / Ds, c = idx - r * Ds;
            if (base + r < n_masked) tile[r * ldw + c] = state[(int64_t)masked_rows[base + r] * Ds + c];
        }
    }
    for (int idx = tid; idx < GNN_OUT1_ROWS * NLc; idx += nthr) {
        const int r = idx / NLc, c = idx - r * NLc;
        if (base + r < n_masked) tile[r * ldw + Ds + c] = nodes_own[(int64_t)masked_rows[base + r] * NL + c];
    }
    __syncthreads();
    const int r = tid / T, j = tid - r * T;
    const int64_t m = base + r;
    const bool live = r < GNN_OUT1_ROWS && m < n_masked;
    if (live) {
        float acc = 0.0f;
        const float *x = tile + r * ldw;
        for (int k = 0; k < wf; ++k) acc = __builtin_fmaf(x[k], wsh[k * T + j], acc);
        vsh[r * T + j] = acc + wsh[wf * T + j];
    }
    __syncthreads();
    if (!live) return;
    const float *y = vsh + r * T;
    float v;
    if (act == GNN_ACT_SOFTMAX) {
        float mx = y[0];
        for (int q = 1; q < T; ++q) mx = y[q] > mx ? y[q] : mx;
        float sum = 0.0f, mine = 0.0f;
        for (int q = 0; q < T; ++q) { const float e = gnn_expf(y[q] - mx); sum = sum + e; if (q == j) mine = e; }
        v = __fdiv_rn(mine, sum);
    } else
        v = gnn_act(y[j], act);
    if (bn_scale) { const float t2 = v * bn_scale[j]; v = t2 + bn_shift[j]; }
    out[m * T + j] = v;
}

// graph readout: out_graph[g, t] = sum over the stored (node, w) of graph g, ascending node, fmaf(w, out_nodes[node, t])
// Sharded: a rank sums the nodes it owns ([row_begin, row_begin + n_rows), out_nodes indexed from row_begin); the partial
// results are then added in rank order by k_sum_partials (exact when no graph straddles two shards: x + 0 == x).
__global__ void k_readout(int G, int T, const int32_t *__restrict__ indptr, const int32_t *__restrict__ node,
                          const float *__restrict__ w, const float *__restrict__ out_nodes, int64_t row_begin, int64_t n_rows,
                          float *__restrict__ out_graph)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= G * T) return;
    const int g = t / T, c = t - g * T;
    float acc = 0.0f;
    for (int e = indptr[g]; e < indptr[g + 1]; ++e) {
        const int64_t i = (int64_t)node[e] - row_begin;
        if (i >= 0 && i < n_rows) acc = __builtin_fmaf(w[e], out_nodes[i * T + c], acc);
    }
    out_graph[t] = acc;
}

__global__ void k_sum_partials(int count, int world, const float *__restrict__ partial, float *__restrict__ out)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    float acc = partial[t];
    for (int p = 1; p < world; ++p) acc = acc + partial[(size_t)p * count + t];
    out[t] = acc;
}

// LGNN.update_graph: dst[i] = [base[i, :NLb] | state[i] (if get_state) | mask[i] ? out[pos(i)] : 0 (if get_output)]
__global__ void k_relabel(int64_t N, int NLb, const float *__restrict__ base_nodes, int Ds, const float *s0,
                          const float *s1, const int *kfinal, int get_state, int T, const float *__restrict__ out,
                          const uint8_t *__restrict__ mask, const int32_t *__restrict__ mask_pos, int get_output,
                          float *__restrict__ dst, int NLd)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N * NLd) return;
    const int64_t i = t / NLd;
    int c = (int)(t - i * NLd);
    float v;
    if (c < NLb) {
        v = base_nodes[i * NLb + c];
    } else {
        c -= NLb;
        if (get_state && c < Ds) {
            const float *state = ((*kfinal) & 1) ? s1 : s0;
            v = state[i * Ds + c];
        } else {
            if (get_state) c -= Ds;
            v = mask[i] ? out[(int64_t)mask_pos[i] * T + c] : 0.0f;
        }
    }
    dst[t] = v;
}

// own RNG for the initial state when none is injected (tf.random.normal(stddev=0.1), GNN.py:262, cannot be matched)
__device__ __forceinline__ uint64_t splitmix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// arc side of LGNN.update_graph (LGNN.py:253-254), original arc order: dst[p] = [base labels of arc p | 0 ...]; the output rows
// are then scattered over the masked positions by k_arc_scatter
__global__ void k_arc_base(int64_t E, int ALb, const float *__restrict__ base, int ALd, float *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * ALd) return;
    const int64_t p = t / ALd;
    const int c = (int)(t - p * ALd);
    dst[t] = c < ALb ? base[p * ALb + c] : 0.0f;
}

__global__ void k_arc_scatter(int64_t M, int T, const int32_t *__restrict__ rows, const float *__restrict__ out, int ALb, int ALd, float *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= M * T) return;
    const int64_t m = t / T;
    const int c = (int)(t - m * T);
    dst[(int64_t)rows[m] * ALd + ALb + c] = out[t];
}

// ArcNode^T order: entry q carries the labels of arc arc_id[q]
__global__ void k_arc_permute(int64_t E, int AL, const int32_t *__restrict__ arc_id, const float *__restrict__ orig, float *__restrict__ csr)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= E * AL) return;
    const int64_t q = t / AL;
    const int c = (int)(t - q * AL);
    csr[t] = orig[(int64_t)arc_id[q] * AL + c];
}

__global__ void k_randn(int64_t count, int64_t offset, uint64_t seed, float stddev, float *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    const uint64_t h = splitmix64(seed ^ splitmix64((uint64_t)(t + offset)));
    const float u1 = ((float)(uint32_t)(h >> 40) + 1.0f) * (1.0f / 16777217.0f);
    const float u2 = (float)(uint32_t)((h >> 8) & 0xFFFFFF) * (1.0f / 16777216.0f);
    out[t] = stddev * sqrtf(-2.0f * logf(u1)) * cosf(6.28318530717958647692f * u2);
}

__global__ void k_fill(int64_t count, float v, float *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < count) out[t] = v;
}

// ---------------------------------------------------------------------------------------------------------------------
// launch helpers
// ---------------------------------------------------------------------------------------------------------------------
static inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

int gnn_launch_spmm(hipStream_t st, int64_t n_rows, const int32_t *indptr, const int32_t *idx, const float *w,
                       const float *X, int width, int64_t ldx, float *out, int64_t ldo, const int *gate, int world)
{
    if (n_rows == 0 || width == 0) return GNN_OK;
    const bool vec4 = (width % 4 == 0) && (ldx % 4 == 0) && (((uintptr_t)X & 15) == 0);
    const int vec = vec4 ? 4 : 1;
    int lpr = 1;
    while (lpr * vec < width && lpr < 64) lpr <<= 1;
    const int64_t threads = n_rows * lpr;
    const unsigned grid = (unsigned)std::min<int64_t>(cdiv(threads, 256), 256 * 32);
    const bool indexed = idx != nullptr;
    // (A one-lane-per-row form with eight entries in flight for the 8-column slices of the feature-sliced exchange was measured in round 3:
    //  257 us against 203 us for this kernel's two lanes per row - profiles/r03_exchange_layouts.txt - and removed.  Round 5: see below.)
    // narrow column slices (feature-sliced exchange at 4 / 8 ranks: 16 / 8 columns of all nodes): eight entries in flight per lane and the tail as a
    // masked batch - 183 against 204 us per rank and iteration at P = 8 (profiles/r05_slice_counters.txt).  The kernel is bound by the L2's miss path:
    // 11.4 M 64-byte fetches per launch for 10 M random 32-byte row pieces out of a 32 MB table that no XCD's 4 MiB L2 can hold (hit rate 23 %).
    if (vec4 && indexed && width <= 16) hipLaunchKernelGGL((k_spmm<4, true, 8>), grid, 256, 0, st, n_rows, indptr, idx, w, X, width, ldx, out, ldo, lpr, gate, world);
    else if (vec4 && indexed) hipLaunchKernelGGL((k_spmm<4, true>), grid, 256, 0, st, n_rows, indptr, idx, w, X, width, ldx, out, ldo, lpr, gate, world);
    else if (vec4) hipLaunchKernelGGL((k_spmm<4, false>), grid, 256, 0, st, n_rows, indptr, idx, w, X, width, ldx, out, ldo, lpr, gate, world);
    else if (indexed) hipLaunchKernelGGL((k_spmm<1, true>), grid, 256, 0, st, n_rows, indptr, idx, w, X, width, ldx, out, ldo, lpr, gate, world);
    else hipLaunchKernelGGL((k_spmm<1, false>), grid, 256, 0, st, n_rows, indptr, idx, w, X, width, ldx, out, ldo, lpr, gate, world);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

// one Dense layer without activation epilogue choices hidden: Y = act(X . W + b) [then x * scale + shift]; used by the
// training path (gnn_train.hip) for z = h . W + b and d h = d z . W^T
int gnn_launch_dense(hipStream_t st, int64_t n, int n_in, int n_out, const float *X, int64_t ldx, const float *W, const float *b,
                     int act, float *Y, int64_t ldy)
{
    if (n == 0) return GNN_OK;
    constexpr int R = 8;
    const int n_in_pad = (n_in + 3) & ~3;
    const size_t lds = sizeof(float) * R * n_in_pad;
    if (lds > 64 * 1024) return gnn_fail(GNN_ERR_UNSUPPORTED, "layer input width %d too large", n_in);
    if (act == GNN_ACT_SOFTMAX) return gnn_fail(GNN_ERR_ARG, "gnn_launch_dense: softmax is applied by the caller");
    const int threads = std::min(256, ((n_out + 63) / 64) * 64);
    hipLaunchKernelGGL((k_dense<R>), cdiv(n, R), threads, lds, st, n, n_in, n_in_pad, n_out, X, ldx, W, b, act, (const float *)nullptr,
                       (const float *)nullptr, Y, ldy, (const int *)nullptr, 1);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

static int launch_check(hipStream_t st, int64_t n_rows, int d, const float *s, const float *so, float thr, int *flag_out,
                        const int *gate, int world)
{
    if (n_rows == 0) return GNN_OK;
    // two staged tiles are 66 KB of dynamic LDS: the limit is a per-device attribute of the kernel, raised once per device
    static bool big_lds[64] = {false};
    int dev = 0;
    HIPCHK(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64 || !big_lds[dev]) {
        HIPCHK(hipFuncSetAttribute(reinterpret_cast<const void *>(k_check), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * 256 * 33 * 4));
        if (dev >= 0 && dev < 64) big_lds[dev] = true;
    }
    hipLaunchKernelGGL(k_check, cdiv(n_rows, 256), 256, sizeof(float) * 256 * 33 * (so ? 2 : 1), st, n_rows, d, s, so, thr, flag_out, gate, world);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

int gnn_launch_feats_edge(hipStream_t st, const gnn_loop *l, const float *state, float *feats)
{
    if (!l->n_edge_masked) return GNN_OK;
    const gnn_graph *g = l->g;
    const int64_t tot = l->n_edge_masked * l->ou->dims[0];
    hipLaunchKernelGGL(k_feats_edge, cdiv(tot, 256), 256, 0, st, l->n_edge_masked, l->edge_rows, l->edge_dst, g->sh->adj_src, state, state,
                       l->kfinal_dev, l->Ds, g->nodes, g->NL, l->NLc, g->arc_labels_orig_own ? g->arc_labels_orig_own : l->edge_labels, g->AL, feats, l->own_off);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

int gnn_launch_check(hipStream_t st, int64_t n_rows, int d, const float *s, const float *so, float thr, int *flag_rank_base)
{
    return launch_check(st, n_rows, d, s, so, thr, flag_rank_base, nullptr, 1);
}

// Sequential forward on device buffers: X[n, dims[0]] (ldx) -> Y[n, dims.back()] (ldy); t0/t1: [n, max hidden width]
static int launch_mlp(hipStream_t st, const gnn_mlp *m, int64_t n, const float *X, int64_t ldx, float *Y, int64_t ldy,
                      float *t0, float *t1, const int *gate, int world)
{
    if (n == 0) return GNN_OK;
    const float *in = X;
    int64_t ldin = ldx;
    constexpr int R = 8;
    for (int l = 0; l < m->n_layers; ++l) {
        const bool last = l == m->n_layers - 1;
        float *out = last ? Y : ((l & 1) ? t1 : t0);
        const int64_t ldo = last ? ldy : m->dims[l + 1];
        const int n_in = m->dims[l], n_out = m->dims[l + 1], act = m->acts[l];
        const int n_in_pad = (n_in + 3) & ~3;
        const size_t lds = sizeof(float) * R * n_in_pad;
        if (lds > 64 * 1024) return gnn_fail(GNN_ERR_UNSUPPORTED, "layer input width %d too large", n_in);
        const int threads = std::min(256, ((n_out + 63) / 64) * 64);
        const float *sc = (last && m->has_bn) ? m->bn_scale : nullptr, *sh = (last && m->has_bn) ? m->bn_shift : nullptr;
        hipLaunchKernelGGL((k_dense<R>), cdiv(n, R), threads, lds, st, n, n_in, n_in_pad, n_out, in, ldin, m->W[l], m->b[l],
                           act, sc, sh, out, ldo, gate, world);
        HIPCHK(hipGetLastError());
        if (act == GNN_ACT_SOFTMAX) {
            hipLaunchKernelGGL(k_softmax_bn, cdiv(n, 256), 256, 0, st, n, n_out, out, ldo, sc, sh, gate, world);
            HIPCHK(hipGetLastError());
        }
        in = out;
        ldin = ldo;
    }
    return GNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// graph
// ---------------------------------------------------------------------------------------------------------------------
static void graph_release_shared(gnn_graph_shared *sh)
{
    if (!sh || --sh->refs > 0) return;
    (void)hipFree(sh->indptr); (void)hipFree(sh->adj_src); (void)hipFree(sh->masked_rows);
    (void)hipFree(sh->adj_w); (void)hipFree(sh->arc_w); (void)hipFree(sh->arc_labels); (void)hipFree(sh->mask);
    (void)hipFree(sh->src_indptr); (void)hipFree(sh->src_dst); (void)hipFree(sh->src_w);
    (void)hipFree(sh->arc_id); (void)hipFree(sh->arc_labels_orig);
    (void)hipFree(sh->full_indptr); (void)hipFree(sh->full_src); (void)hipFree(sh->full_w);
    delete sh;
}

static int graph_create_impl(int64_t n_index, int64_t n_global, int64_t row_begin, int64_t own_off, int64_t n_rows, int64_t n_arcs,
                             const int32_t *indptr, const int32_t *adj_src, const float *adj_w, const float *arc_w,
                             const float *arc_labels, int dim_arc_label, const float *nodes, int dim_node_label,
                             const uint8_t *mask, int device, gnn_graph **out)
{
    ARGCHK(out, "out is NULL");
    *out = nullptr;
    ARGCHK(n_index > 0 && n_index < (int64_t)1 << 31 && n_global > 0 && n_global < (int64_t)1 << 31, "n_nodes=%lld out of range", (long long)n_global);
    ARGCHK(row_begin >= 0 && n_rows >= 0 && row_begin + n_rows <= n_global && own_off >= 0 && own_off + n_rows <= n_index,
           "owned rows [%lld,+%lld) outside [0,%lld)", (long long)row_begin, (long long)n_rows, (long long)n_global);
    ARGCHK(n_arcs >= 0 && n_arcs < (int64_t)1 << 31, "n_arcs=%lld out of range", (long long)n_arcs);
    ARGCHK(dim_node_label > 0 && dim_arc_label >= 0, "label dims must be NL>0, AL>=0");
    ARGCHK(indptr && nodes && mask, "indptr/nodes/mask are required");
    ARGCHK(n_arcs == 0 || (adj_src && adj_w && arc_w && (arc_labels || dim_arc_label == 0)), "arc arrays are required");
    ARGCHK(indptr[0] == 0 && indptr[n_rows] == n_arcs, "indptr[0]=%d, indptr[n_rows]=%d, n_arcs=%lld", indptr[0],
           indptr[n_rows], (long long)n_arcs);
    int maxdeg = 0;
    for (int64_t r = 0; r < n_rows; ++r) {
        const int d = indptr[r + 1] - indptr[r];
        ARGCHK(d >= 0, "indptr not monotone at row %lld", (long long)r);
        maxdeg = std::max(maxdeg, d);
    }
    for (int64_t e = 0; e < n_arcs; ++e)
        ARGCHK(adj_src[e] >= 0 && adj_src[e] < n_index, "adj_src[%lld]=%d outside [0,%lld)", (long long)e, adj_src[e], (long long)n_index);

    HIPCHK(hipSetDevice(device));
    gnn_graph *g = new gnn_graph();
    g->device = device; g->N = n_index; g->N_global = n_global; g->row_begin = row_begin; g->own_off = own_off; g->n_rows = n_rows; g->E = n_arcs;
    g->NL = dim_node_label; g->AL = dim_arc_label; g->base_NL = dim_node_label; g->base_AL = dim_arc_label;
    g->nodes_rows = n_index;
    g->sh = new gnn_graph_shared();
    g->sh->max_degree = maxdeg;
    // masked_rows holds [n_masked] owned-row indices with mask set, followed by [n_rows] exclusive positions
    std::vector<int32_t> rows, pos((size_t)n_rows);
    for (int64_t r = 0; r < n_rows; ++r) { pos[r] = (int32_t)rows.size(); if (mask[r]) rows.push_back((int32_t)r); }
    g->n_masked = (int64_t)rows.size();
    std::vector<int32_t> both(rows);
    both.insert(both.end(), pos.begin(), pos.end());
    int rc = 0;
    if ((rc = dev_upload(&g->sh->indptr, indptr, (size_t)n_rows + 1)) || (rc = dev_upload(&g->sh->adj_src, adj_src, (size_t)n_arcs)) ||
        (rc = dev_upload(&g->sh->adj_w, adj_w, (size_t)n_arcs)) || (rc = dev_upload(&g->sh->arc_w, arc_w, (size_t)n_arcs)) ||
        (rc = dev_upload(&g->sh->arc_labels, arc_labels, (size_t)n_arcs * dim_arc_label)) ||
        (rc = dev_upload(&g->sh->mask, mask, (size_t)n_rows)) || (rc = dev_upload(&g->sh->masked_rows, both.data(), both.size())) ||
        (rc = dev_upload(&g->nodes, nodes, (size_t)n_index * dim_node_label))) {
        gnn_graph_destroy(g);
        return rc;
    }
    *out = g;
    return GNN_OK;
}

extern "C" int gnn_graph_create(int64_t n_nodes, int64_t row_begin, int64_t n_rows, int64_t n_arcs,
                                const int32_t *indptr, const int32_t *adj_src, const float *adj_w, const float *arc_w,
                                const float *arc_labels, int dim_arc_label, const float *nodes, int dim_node_label,
                                const uint8_t *mask, int device, gnn_graph **out)
{
    return graph_create_impl(n_nodes, n_nodes, row_begin, row_begin, n_rows, n_arcs, indptr, adj_src, adj_w, arc_w, arc_labels, dim_arc_label,
                             nodes, dim_node_label, mask, device, out);
}

// Shard with a BOUNDARY exchange ("halo"): the state replica of rank r holds its own shard followed by one block per rank
// with only the rows that some OTHER rank reads (gnn_halo_plan computes the blocks from the whole graph), so the
// per-iteration all-gather moves boundary rows instead of whole shards.  Index space of adj_src / nodes:
//   [0, shard)                           owned rows (shard = rows per rank of gnn_shard_range, the last shard may be short)
//   shard + q * block + j                j-th boundary row of rank q (ascending global id), j < count_q <= block
extern "C" int gnn_graph_create_halo(int64_t n_nodes_global, int rank, int world, int64_t halo_block, int64_t n_send, const int32_t *send_rows,
                                     int64_t n_arcs, const int32_t *indptr, const int32_t *adj_src_replica, const float *adj_w,
                                     const float *arc_w, const float *arc_labels, int dim_arc_label, const float *nodes_replica,
                                     int dim_node_label, const uint8_t *mask, int device, gnn_graph **out)
{
    ARGCHK(out, "out is NULL");
    *out = nullptr;
    ARGCHK(world >= 2 && rank >= 0 && rank < world && halo_block >= 0 && n_send >= 0 && n_send <= halo_block && (n_send == 0 || send_rows), "bad halo description");
    int64_t rb = 0, nr = 0;
    int rc = gnn_shard_range(n_nodes_global, rank, world, &rb, &nr);
    if (rc) return rc;
    const int64_t shard = ((n_nodes_global + world - 1) / world + 31) / 32 * 32;
    for (int64_t j = 0; j < n_send; ++j)
        ARGCHK(send_rows[j] >= 0 && send_rows[j] < nr && (j == 0 || send_rows[j] > send_rows[j - 1]), "send_rows must be ascending owned-row indices");
    const int64_t n_index = shard + (int64_t)world * halo_block;
    rc = graph_create_impl(n_index, n_nodes_global, rb, 0, nr, n_arcs, indptr, adj_src_replica, adj_w, arc_w, arc_labels, dim_arc_label, nodes_replica,
                           dim_node_label, mask, device, out);
    if (rc) return rc;
    gnn_graph *g = *out;
    g->halo_world = world; g->halo_rank = rank; g->halo_block = halo_block; g->halo_count = n_send;
    rc = dev_upload(&g->halo_send, send_rows, (size_t)n_send);
    if (rc) { gnn_graph_destroy(g); *out = nullptr; return rc; }
    return GNN_OK;
}

// Host helper for gnn_graph_create_halo: from the CSR-by-destination of the WHOLE graph, the boundary rows of every rank.
// is_boundary[v] = 1 iff some arc v -> d has owner(d) != owner(v); counts[q] = boundary rows owned by rank q;
// slot[v] = position of v among the boundary rows of its owner (ascending id), -1 otherwise.  *block = max_q counts[q].
extern "C" int gnn_halo_plan(int64_t n_nodes, int world, const int32_t *indptr, const int32_t *adj_src, int32_t *slot, int64_t *counts, int64_t *block)
{
    ARGCHK(n_nodes > 0 && world >= 1 && indptr && slot && counts && block, "bad arguments");
    const int64_t shard = ((n_nodes + world - 1) / world + 31) / 32 * 32;
    std::vector<uint8_t> bnd((size_t)n_nodes, 0);
    for (int64_t d = 0; d < n_nodes; ++d) {
        const int64_t od = d / shard;
        for (int32_t e = indptr[d]; e < indptr[d + 1]; ++e) {
            const int32_t v = adj_src[e];
            ARGCHK(v >= 0 && v < n_nodes, "adj_src[%d]=%d outside [0,%lld)", e, v, (long long)n_nodes);
            if (v / shard != od) bnd[v] = 1;
        }
    }
    int64_t mx = 0;
    for (int q = 0; q < world; ++q) {
        int64_t c = 0;
        const int64_t b = std::min<int64_t>(n_nodes, shard * q), e = std::min<int64_t>(n_nodes, shard * (q + 1));
        for (int64_t v = b; v < e; ++v) slot[v] = bnd[v] ? (int32_t)c++ : -1;
        counts[q] = c;
        mx = std::max(mx, c);
    }
    *block = mx;
    return GNN_OK;
}

static inline const int32_t *graph_mask_pos(const gnn_graph *g) { return g->sh->masked_rows + g->n_masked; }

// rows allocated for the node labels of a derived graph: the sharded relabelling all-gathers whole shards in place, and
// shard * world <= N + 33 * world
static inline int64_t derived_node_rows(int64_t n) { return n + 33 * 64; }

extern "C" int gnn_graph_derive(const gnn_graph *base, int extra, gnn_graph **out)
{
    ARGCHK(base && out && extra >= 0, "bad arguments");
    *out = nullptr;
    HIPCHK(hipSetDevice(base->device));
    gnn_graph *g = new gnn_graph(*base);
    g->sh->refs++;
    g->NL = base->base_NL + extra;
    g->nodes = nullptr;
    g->arc_labels_own = g->arc_labels_orig_own = nullptr;      // never share the owned arc labels of a derived base
    g->halo_send_owned = false;                                // (boundary-exchange shards: same shard, same boundary rows; the base outlives its derived graphs' use of them)
    g->AL = base->base_AL;
    g->nodes_rows = derived_node_rows(g->N);
    g->ready = nullptr;                                        // (the base's event, if any, stays the base's)
    int rc = dev_alloc(&g->nodes, (size_t)g->nodes_rows * g->NL);
    if (!rc) rc = graph_zero_fill(g, g->nodes, (size_t)g->nodes_rows * g->NL * sizeof(float));
    if (rc) { gnn_graph_destroy(g); return rc; }
    *out = g;
    return GNN_OK;
}

extern "C" int gnn_graph_set_arc_order(gnn_graph *g, const int32_t *arc_id, const float *arc_labels_orig)
{
    ARGCHK(g && (g->E == 0 || (arc_id && (arc_labels_orig || g->AL == 0))), "bad arguments");
    ARGCHK(g->NL == g->base_NL && !g->arc_labels_own, "set the arc order on the original (underived) graph");
    HIPCHK(hipSetDevice(g->device));
    std::vector<uint8_t> seen((size_t)g->E, 0);
    for (int64_t q = 0; q < g->E; ++q) {
        ARGCHK(arc_id[q] >= 0 && arc_id[q] < g->E && !seen[arc_id[q]], "arc_id is not a permutation of the arcs at entry %lld", (long long)q);
        seen[arc_id[q]] = 1;
    }
    (void)hipFree(g->sh->arc_id); (void)hipFree(g->sh->arc_labels_orig);
    g->sh->arc_id = nullptr; g->sh->arc_labels_orig = nullptr;
    int rc = dev_upload(&g->sh->arc_id, arc_id, (size_t)g->E);
    if (!rc) rc = dev_upload(&g->sh->arc_labels_orig, arc_labels_orig, (size_t)g->E * g->AL);
    return rc;
}

extern "C" int gnn_graph_derive_edge(const gnn_graph *base, int extra_nodes, int extra_arcs, gnn_graph **out)
{
    ARGCHK(base && out && extra_nodes >= 0 && extra_arcs >= 0, "bad arguments");
    ARGCHK(base->sh->arc_id, "call gnn_graph_set_arc_order on the base graph first");
    ARGCHK(base->n_rows == base->N, "edge-based LGNN stacks are single-GPU only");
    int rc = gnn_graph_derive(base, extra_nodes, out);
    if (rc) return rc;
    gnn_graph *g = *out;
    *out = nullptr;
    g->AL = base->base_AL + extra_arcs;
    g->arc_labels_own = g->arc_labels_orig_own = nullptr;
    rc = dev_alloc(&g->arc_labels_own, (size_t)g->E * g->AL);
    if (!rc) rc = dev_alloc(&g->arc_labels_orig_own, (size_t)g->E * g->AL);
    if (!rc) rc = graph_zero_fill(g, g->arc_labels_own, std::max<size_t>(1, (size_t)g->E * g->AL) * sizeof(float));
    if (!rc) rc = graph_zero_fill(g, g->arc_labels_orig_own, std::max<size_t>(1, (size_t)g->E * g->AL) * sizeof(float));
    if (rc) { gnn_graph_destroy(g); return rc; }
    *out = g;
    return GNN_OK;
}

extern "C" int gnn_graph_get_nodes(const gnn_graph *g, float *nodes_out)
{
    ARGCHK(g && nodes_out, "bad arguments");
    HIPCHK(hipSetDevice(g->device));
    if (g->ready) HIPCHK(hipEventSynchronize(g->ready));
    HIPCHK(hipMemcpy(nodes_out, g->nodes, (size_t)g->N * g->NL * sizeof(float), hipMemcpyDeviceToHost));   // index-space rows (all nodes for full replicas)
    return GNN_OK;
}

extern "C" int gnn_graph_dims(const gnn_graph *g, int64_t *n_nodes, int64_t *n_rows, int64_t *n_arcs, int *nl, int *al,
                              int64_t *n_masked)
{
    ARGCHK(g, "graph is NULL");
    if (n_nodes) *n_nodes = g->N;
    if (n_rows) *n_rows = g->n_rows;
    if (n_arcs) *n_arcs = g->E;
    if (nl) *nl = g->NL;
    if (al) *al = g->AL;
    if (n_masked) *n_masked = g->n_masked;
    return GNN_OK;
}

extern "C" int gnn_graph_set_full_adjacency(gnn_graph *g, int64_t n_global, const int32_t *indptr, const int32_t *adj_src, const float *adj_w)
{
    ARGCHK(g && indptr && n_global > 0, "bad arguments");
    ARGCHK(n_global == g->N_global, "the shard belongs to a graph of %lld nodes, not %lld", (long long)g->N_global, (long long)n_global);
    ARGCHK(!g->halo_world, "a boundary-exchange shard numbers its sources in its own compact space: use a full-replica shard");
    const int64_t e = indptr[n_global];
    ARGCHK(indptr[0] == 0 && e >= 0 && (e == 0 || (adj_src && adj_w)), "bad CSR");
    for (int64_t i = 0; i < n_global; ++i) ARGCHK(indptr[i] <= indptr[i + 1], "indptr must be non-decreasing");
    for (int64_t q = 0; q < e; ++q) ARGCHK(adj_src[q] >= 0 && adj_src[q] < n_global, "adj_src[%lld] out of range", (long long)q);
    HIPCHK(hipSetDevice(g->device));
    gnn_graph_shared *sh = g->sh;                  // shared with the graphs derived from g
    (void)hipFree(sh->full_indptr); (void)hipFree(sh->full_src); (void)hipFree(sh->full_w);
    sh->full_indptr = nullptr; sh->full_src = nullptr; sh->full_w = nullptr; sh->full_rows = 0;
    int rc = dev_upload(&sh->full_indptr, indptr, (size_t)n_global + 1);
    if (!rc) rc = dev_upload(&sh->full_src, adj_src, (size_t)e);
    if (!rc) rc = dev_upload(&sh->full_w, adj_w, (size_t)e);
    if (rc) return rc;
    sh->full_rows = n_global;
    return GNN_OK;
}

extern "C" int gnn_graph_destroy(gnn_graph *g)
{
    if (!g) return GNN_OK;
    (void)hipSetDevice(g->device);
    (void)hipFree(g->arc_labels_own); (void)hipFree(g->arc_labels_orig_own);
    if (g->halo_send_owned) (void)hipFree(g->halo_send);
    (void)hipFree(g->nodes);
    if (g->ready) (void)hipEventDestroy(g->ready);
    graph_release_shared(g->sh);
    delete g;
    return GNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// MLP
// ---------------------------------------------------------------------------------------------------------------------
__global__ void k_bn_inference_form(int f, float eps, const float *raw, float *scale, float *shift)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= f) return;
    // same expression order as oracle/gnn_oracle.c:orc_batchnorm.  sqrtf and operator/ are correctly rounded on the device;
    // __fsqrt_rn is NOT (tools/sqrt_check.hip: 16 % of 4 M inputs differ from the host's sqrtf by one ulp), so it is not used anywhere.
    const float g = raw[j], be = raw[f + j], mu = raw[2 * f + j], var = raw[3 * f + j];
    const float sc = (1.0f / sqrtf(var + eps)) * g;
    const float t = mu * sc;
    scale[j] = sc;
    shift[j] = be - t;
}

// scale / shift of the inference form from the raw BatchNormalization arrays on the device (after an upload or an optimizer step)
int gnn_mlp_refresh_bn(gnn_mlp *m, hipStream_t st)
{
    if (!m->has_bn) return GNN_OK;
    const int f = m->dims.back();
    hipLaunchKernelGGL(k_bn_inference_form, cdiv(f, 64), 64, 0, st, f, m->eps, m->bn_raw, m->bn_scale, m->bn_shift);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

static int mlp_upload(gnn_mlp *m, const float *const *W, const float *const *b, const float *bn)
{
    // one staging copy of the whole slab: a single transfer instead of two per layer
    std::vector<float> stage(m->slab_floats, 0.0f);
    for (int l = 0; l < m->n_layers; ++l) {
        ARGCHK(W[l] && b[l], "W[%d]/b[%d] is NULL", l, l);
        memcpy(stage.data() + (m->W[l] - m->slab), W[l], sizeof(float) * (size_t)m->dims[l] * m->dims[l + 1]);
        memcpy(stage.data() + (m->b[l] - m->slab), b[l], sizeof(float) * (size_t)m->dims[l + 1]);
    }
    HIPCHK(hipMemcpy(m->slab, stage.data(), sizeof(float) * m->slab_floats, hipMemcpyHostToDevice));
    if (m->has_bn) {
        ARGCHK(bn, "this MLP ends with BatchNormalization: bn is required");
        const int f = m->dims.back();
        HIPCHK(hipMemcpy(m->bn_raw, bn, sizeof(float) * 4 * f, hipMemcpyHostToDevice));
        int rc = gnn_mlp_refresh_bn(m, nullptr);
        if (rc) return rc;
        HIPCHK(hipStreamSynchronize(nullptr));
    }
    m->version++;
    m->pack_dirty = true;       // packed images of the fused kernel: rebuilt lazily (gnn_fused_supported)
    return GNN_OK;
}

extern "C" int gnn_mlp_create(int n_layers, const int32_t *dims, const int32_t *acts, const float *const *W,
                              const float *const *b, const float *bn, float bn_eps, int device, gnn_mlp **out)
{
    ARGCHK(out, "out is NULL");
    *out = nullptr;
    ARGCHK(n_layers >= 1 && n_layers <= 16 && dims && acts && W && b, "bad MLP description");
    for (int l = 0; l <= n_layers; ++l) ARGCHK(dims[l] > 0 && dims[l] <= 8192, "dims[%d]=%d", l, dims[l]);
    for (int l = 0; l < n_layers; ++l) {
        ARGCHK(acts[l] >= GNN_ACT_LINEAR && acts[l] <= GNN_ACT_SOFTMAX, "unknown activation code %d", acts[l]);
        ARGCHK(acts[l] != GNN_ACT_SOFTMAX || dims[l + 1] <= 1024, "softmax width %d too large", dims[l + 1]);
    }
    HIPCHK(hipSetDevice(device));
    gnn_mlp *m = new gnn_mlp();
    m->device = device; m->n_layers = n_layers; m->eps = bn_eps; m->has_bn = bn != nullptr;
    m->dims.assign(dims, dims + n_layers + 1);
    m->acts.assign(acts, acts + n_layers);
    m->W.assign(n_layers, nullptr);
    m->b.assign(n_layers, nullptr);
    int rc = 0;
    auto pad = [](size_t n) { return (n + 63) & ~(size_t)63; };
    size_t off = 0;
    for (int l = 0; l < n_layers; ++l) off += pad((size_t)dims[l] * dims[l + 1]) + pad((size_t)dims[l + 1]);
    m->slab_floats = off;
    rc = dev_alloc(&m->slab, off);
    off = 0;
    for (int l = 0; l < n_layers && !rc; ++l) {
        m->W[l] = m->slab + off; off += pad((size_t)dims[l] * dims[l + 1]);
        m->b[l] = m->slab + off; off += pad((size_t)dims[l + 1]);
    }
    if (!rc && m->has_bn) {
        rc = dev_alloc(&m->bn_scale, (size_t)dims[n_layers]);
        if (!rc) rc = dev_alloc(&m->bn_shift, (size_t)dims[n_layers]);
        if (!rc) rc = dev_alloc(&m->bn_raw, (size_t)4 * dims[n_layers]);
    }
    if (!rc) rc = mlp_upload(m, W, b, bn);
    if (rc) { gnn_mlp_destroy(m); return rc; }
    *out = m;
    return GNN_OK;
}

extern "C" int gnn_mlp_set_weights(gnn_mlp *m, const float *const *W, const float *const *b, const float *bn)
{
    ARGCHK(m && W && b, "bad arguments");
    HIPCHK(hipSetDevice(m->device));
    return mlp_upload(m, W, b, bn);
}

extern "C" int gnn_mlp_get_weights(gnn_mlp *m, float *const *W, float *const *b, float *bn)
{
    ARGCHK(m && W && b, "bad arguments");
    ARGCHK(!m->has_bn || bn, "this MLP ends with BatchNormalization: bn [4 * width] is required");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipDeviceSynchronize());                 // an optimizer step may still be running on a loop's stream
    std::vector<float> stage(m->slab_floats);
    HIPCHK(hipMemcpy(stage.data(), m->slab, sizeof(float) * m->slab_floats, hipMemcpyDeviceToHost));
    for (int l = 0; l < m->n_layers; ++l) {
        ARGCHK(W[l] && b[l], "W[%d]/b[%d] is NULL", l, l);
        memcpy(W[l], stage.data() + (m->W[l] - m->slab), sizeof(float) * (size_t)m->dims[l] * m->dims[l + 1]);
        memcpy(b[l], stage.data() + (m->b[l] - m->slab), sizeof(float) * (size_t)m->dims[l + 1]);
    }
    if (m->has_bn) HIPCHK(hipMemcpy(bn, m->bn_raw, sizeof(float) * 4 * m->dims.back(), hipMemcpyDeviceToHost));
    return GNN_OK;
}

extern "C" int gnn_mlp_reset_optimizer(gnn_mlp *m)
{
    ARGCHK(m, "mlp is NULL");
    HIPCHK(hipSetDevice(m->device));
    HIPCHK(hipDeviceSynchronize());
    (void)hipFree(m->opt_a); (void)hipFree(m->opt_b);
    m->opt_a = nullptr; m->opt_b = nullptr;      // allocated and zeroed again by the next device-side optimizer step
    return GNN_OK;
}

extern "C" int gnn_mlp_forward(gnn_mlp *m, int64_t n_rows, const float *x, float *y)
{
    ARGCHK(m && n_rows >= 0 && (n_rows == 0 || (x && y)), "bad arguments");
    if (n_rows == 0) return GNN_OK;
    HIPCHK(hipSetDevice(m->device));
    int maxw = 1;
    for (int l = 1; l <= m->n_layers; ++l) maxw = std::max(maxw, m->dims[l]);
    float *dx = nullptr, *dy = nullptr, *t0 = nullptr, *t1 = nullptr;
    int rc = dev_upload(&dx, x, (size_t)n_rows * m->dims[0]);
    if (!rc) rc = dev_alloc(&dy, (size_t)n_rows * m->dims.back());
    if (!rc) rc = dev_alloc(&t0, (size_t)n_rows * maxw);
    if (!rc) rc = dev_alloc(&t1, (size_t)n_rows * maxw);
    if (!rc) rc = launch_mlp(nullptr, m, n_rows, dx, m->dims[0], dy, m->dims.back(), t0, t1, nullptr, 1);
    if (!rc) {
        hipError_t e = hipMemcpy(y, dy, sizeof(float) * (size_t)n_rows * m->dims.back(), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = gnn_fail(GNN_ERR_HIP, "copy back: %s", hipGetErrorString(e));
    }
    (void)hipFree(dx); (void)hipFree(dy); (void)hipFree(t0); (void)hipFree(t1);
    return rc;
}

extern "C" int gnn_mlp_destroy(gnn_mlp *m)
{
    if (!m) return GNN_OK;
    (void)hipSetDevice(m->device);
    (void)hipFree(m->slab);
    (void)hipFree(m->bn_scale); (void)hipFree(m->bn_shift); (void)hipFree(m->bn_raw);
    (void)hipFree(m->opt_a); (void)hipFree(m->opt_b);
    gnn_fused_release(m);
    delete m;
    return GNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// RCCL (loaded lazily so that the library itself has no link-time dependency on it)
// ---------------------------------------------------------------------------------------------------------------------
struct Id128 { char b[128]; };   // ncclUniqueId, passed by value
namespace {
struct Rccl {
    void *h = nullptr;
    int (*GetUniqueId)(void *) = nullptr;
    int (*CommInitRank)(void **, int, Id128, int) = nullptr;
    int (*CommDestroy)(void *) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, void *, hipStream_t) = nullptr;
    int (*AllReduce)(const void *, void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Send)(const void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*Recv)(void *, size_t, int, int, void *, hipStream_t) = nullptr;
    int (*GroupStart)() = nullptr;
    int (*GroupEnd)() = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
};
}   // namespace
static Rccl g_rccl;
enum { NCCL_INT32 = 2, NCCL_FLOAT32 = 7, NCCL_FLOAT64 = 8, NCCL_MAX = 2 };

static int rccl_load()
{
    if (g_rccl.h) return GNN_OK;
    void *h = nullptr;
    // GNN_RCCL_LIBRARY=<path>: the collectives library to load instead of the system's RCCL (another RCCL build; the multi-process tests
    // point it at their stand-in transport, tests/mock_rccl, to run several ranks on one GPU).  No fallback when it is set.
    if (const char *path = getenv("GNN_RCCL_LIBRARY")) {
        h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
        if (!h) return gnn_fail(GNN_ERR_COMM, "cannot load GNN_RCCL_LIBRARY=%s: %s", path, dlerror());
    }
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) return gnn_fail(GNN_ERR_COMM, "cannot load librccl: %s", dlerror());
#define SYM(field, name)                                                         \
    *(void **)(&g_rccl.field) = dlsym(h, name);                                  \
    if (!g_rccl.field) return gnn_fail(GNN_ERR_COMM, "librccl lacks %s", name);
    SYM(GetUniqueId, "ncclGetUniqueId") SYM(CommInitRank, "ncclCommInitRank") SYM(CommDestroy, "ncclCommDestroy")
    SYM(AllGather, "ncclAllGather") SYM(AllReduce, "ncclAllReduce") SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd") SYM(GetErrorString, "ncclGetErrorString") SYM(Send, "ncclSend") SYM(Recv, "ncclRecv")
#undef SYM
    g_rccl.h = h;
    return GNN_OK;
}

#define NCCLCHK(expr)                                                                                   \
    do {                                                                                                \
        int r_ = (expr);                                                                                \
        if (r_ != 0) return gnn_fail(GNN_ERR_COMM, "%s -> %s", #expr, g_rccl.GetErrorString(r_));      \
    } while (0)

extern "C" int gnn_comm_unique_id(uint8_t id[128])
{
    ARGCHK(id, "id is NULL");
    int rc = rccl_load();
    if (rc) return rc;
    NCCLCHK(g_rccl.GetUniqueId(id));
    return GNN_OK;
}

extern "C" int gnn_comm_create(const uint8_t id[128], int rank, int world, int device, gnn_comm **out)
{
    ARGCHK(id && out && world >= 1 && rank >= 0 && rank < world, "bad arguments");
    *out = nullptr;
    int rc = rccl_load();
    if (rc) return rc;
    HIPCHK(hipSetDevice(device));
    gnn_comm *c = new gnn_comm();
    c->rank = rank; c->world = world; c->device = device;
    Id128 uid;
    memcpy(uid.b, id, 128);
    int r = g_rccl.CommInitRank(&c->nccl, world, uid, rank);
    if (r != 0) { delete c; return gnn_fail(GNN_ERR_COMM, "ncclCommInitRank -> %s", g_rccl.GetErrorString(r)); }
    if (hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking) != hipSuccess || gnn_dev_malloc((void **)&c->scratch, sizeof(double)) != hipSuccess) {
        gnn_comm_destroy(c);
        return gnn_fail(GNN_ERR_HIP, "communicator stream / scratch allocation failed");
    }
    *out = c;
    return GNN_OK;
}

// `world` communicators on ONE device sharing one stream (see gnn_comm_group): the sharded engine path on a single GPU.
extern "C" int gnn_comm_create_loopback(int world, int device, gnn_comm **out /* [world] */)
{
    ARGCHK(out && world >= 1 && world <= 64, "bad arguments");
    for (int r = 0; r < world; ++r) out[r] = nullptr;
    HIPCHK(hipSetDevice(device));
    gnn_comm_group *grp = new gnn_comm_group();
    grp->world = world;
    grp->member.assign((size_t)world, nullptr);
    hipError_t e = hipStreamCreateWithFlags(&grp->stream, hipStreamNonBlocking);
    if (e != hipSuccess) { delete grp; return gnn_fail(GNN_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    for (int r = 0; r < world; ++r) {
        gnn_comm *c = new gnn_comm();
        c->rank = r; c->world = world; c->device = device; c->grp = grp; c->stream = grp->stream;
        grp->refs++;
        out[r] = c;
    }
    return GNN_OK;
}

extern "C" int gnn_comm_allreduce_max(gnn_comm *c, double *value)
{
    ARGCHK(c && value, "bad arguments");
    HIPCHK(hipSetDevice(c->device));
    if (c->grp) { HIPCHK(hipStreamSynchronize(c->stream)); return GNN_OK; }   // one process: the value is already the maximum
    HIPCHK(hipMemcpyAsync(c->scratch, value, sizeof(double), hipMemcpyHostToDevice, c->stream));
    NCCLCHK(g_rccl.AllReduce(c->scratch, c->scratch, 1, NCCL_FLOAT64, NCCL_MAX, c->nccl, c->stream));
    HIPCHK(hipMemcpyAsync(value, c->scratch, sizeof(double), hipMemcpyDeviceToHost, c->stream));
    HIPCHK(hipStreamSynchronize(c->stream));
    return GNN_OK;
}

int gnn_comm_allgather32(gnn_comm *c, const void *send, void *recv, size_t count, hipStream_t st)
{
    if (!c || c->grp || !c->nccl) return gnn_fail(GNN_ERR_UNSUPPORTED, "this exchange needs an RCCL communicator (one process per rank), not a loopback group");
    NCCLCHK(g_rccl.AllGather(send, recv, count, NCCL_INT32, c->nccl, st));
    return GNN_OK;
}

extern "C" int gnn_comm_destroy(gnn_comm *c)
{
    if (!c) return GNN_OK;
    if (c->loops > 0) { c->closed = true; return GNN_OK; }      // released by the last gnn_loop_destroy
    (void)hipSetDevice(c->device);
    if (c->grp) {
        if (--c->grp->refs == 0) { (void)hipStreamDestroy(c->grp->stream); delete c->grp; }
        if (c->xstream) (void)hipStreamDestroy(c->xstream);
        delete c;
        return GNN_OK;
    }
    if (c->nccl && g_rccl.CommDestroy) g_rccl.CommDestroy(c->nccl);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    if (c->xstream) (void)hipStreamDestroy(c->xstream);
    (void)hipFree(c->scratch);
    delete c;
    return GNN_OK;
}

extern "C" int gnn_shard_range(int64_t n_nodes, int rank, int world, int64_t *row_begin, int64_t *n_rows)
{
    ARGCHK(n_nodes > 0 && world >= 1 && rank >= 0 && rank < world && row_begin && n_rows, "bad arguments");
    const int64_t shard = ((n_nodes + world - 1) / world + 31) / 32 * 32;   // whole 32-node tiles per rank
    const int64_t b = std::min<int64_t>(n_nodes, shard * rank), e = std::min<int64_t>(n_nodes, shard * (rank + 1));
    *row_begin = b;
    *n_rows = e - b;
    return GNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// loop
// ---------------------------------------------------------------------------------------------------------------------
extern "C" int gnn_loop_create(gnn_graph *g, gnn_mlp *net_state, gnn_mlp *net_output, int state_dim, int max_iter,
                               float threshold, gnn_comm *comm, gnn_loop **out)
{
    ARGCHK(out, "out is NULL");
    *out = nullptr;
    ARGCHK(g && net_state && net_output, "graph and both MLPs are required");
    ARGCHK(state_dim >= 0, "param <state_vect_dim> must be int>=0");           // GNN/GNN.py:53
    ARGCHK(max_iter >= 0 && max_iter <= 100000, "max_iteration=%d out of range", max_iter);
    ARGCHK(g->device == net_state->device && g->device == net_output->device, "handles live on different devices");
    const int Ds = state_dim ? state_dim : g->NL;
    const int NLc = state_dim ? g->NL : 0;
    const int in_s = Ds + NLc + Ds + NLc + g->AL;                              // GNN/MLP.py:104
    ARGCHK(net_state->dims[0] == in_s, "net_state input width %d != AL + 2*(NL + D) = %d", net_state->dims[0], in_s);
    ARGCHK(net_state->dims.back() == Ds, "net_state output width %d != state width %d", net_state->dims.back(), Ds);
    const bool edge_width = net_output->dims[0] == 2 * (Ds + NLc) + g->AL && net_output->dims[0] != Ds + NLc;
    ARGCHK(net_output->dims[0] == Ds + NLc || edge_width, "net_output input width %d is neither NL + D = %d (node/graph based) nor 2 (NL + D) + AL = %d (edge based)",
           net_output->dims[0], Ds + NLc, 2 * (Ds + NLc) + g->AL);
    const int world = comm ? comm->world : 1, rank = comm ? comm->rank : 0;
    ARGCHK(!g->halo_world || (g->halo_world == world && g->halo_rank == rank), "boundary-exchange shard of rank %d/%d used with rank %d/%d",
           g->halo_rank, g->halo_world, rank, world);
    ARGCHK(!comm || !comm->grp || !comm->grp->member[rank], "loopback rank %d already has a loop (one loop per rank and group)", rank);
    int64_t rb = 0, nr = 0;
    gnn_shard_range(g->N_global, rank, world, &rb, &nr);
    ARGCHK(rb == g->row_begin && nr == g->n_rows, "graph owns rows [%lld,+%lld) but rank %d/%d must own [%lld,+%lld)",
           (long long)g->row_begin, (long long)g->n_rows, rank, world, (long long)rb, (long long)nr);
    ARGCHK(!comm || comm->device == g->device, "communicator and graph live on different devices");

    HIPCHK(hipSetDevice(g->device));
    gnn_loop *l = new gnn_loop();
    if (comm) comm->loops++;                 // (gnn_loop_destroy, also on the failure paths below, gives it back)
    l->g = g; l->st = net_state; l->ou = net_output; l->comm = comm; l->device = g->device; l->rank = rank; l->world = world;
    l->D = state_dim; l->Ds = Ds; l->NLc = NLc; l->in_s = in_s; l->wf = Ds + NLc; l->T = net_output->dims.back();
    l->max_iter = max_iter; l->thr = threshold;
    l->edge_expected = edge_width;
    l->shard_rows = ((g->N_global + world - 1) / world + 31) / 32 * 32;
    l->N_pad = g->halo_world ? g->N : l->shard_rows * world;                  // rows of the state replica
    l->own_off = g->halo_world ? 0 : l->shard_rows * rank;
    int rc = 0;
    if (comm) l->stream = comm->stream;
    else {
        hipError_t e = hipStreamCreateWithFlags(&l->stream, hipStreamNonBlocking);
        if (e != hipSuccess) { gnn_loop_destroy(l); return gnn_fail(GNN_ERR_HIP, "hipStreamCreate: %s", hipGetErrorString(e)); }
    }
    int maxw_s = 1, maxw_o = 1;
    for (int i = 1; i <= net_state->n_layers; ++i) maxw_s = std::max(maxw_s, net_state->dims[i]);
    for (int i = 1; i <= net_output->n_layers; ++i) maxw_o = std::max(maxw_o, net_output->dims[i]);
    for (int b = 0; b < 2 && !rc; ++b) {
        rc = dev_alloc(&l->state[b], (size_t)l->N_pad * Ds);
        if (!rc) rc = zero_on_stream(l->state[b], sizeof(float) * (size_t)l->N_pad * Ds, l->stream);      // ordered before everything this loop ever queues
    }
    if (!rc) rc = dev_alloc(&l->flags, (size_t)(max_iter + 2) * world * GNN_FLAG_WORDS + 4);   // + barrier counter / status of the persistent loop
    if (!rc) rc = dev_alloc(&l->kfinal_dev, 4);        // k, status word of the persistent loop, "gate not certified", pad
    if (!rc) rc = dev_alloc(&l->tile_ctr, (2 * ((size_t)max_iter + 1) + 3) & ~(size_t)3);      // one ticket counter per body (the second half is spare: a partial last tile used to get a launch of its own)
    if (!rc && hipHostMalloc((void **)&l->kfinal_host, 4 * sizeof(int)) != hipSuccess) rc = gnn_fail(GNN_ERR_HIP, "hipHostMalloc");
    if (!rc) l->kfinal_host[1] = l->kfinal_host[2] = 0;
    if (!rc && hipHostMalloc((void **)&l->gate_host, sizeof(int) * (size_t)world * GNN_FLAG_WORDS) != hipSuccess) rc = gnn_fail(GNN_ERR_HIP, "hipHostMalloc");
    if (!edge_width) {      // the edge-based buffers are sized in gnn_loop_set_edge_readout
        if (!rc) rc = dev_alloc(&l->feats, (size_t)g->n_masked * l->wf);
        if (!rc) rc = dev_alloc(&l->out, (size_t)g->n_masked * l->T);
        for (int b = 0; b < 2 && !rc; ++b) rc = dev_alloc(&l->otmp[b], (size_t)g->n_masked * maxw_o);
    }
    if (!rc && hipEventCreate(&l->ev_total[0]) != hipSuccess) rc = gnn_fail(GNN_ERR_HIP, "hipEventCreate");
    if (!rc && hipEventCreate(&l->ev_total[1]) != hipSuccess) rc = gnn_fail(GNN_ERR_HIP, "hipEventCreate");
    if (rc) { gnn_loop_destroy(l); return rc; }
    l->impl_req = 2;            // fastest supported path by default; gnn_loop_set_impl(1) selects the bit-exact f32 MFMA
    (void)maxw_s;
    if (comm && comm->grp) comm->grp->member[rank] = l;
    *out = l;
    return GNN_OK;
}

static int loop_ensure_unfused(gnn_loop *l)
{
    if (l->inp) return GNN_OK;
    int maxw = 1;
    for (int i = 1; i <= l->st->n_layers; ++i) maxw = std::max(maxw, l->st->dims[i]);
    int rc = dev_alloc(&l->inp, (size_t)l->g->n_rows * l->in_s);
    for (int b = 0; b < 2 && !rc; ++b) rc = dev_alloc(&l->tmp[b], (size_t)l->g->n_rows * maxw);
    return rc;
}

extern "C" int gnn_loop_set_impl(gnn_loop *l, int impl, int *used)
{
    ARGCHK(l && impl >= 0 && impl <= 2, "impl must be 0 (unfused), 1 (fused, exact f32 MFMA) or 2 (fused, split bf16 MFMA)");
    l->impl_req = impl;
    if (used) *used = (impl >= 1 && gnn_fused_supported(l)) ? impl : 0;
    return GNN_OK;
}

extern "C" int gnn_loop_gate_info(const gnn_loop *l, int *last_run_repeated, int *repeats_total)
{
    ARGCHK(l, "loop is NULL");
    if (last_run_repeated) *last_run_repeated = l->last_run_rerun ? 1 : 0;
    if (repeats_total) *repeats_total = l->certified_reruns;
    return GNN_OK;
}

// forget the loop-invariant label aggregates (GNN.py:259, :263) kept from the previous run: the next run rebuilds them, as every
// Loop() call of the reference does
extern "C" int gnn_loop_drop_cached_aggregates(gnn_loop *l)
{
    ARGCHK(l, "loop is NULL");
    l->inv_version = 0;
    return GNN_OK;
}

// small graphs run all bodies of a Loop inside one persistent launch (gnn_small.hip); enable = 0 keeps to one launch per body
extern "C" int gnn_loop_set_persistent(gnn_loop *l, int enable, int *used)
{
    ARGCHK(l, "loop is NULL");
    l->small_disabled = enable == 0;
    if (used) *used = gnn_small_supported(l) ? 1 : 0;
    return GNN_OK;
}

extern "C" int gnn_loop_set_tile_form(gnn_loop *l, int form, int *used)
{
    ARGCHK(l && form >= 0 && form <= 2, "form must be 0 (library's choice), 1 (one wave per tile) or 2 (wave pair per tile)");
    l->tile_form = form;
    if (used) *used = (l->impl_req >= 1 && gnn_fused_supported(l)) ? (gnn_fused_pair_selected(l) ? 2 : 1) : 0;
    return GNN_OK;
}

extern "C" int gnn_loop_set_profiling(gnn_loop *l, int enable)
{
    ARGCHK(l, "loop is NULL");
    l->profiling = enable != 0;
    return GNN_OK;
}

extern "C" int gnn_loop_get_timing(const gnn_loop *l, float *total_ms, float *avg_iter_ms, int *n_iter_timed)
{
    ARGCHK(l, "loop is NULL");
    if (!l->ran) return gnn_fail(GNN_ERR_STATE, "gnn_loop_run has not been called");
    if (total_ms) *total_ms = l->total_ms;
    if (avg_iter_ms) *avg_iter_ms = l->avg_iter_ms;
    if (n_iter_timed) *n_iter_timed = l->n_iter_timed;
    return GNN_OK;
}

extern "C" int gnn_loop_get_exchange_timing(const gnn_loop *l, float *avg_between_bodies_ms)
{
    ARGCHK(l && avg_between_bodies_ms, "bad arguments");
    if (!l->ran) return gnn_fail(GNN_ERR_STATE, "gnn_loop_run has not been called");
    *avg_between_bodies_ms = l->avg_gap_ms;
    return GNN_OK;
}

extern "C" int gnn_counters_get(const gnn_loop *l, double *bytes_per_iteration, double *flops_per_iteration, int *iterations, float *total_ms,
                                float *avg_iteration_ms)
{
    ARGCHK(l, "loop is NULL");
    const gnn_graph *g = l->g;
    const double n = (double)g->n_rows, e = (double)g->E, ds = (double)l->Ds;
    if (bytes_per_iteration) *bytes_per_iteration = e * (4.0 * ds + 8.0) + 4.0 * (n + 1.0) + n * (8.0 * ds + 4.0 * (2.0 * l->NLc + g->AL));
    if (flops_per_iteration) {
        double f = 0.0;
        for (int i = 0; i < l->st->n_layers; ++i) f += 2.0 * (double)l->st->dims[i] * (double)l->st->dims[i + 1];
        *flops_per_iteration = n * f + 2.0 * e * ds;
    }
    if (iterations) *iterations = l->ran ? l->kfinal : 0;
    if (total_ms) *total_ms = l->ran ? l->total_ms : 0.f;
    if (avg_iteration_ms) *avg_iteration_ms = l->ran ? l->avg_iter_ms : 0.f;
    return GNN_OK;
}

extern "C" int gnn_loop_set_state0(gnn_loop *l, const float *state0, uint64_t seed)
{
    ARGCHK(l, "loop is NULL");
    HIPCHK(hipSetDevice(l->device));
    const gnn_graph *g = l->g;
    const size_t cnt = (size_t)g->n_rows * l->Ds;
    if (!l->state_init && l->D) { int rc = dev_alloc(&l->state_init, cnt); if (rc) return rc; }
    float *own = l->state_init;
    if (l->D == 0) {   // state <- node labels (GNN.py:265); taken from the graph at run time
        l->have_state0 = true;
        return GNN_OK;
    }
    if (state0) {
        HIPCHK(hipMemcpyAsync(own, state0, cnt * sizeof(float), hipMemcpyHostToDevice, l->stream));
        HIPCHK(hipStreamSynchronize(l->stream));
    } else if (cnt) {
        hipLaunchKernelGGL(k_randn, cdiv((int64_t)cnt, 256), 256, 0, l->stream, (int64_t)cnt, (int64_t)g->row_begin * l->Ds, seed, 0.1f, own);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(l->stream));
    }
    l->have_state0 = true;
    return GNN_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// exchange step of the sharded loop (reference: the reads of `state` at GNN/GNN.py:234 and the global reduce_any at :218
// span all nodes; here every rank owns a node range).  One call = "everybody gets the owned state rows of buffer `b` and the
// flag words at int offset `flag_off` of every rank".  b < 0 / flag_off == NO_FLAGS skip that part.
//   RCCL communicator      one grouped call: in-place all-gather of the owned rows (full replicas) or of the packed
//                          boundary rows (halo shards) + all-gather of the rank's flag block
//   loopback communicator  the same data movement as device-to-device copies into the other members' buffers, on the
//                          group's single stream (so ordering is program order)
// ---------------------------------------------------------------------------------------------------------------------
static const size_t NO_FLAGS = ~(size_t)0;

__global__ void k_pack_rows(int64_t count, int Ds, const int32_t *__restrict__ rows, const float *__restrict__ own, float *__restrict__ dst)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count * Ds) return;
    const int64_t r = t / Ds;
    const int c = (int)(t - r * Ds);
    dst[t] = own[(int64_t)rows[r] * Ds + c];
}

// replica segment this rank publishes: [seg_off, seg_off + seg_rows) rows of the state replica
static inline void loop_segment(const gnn_loop *l, int rank, size_t *off_rows, size_t *rows)
{
    const gnn_graph *g = l->g;
    if (g->halo_world) { *off_rows = (size_t)l->shard_rows + (size_t)rank * g->halo_block; *rows = (size_t)g->halo_block; }
    else { *off_rows = (size_t)l->shard_rows * rank; *rows = (size_t)l->shard_rows; }
}

static int loop_exchange(gnn_loop *l, int b, size_t flag_off)
{
    if (l->world == 1) return GNN_OK;
    const gnn_graph *g = l->g;
    size_t off = 0, rows = 0;
    loop_segment(l, l->rank, &off, &rows);
    if (b >= 0 && g->halo_world && g->halo_count) {      // boundary rows of the owned range -> this rank's block of the replica
        const int64_t tot = g->halo_count * l->Ds;
        hipLaunchKernelGGL(k_pack_rows, cdiv(tot, 256), 256, 0, l->stream, g->halo_count, l->Ds, g->halo_send, l->state[b] + (size_t)l->own_off * l->Ds,
                           l->state[b] + off * l->Ds);
        HIPCHK(hipGetLastError());
    }
    if (l->comm->grp) {
        gnn_comm_group *grp = l->comm->grp;
        for (int p = 0; p < l->world; ++p) {
            gnn_loop *peer = grp->member[p];
            if (p == l->rank) continue;
            if (!peer) return gnn_fail(GNN_ERR_STATE, "loopback rank %d has no loop: create one loop per rank and run them with gnn_loop_run_group", p);
            if (b >= 0 && rows)
                HIPCHK(hipMemcpyAsync(peer->state[b] + off * l->Ds, l->state[b] + off * l->Ds, sizeof(float) * rows * l->Ds, hipMemcpyDeviceToDevice, l->stream));
            if (flag_off != NO_FLAGS)
                HIPCHK(hipMemcpyAsync(peer->flags + flag_off + (size_t)l->rank * GNN_FLAG_WORDS, l->flags + flag_off + (size_t)l->rank * GNN_FLAG_WORDS,
                                      sizeof(int) * GNN_FLAG_WORDS, hipMemcpyDeviceToDevice, l->stream));
        }
        return GNN_OK;
    }
    size_t base = 0, unused = 0;
    loop_segment(l, 0, &base, &unused);
    NCCLCHK(g_rccl.GroupStart());
    if (b >= 0 && rows)
        NCCLCHK(g_rccl.AllGather(l->state[b] + off * l->Ds, l->state[b] + base * l->Ds, rows * l->Ds, NCCL_FLOAT32, l->comm->nccl, l->stream));
    if (flag_off != NO_FLAGS)
        NCCLCHK(g_rccl.AllGather(l->flags + flag_off + (size_t)l->rank * GNN_FLAG_WORDS, l->flags + flag_off, GNN_FLAG_WORDS, NCCL_INT32, l->comm->nccl, l->stream));
    NCCLCHK(g_rccl.GroupEnd());
    return GNN_OK;
}

// dst[r, 0:w) = src[r, 0:w) for r < n_rows (strided rows on both sides).  Own kernel instead of hipMemcpy2DAsync: the
// runtime's 2-D device-to-device copy was the only HIP call of the per-op path that the fused path never makes, and a
// profiled run (rocprofv3 --kernel-trace) of the per-op path died inside the runtime on its first use.
__global__ void k_copy_cols(int64_t n_rows, int w, const float *__restrict__ src, int64_t lds_, float *__restrict__ dst, int64_t ldd,
                            const int *gate, int world)
{
    if (!gnn_gate_open_block(gate, world)) return;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n_rows * w) return;
    const int64_t r = t / w;
    const int c = (int)(t - r * w);
    dst[r * ldd + c] = src[r * lds_ + c];
}

int gnn_launch_copy_cols(hipStream_t st, int64_t n_rows, int w, const float *src, int64_t lds_, float *dst, int64_t ldd, const int *gate, int world)
{
    if (n_rows == 0 || w == 0) return GNN_OK;
    hipLaunchKernelGGL(k_copy_cols, cdiv(n_rows * w, 256), 256, 0, st, n_rows, w, src, lds_, dst, ldd, gate, world);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

// ---- feature-sliced exchange --------------------------------------------------------------------------------------------
// The aggregation state_agg = Adjacency^T . state (GNN.py:234) is independent per COLUMN of the state.  Instead of handing every
// rank every row of the state (full or boundary replicas: (P - 1) / P of N Ds floats received per rank and iteration), rank q
// aggregates columns [q Cs, (q + 1) Cs), Cs = Ds / P, for ALL nodes over the whole graph's adjacency, and two all-to-all steps move
// column slices in and aggregated slices back: 2 (P - 1) / P of (N / P) Ds floats per rank and iteration, P / 2 times less
// (56 MB instead of 224 MB at N = 1 M, Ds = 64, P = 8).  The fmaf chain of an aggregated element is the same CSR-ordered chain
// as in the replicated layouts, so the results are bit-identical.
// VEC = 4 when Cs is a multiple of 4 (16-byte pieces), else 1; one thread per piece
template <int VEC>
__global__ void k_slice_pack(int64_t n_rows, int64_t shard_rows, int Ds, int Cs, const float *__restrict__ own, float *__restrict__ send,
                             const int *gate, int world)
{
    if (!gnn_gate_open_block(gate, world)) return;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pr = Ds / VEC;                                        // pieces per row
    if (t >= shard_rows * pr) return;
    const int64_t r = t / pr;
    const int f = (int)(t - r * pr) * VEC, q = f / Cs, c = f - q * Cs;
    float *dst = send + ((size_t)q * shard_rows + r) * Cs + c;
    if (VEC == 4) *reinterpret_cast<float4 *>(dst) = r < n_rows ? *reinterpret_cast<const float4 *>(own + r * Ds + f) : float4{0.f, 0.f, 0.f, 0.f};
    else *dst = r < n_rows ? own[r * Ds + f] : 0.0f;               // padding rows of a short shard travel as zeros
}

template <int VEC>
__global__ void k_slice_unpack(int64_t n_rows, int64_t shard_rows, int Ds, int Cs, const float *__restrict__ recv, float *__restrict__ agg,
                               const int *gate, int world)
{
    if (!gnn_gate_open_block(gate, world)) return;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int pr = Ds / VEC;
    if (t >= n_rows * pr) return;
    const int64_t r = t / pr;
    const int f = (int)(t - r * pr) * VEC, q = f / Cs, c = f - q * Cs;
    const float *src = recv + ((size_t)q * shard_rows + r) * Cs + c;
    if (VEC == 4) *reinterpret_cast<float4 *>(agg + r * Ds + f) = *reinterpret_cast<const float4 *>(src);
    else agg[r * Ds + f] = *src;
}

// all-to-all of equal blocks: block q of `send` goes to rank q, which stores it as block `rank` of its `recv`
// (which: 0 = sl_send -> peers' sl_state, 1 = sl_agg -> peers' sl_recv)
static int slice_alltoall(gnn_loop *l, int which)
{
    const size_t block = (size_t)l->shard_rows * l->Cs;
    float *send = which == 0 ? l->sl_send : l->sl_agg;
    if (l->comm->grp) {
        gnn_comm_group *grp = l->comm->grp;
        for (int p = 0; p < l->world; ++p) {
            gnn_loop *peer = grp->member[p];
            if (!peer) return gnn_fail(GNN_ERR_STATE, "loopback rank %d has no loop: create one loop per rank and run them with gnn_loop_run_group", p);
            if (!peer->slice_mode) return gnn_fail(GNN_ERR_STATE, "rank %d does not use the feature-sliced exchange", p);
            float *dst = (which == 0 ? peer->sl_state : peer->sl_recv) + (size_t)l->rank * block;
            HIPCHK(hipMemcpyAsync(dst, send + (size_t)p * block, sizeof(float) * block, hipMemcpyDeviceToDevice, l->stream));
        }
        return GNN_OK;
    }
    float *recv = which == 0 ? l->sl_state : l->sl_recv;
    // the rank's own block is a device copy; the other P - 1 pairs are one grouped send / receive each
    HIPCHK(hipMemcpyAsync(recv + (size_t)l->rank * block, send + (size_t)l->rank * block, sizeof(float) * block, hipMemcpyDeviceToDevice, l->stream));
    NCCLCHK(g_rccl.GroupStart());
    for (int p = 0; p < l->world; ++p) {
        if (p == l->rank) continue;
        NCCLCHK(g_rccl.Send(send + (size_t)p * block, block, NCCL_FLOAT32, p, l->comm->nccl, l->stream));
        NCCLCHK(g_rccl.Recv(recv + (size_t)p * block, block, NCCL_FLOAT32, p, l->comm->nccl, l->stream));
    }
    NCCLCHK(g_rccl.GroupEnd());
    return GNN_OK;
}

// the three steps of the sliced aggregation of body k; between them the other ranks of a loopback group take their turn
static int slice_step_pack(gnn_loop *l, int k)
{
    const int *gate = l->flags + (size_t)k * l->world * GNN_FLAG_WORDS;
    const float *own = l->state[k & 1] + (size_t)l->own_off * l->Ds;
    if (l->Cs % 4 == 0)
        hipLaunchKernelGGL((k_slice_pack<4>), cdiv(l->shard_rows * (l->Ds / 4), 256), 256, 0, l->stream, l->g->n_rows, l->shard_rows, l->Ds, l->Cs, own,
                           l->sl_send, gate, l->world);
    else
        hipLaunchKernelGGL((k_slice_pack<1>), cdiv(l->shard_rows * l->Ds, 256), 256, 0, l->stream, l->g->n_rows, l->shard_rows, l->Ds, l->Cs, own,
                           l->sl_send, gate, l->world);
    HIPCHK(hipGetLastError());
    return slice_alltoall(l, 0);
}

// Aggregation of the rank's column slice + the return all-to-all.  Pipelined form (default): the rows are aggregated in P blocks, one per
// destination rank, in the order rank + 1, rank + 2, ..., rank (every step of the schedule is a permutation: at step t rank r sends
// to r + 1 + t and receives from r - 1 - t, so no link carries two blocks at once), and block t travels on the communicator's second
// stream while block t + 1 is aggregated on the loop's stream; only the last block (the rank's own: a device copy) is exposed.
static int slice_step_aggregate(gnn_loop *l, int k)
{
    const gnn_graph *g = l->g;
    const int P = l->world;
    const int *gate = l->flags + (size_t)k * P * GNN_FLAG_WORDS;
    if (!l->sl_pipeline) {
        int rc = gnn_launch_spmm(l->stream, g->sh->full_rows, g->sh->full_indptr, g->sh->full_src, g->sh->full_w, l->sl_state, l->Cs, l->Cs, l->sl_agg, l->Cs, gate, P);
        if (rc) return rc;
        return slice_alltoall(l, 1);
    }
    gnn_comm *cm = l->comm;
    if (!cm->xstream) HIPCHK(hipStreamCreateWithFlags(&cm->xstream, hipStreamNonBlocking));
    if ((int)l->sl_ev.size() < P) {
        const size_t old = l->sl_ev.size();
        l->sl_ev.resize((size_t)P, nullptr);
        for (size_t i = old; i < l->sl_ev.size(); ++i) HIPCHK(hipEventCreateWithFlags(&l->sl_ev[i], hipEventDisableTiming));
    }
    if (!l->sl_done) HIPCHK(hipEventCreateWithFlags(&l->sl_done, hipEventDisableTiming));
    const size_t block = (size_t)l->shard_rows * l->Cs;
    for (int t = 0; t < P; ++t) {
        const int q = (l->rank + 1 + t) % P, from = ((l->rank - 1 - t) % P + P) % P;
        const int64_t r0 = (int64_t)q * l->shard_rows, r1 = std::min<int64_t>(r0 + l->shard_rows, g->sh->full_rows);
        if (r1 > r0) {
            int rc = gnn_launch_spmm(l->stream, r1 - r0, g->sh->full_indptr + r0, g->sh->full_src, g->sh->full_w, l->sl_state, l->Cs, l->Cs,
                                     l->sl_agg + (size_t)r0 * l->Cs, l->Cs, gate, P);
            if (rc) return rc;
        }
        HIPCHK(hipEventRecord(l->sl_ev[t], l->stream));
        HIPCHK(hipStreamWaitEvent(cm->xstream, l->sl_ev[t], 0));
        const float *src = l->sl_agg + (size_t)q * block;
        if (cm->grp) {
            gnn_loop *peer = cm->grp->member[q];
            if (!peer) return gnn_fail(GNN_ERR_STATE, "loopback rank %d has no loop: create one loop per rank and run them with gnn_loop_run_group", q);
            if (!peer->slice_mode) return gnn_fail(GNN_ERR_STATE, "rank %d does not use the feature-sliced exchange", q);
            HIPCHK(hipMemcpyAsync(peer->sl_recv + (size_t)l->rank * block, src, sizeof(float) * block, hipMemcpyDeviceToDevice, cm->xstream));
        } else if (q == l->rank) {
            HIPCHK(hipMemcpyAsync(l->sl_recv + (size_t)l->rank * block, src, sizeof(float) * block, hipMemcpyDeviceToDevice, cm->xstream));
        } else {
            NCCLCHK(g_rccl.GroupStart());
            NCCLCHK(g_rccl.Send(src, block, NCCL_FLOAT32, q, cm->nccl, cm->xstream));
            NCCLCHK(g_rccl.Recv(l->sl_recv + (size_t)from * block, block, NCCL_FLOAT32, from, cm->nccl, cm->xstream));
            NCCLCHK(g_rccl.GroupEnd());
        }
    }
    HIPCHK(hipEventRecord(l->sl_done, cm->xstream));
    return GNN_OK;
}

static int slice_step_unpack(gnn_loop *l, int k)
{
    const int *gate = l->flags + (size_t)k * l->world * GNN_FLAG_WORDS;
    if (l->sl_pipeline) {          // the blocks of this rank's rows have arrived: its own transfers (RCCL: each carries the matching receive), or every member's (loopback: they push)
        if (l->comm->grp) {
            for (int p = 0; p < l->world; ++p) {
                gnn_loop *peer = l->comm->grp->member[p];
                if (peer && peer->sl_done) HIPCHK(hipStreamWaitEvent(l->stream, peer->sl_done, 0));
            }
        } else if (l->sl_done) HIPCHK(hipStreamWaitEvent(l->stream, l->sl_done, 0));
    }
    if (l->g->n_rows) {
        if (l->Cs % 4 == 0)
            hipLaunchKernelGGL((k_slice_unpack<4>), cdiv(l->g->n_rows * (l->Ds / 4), 256), 256, 0, l->stream, l->g->n_rows, l->shard_rows, l->Ds, l->Cs,
                               l->sl_recv, l->agg_own, gate, l->world);
        else
            hipLaunchKernelGGL((k_slice_unpack<1>), cdiv(l->g->n_rows * l->Ds, 256), 256, 0, l->stream, l->g->n_rows, l->shard_rows, l->Ds, l->Cs,
                               l->sl_recv, l->agg_own, gate, l->world);
        HIPCHK(hipGetLastError());
    }
    return GNN_OK;
}

extern "C" int gnn_loop_set_slice_exchange(gnn_loop *l, int on)
{
    ARGCHK(l, "loop is NULL");
    if (!on) { l->slice_mode = false; return GNN_OK; }
    ARGCHK(l->world > 1, "the feature-sliced exchange needs a communicator");
    ARGCHK(l->Ds % l->world == 0, "state width %d is not a multiple of the world size %d", l->Ds, l->world);
    ARGCHK(l->g->sh->full_indptr, "call gnn_graph_set_full_adjacency first (on this graph or on the graph it was derived from)");
    ARGCHK(!l->g->halo_world, "not on a boundary-exchange shard");
    HIPCHK(hipSetDevice(l->device));
    l->Cs = l->Ds / l->world;
    if (!l->sl_send) {
        const size_t slice = (size_t)l->N_pad * l->Cs;           // == world * shard_rows * Cs
        int rc = dev_alloc(&l->sl_send, slice);
        if (!rc) rc = dev_alloc(&l->sl_state, slice);
        if (!rc) rc = dev_alloc(&l->sl_agg, slice);
        if (!rc) rc = dev_alloc(&l->sl_recv, slice);
        if (!rc) rc = dev_alloc(&l->agg_own, (size_t)l->shard_rows * l->Ds);
        if (rc) return rc;
        if ((rc = zero_on_stream(l->sl_agg, sizeof(float) * std::max<size_t>(slice, 1), l->stream))) return rc;      // rows past N_global are never written
    }
    l->slice_mode = true;
    l->sl_pipeline = on != 2;          // 2: the whole slice is aggregated, then one grouped all-to-all (round-2 form; kept for comparison)
    return GNN_OK;
}

static int unfused_iteration(gnn_loop *l, int k)
{
    const gnn_graph *g = l->g;
    const int cur = k & 1, nxt = cur ^ 1, P = l->world;
    const int *gate = l->flags + (size_t)k * P * GNN_FLAG_WORDS;
    const float *own_cur = l->state[cur] + (size_t)l->own_off * l->Ds;
    float *own_nxt = l->state[nxt] + (size_t)l->own_off * l->Ds;
    // node_components (GNN.py:228): own state into columns [0, Ds) of the concat
    int rc = gnn_launch_copy_cols(l->stream, g->n_rows, l->Ds, own_cur, l->Ds, l->inp, l->in_s, gate, P);
    if (rc) return rc;
    // aggregated_states (GNN.py:234) into columns [Ds + NLc, +Ds)
    if (l->slice_mode) rc = gnn_launch_copy_cols(l->stream, g->n_rows, l->Ds, l->agg_own, l->Ds, l->inp + l->Ds + l->NLc, l->in_s, gate, P);
    else rc = gnn_launch_spmm(l->stream, g->n_rows, g->sh->indptr, g->sh->adj_src, g->sh->adj_w, l->state[cur], l->Ds, l->Ds,
                              l->inp + l->Ds + l->NLc, l->in_s, gate, P);
    if (rc) return rc;
    // net_state (GNN.py:240)
    rc = launch_mlp(l->stream, l->st, g->n_rows, l->inp, l->in_s, own_nxt, l->Ds, l->tmp[0], l->tmp[1], gate, P);
    if (rc) return rc;
    // condition for the next body (GNN.py:206-218)
    return launch_check(l->stream, g->n_rows, l->Ds, own_nxt, own_cur, l->thr, l->flags + ((size_t)(k + 1) * P + l->rank) * GNN_FLAG_WORDS, gate, P);
}

// ---- phases of one Loop; gnn_loop_run runs them for one rank, gnn_loop_run_group rank by rank for a loopback group ----------
static inline bool loop_is_fused(gnn_loop *l) { return l->impl_req >= 1 && gnn_fused_supported(l); }

// state <- initial state, first condition against ones (GNN.py:262-271), loop-invariant aggregates (GNN.py:259, :263)
static int loop_begin(gnn_loop *l, bool fused)
{
    gnn_graph *g = l->g;
    const int P = l->world;
    hipStream_t st = l->stream;
    int rc = 0;
    HIPCHK(hipMemsetAsync(l->flags, 0, sizeof(int) * (size_t)(l->max_iter + 2) * P * GNN_FLAG_WORDS, st));
    l->small_words_clean = false;       // the persistent loop's gate words share this block
    HIPCHK(hipMemsetAsync(l->tile_ctr, 0, sizeof(int) * ((2 * ((size_t)l->max_iter + 1) + 3) & ~(size_t)3), st));
    float *own0 = l->state[0] + (size_t)l->own_off * l->Ds;
    if (g->n_rows)   // state <- nodes (GNN.py:265) or the injected / drawn initial state (GNN.py:262)
        HIPCHK(hipMemcpyAsync(own0, l->D ? l->state_init : g->nodes + (size_t)g->own_off * g->NL,
                              sizeof(float) * (size_t)g->n_rows * l->Ds, hipMemcpyDeviceToDevice, st));
    // first condition: state vs ones (GNN.py:266, :271)
    if ((rc = launch_check(st, g->n_rows, l->Ds, own0, nullptr, l->thr, l->flags + (size_t)l->rank * GNN_FLAG_WORDS, nullptr, 1))) return rc;
    if (!fused) {
        const int c_nodes = l->Ds, c_aggn = l->Ds + l->NLc + l->Ds, c_agga = c_aggn + l->NLc;
        rc = gnn_launch_spmm(st, g->n_rows, g->sh->indptr, nullptr, g->sh->arc_w, gnn_graph_arc_labels(g), g->AL, g->AL, l->inp + c_agga, l->in_s, nullptr, 1);
        if (rc) return rc;
        if (l->D) {
            rc = gnn_launch_spmm(st, g->n_rows, g->sh->indptr, g->sh->adj_src, g->sh->adj_w, g->nodes, g->NL, g->NL, l->inp + c_aggn, l->in_s, nullptr, 1);
            if (rc) return rc;
            rc = gnn_launch_copy_cols(st, g->n_rows, g->NL, g->nodes + (size_t)g->own_off * g->NL, g->NL, l->inp + c_nodes, l->in_s, nullptr, 1);
            if (rc) return rc;
        }
    }
    return GNN_OK;
}

static int loop_body(gnn_loop *l, int k, bool fused)
{
    if (l->profiling) HIPCHK(hipEventRecord(l->ev[2 * k], l->stream));
    int rc = fused ? gnn_fused_iteration(l, k) : unfused_iteration(l, k);
    if (rc) return rc;
    if (l->profiling) HIPCHK(hipEventRecord(l->ev[2 * k + 1], l->stream));
    return GNN_OK;
}

// gate of body k -> host; every rank reads the same exchanged gate, so all ranks stop at the same body
static int loop_gate_closed(gnn_loop *l, int k, bool *closed)
{
    const size_t words = (size_t)l->world * GNN_FLAG_WORDS;
    HIPCHK(hipMemcpyAsync(l->gate_host, l->flags + (size_t)k * words, sizeof(int) * words, hipMemcpyDeviceToHost, l->stream));
    HIPCHK(hipStreamSynchronize(l->stream));
    int any = 0;
    for (size_t i = 0; i < words; i += GNN_FLAG_STRIDE) any |= l->gate_host[i];
    *closed = !any;
    return GNN_OK;
}

// k, apply_filters + net_output on the owned masked rows (GNN.py:275-279)
static int loop_finish(gnn_loop *l, bool finalize, bool output_done)
{
    gnn_graph *g = l->g;
    hipStream_t st = l->stream;
    int rc = 0;
    if (finalize) {        // (the persistent small-graph loop has written k itself)
        hipLaunchKernelGGL(k_finalize, 1, 64, 0, st, l->flags, l->world, l->max_iter, l->kfinal_dev);
        HIPCHK(hipGetLastError());
    }
    if (finalize) {
        HIPCHK(hipMemcpyAsync(l->kfinal_host, l->kfinal_dev, sizeof(int), hipMemcpyDeviceToHost, st));
        HIPCHK(hipMemcpyAsync(l->kfinal_host + 2, l->kfinal_dev + 2, sizeof(int), hipMemcpyDeviceToHost, st));      // "a gate was not certified"
    }
    if (output_done) return GNN_OK;     // (the persistent loop wrote k into the pinned host words and ran the output stage itself)
    const float *own0 = l->state[0] + (size_t)l->own_off * l->Ds, *own1 = l->state[1] + (size_t)l->own_off * l->Ds;
    const float *nodes_own = g->nodes + (size_t)g->own_off * g->NL;

    if (l->edge_mode) {      // GNNedgeBased.apply_filters + net_output on the masked arcs (GNN.py:289-302, :279)
        if (l->n_edge_masked) {
            const int we = l->ou->dims[0];
            const int64_t tot = l->n_edge_masked * we;
            hipLaunchKernelGGL(k_feats_edge, cdiv(tot, 256), 256, 0, st, l->n_edge_masked, l->edge_rows, l->edge_dst, g->sh->adj_src, l->state[0],
                               l->state[1], l->kfinal_dev, l->Ds, g->nodes, g->NL, l->NLc, g->arc_labels_orig_own ? g->arc_labels_orig_own : l->edge_labels, g->AL, l->feats, l->own_off);
            HIPCHK(hipGetLastError());
            rc = launch_mlp(st, l->ou, l->n_edge_masked, l->feats, we, l->out, l->T, l->otmp[0], l->otmp[1], nullptr, 1);
            if (rc) return rc;
        }
        return GNN_OK;
    }
    const size_t out1_lds = sizeof(float) * ((size_t)(l->wf + 1) * l->T + (size_t)GNN_OUT1_ROWS * (l->wf | 1) + (size_t)GNN_OUT1_ROWS * l->T);
    if (g->n_masked && l->ou->n_layers == 1 && l->T <= 8 && out1_lds <= 64 * 1024) {
        const gnn_mlp *ou = l->ou;
        hipLaunchKernelGGL(k_out1, cdiv(g->n_masked, GNN_OUT1_ROWS), GNN_OUT1_ROWS * l->T < 64 ? 64 : GNN_OUT1_ROWS * l->T, out1_lds, st, g->n_masked, g->sh->masked_rows,
                           own0, own1, l->kfinal_dev, l->Ds, nodes_own, g->NL, l->NLc, ou->W[0], ou->b[0], l->T, ou->acts[0],
                           ou->has_bn ? ou->bn_scale : (const float *)nullptr, ou->has_bn ? ou->bn_shift : (const float *)nullptr, l->out);
        HIPCHK(hipGetLastError());
    } else if (g->n_masked) {
        const int64_t tot = g->n_masked * l->wf;
        hipLaunchKernelGGL(k_feats, cdiv(tot, 256), 256, 0, st, g->n_masked, g->sh->masked_rows, own0, own1, l->kfinal_dev, l->Ds, nodes_own, g->NL, l->NLc, l->feats);
        HIPCHK(hipGetLastError());
        rc = launch_mlp(st, l->ou, g->n_masked, l->feats, l->wf, l->out, l->T, l->otmp[0], l->otmp[1], nullptr, 1);
        if (rc) return rc;
    }
    return GNN_OK;
}

static int loop_prepare(gnn_loop *l, bool *fused_out)
{
    if (!l->have_state0) {
        if (l->D == 0) l->have_state0 = true;
        else return gnn_fail(GNN_ERR_STATE, "state_vect_dim > 0: call gnn_loop_set_state0 first");
    }
    if (l->edge_expected && !l->edge_mode)
        return gnn_fail(GNN_ERR_STATE, "net_output has the edge-based input width: call gnn_loop_set_edge_readout first");
    HIPCHK(hipSetDevice(l->device));
    l->ng_inlaunch = false;          // (set again by gnn_small_run when it folds the graph readout into its launch)
    ++l->out_runs;                   // this run rewrites l->out: a readout folded into an earlier launch is stale from here on
    if (!l->graph_ready_seen) {      // a derived graph's creation-time fills (gnn_graph_derive) come before the first read of its labels
        int rcw = gnn_graph_wait_ready(l->g, l->stream);
        if (rcw) return rcw;
        l->graph_ready_seen = true;
    }
    const bool fused = loop_is_fused(l);
    l->impl_used = fused ? l->impl_req : 0;
    int rc = fused ? gnn_fused_prepare(l) : loop_ensure_unfused(l);
    if (rc) return rc;
    if (l->profiling && (int)l->ev.size() < 2 * l->max_iter) {
        const size_t old = l->ev.size();
        l->ev.resize(2 * (size_t)l->max_iter, nullptr);
        for (size_t i = old; i < l->ev.size(); ++i) HIPCHK(hipEventCreate(&l->ev[i]));
    }
    *fused_out = fused;
    return GNN_OK;
}

static int loop_collect(gnn_loop *l, float *k_out)
{
    l->kfinal = *l->kfinal_host;
    l->ran = true;
    l->total_ms = 0.f;
    if (l->profiling) HIPCHK(hipEventElapsedTime(&l->total_ms, l->ev_total[0], l->ev_total[1]));
    l->avg_iter_ms = 0.f;
    l->n_iter_timed = 0;
    if (l->profiling) {
        double sum = 0;
        for (int k = 0; k < l->kfinal; ++k) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, l->ev[2 * k], l->ev[2 * k + 1]));
            sum += ms;
        }
        l->n_iter_timed = l->kfinal;
        l->avg_iter_ms = l->kfinal ? (float)(sum / l->kfinal) : 0.f;
        // what sits between two bodies on the stream: the exchange of the sharded layouts (all-gather of rows / boundary rows + flags, or
        // pack -> all-to-all -> slice aggregation -> all-to-all -> unpack), a few microseconds of launch gap on a single GPU
        double gap = 0;
        for (int k = 0; k + 1 < l->kfinal; ++k) {
            float ms = 0;
            HIPCHK(hipEventElapsedTime(&ms, l->ev[2 * k + 1], l->ev[2 * k + 2]));
            gap += ms;
        }
        l->avg_gap_ms = l->kfinal > 1 ? (float)(gap / (l->kfinal - 1)) : 0.f;
    }
    if (k_out) *k_out = (float)l->kfinal;
    return GNN_OK;
}

// Everything the ranks in `ls` put on their streams for one Loop.  Bodies are enqueued without waiting for each other;
// every GNN_BODY_CHUNK bodies the gate of the next body is copied to the host and checked, so that a loop that converged
// does not pay for max_iteration - k empty launches (about 3 us each).  n == 1: one rank of an RCCL job (or a single GPU);
// n == world: all ranks of a loopback group, stepped phase by phase on the group's stream.
static int run_loops(gnn_loop **ls, int n, float *k_out)
{
    std::vector<char> fused((size_t)n, 0);
    int rc = 0;
    for (int r = 0; r < n; ++r) {
        bool f = false;
        if ((rc = loop_prepare(ls[r], &f))) return rc;
        fused[r] = f;
    }
    const int max_iter = ls[0]->max_iter;
    // small graphs: the initial state, the first condition and every body inside ONE persistent launch (gnn_small.hip)
    const bool small = n == 1 && fused[0] && gnn_small_supported(ls[0]);
    bool output_done = false;
    for (int r = 0; r < n; ++r) {
        if (ls[r]->profiling) HIPCHK(hipEventRecord(ls[r]->ev_total[0], ls[r]->stream));
        if (!small && (rc = loop_begin(ls[r], fused[r]))) return rc;
    }
    // (feature-sliced exchange: no rank ever needs another rank's state rows, only the gates travel)
    for (int r = 0; r < n && !small; ++r) if ((rc = loop_exchange(ls[r], ls[r]->slice_mode ? -1 : 0, 0))) return rc;
    if (small) {
        ls[0]->kfinal_host[1] = 0;
        if ((rc = gnn_small_run(ls[0], &output_done))) return rc;
    }
    for (int k = 0; k < max_iter && !small; ++k) {
        if (ls[0]->slice_mode) {
            for (int r = 0; r < n; ++r) if ((rc = slice_step_pack(ls[r], k))) return rc;
            for (int r = 0; r < n; ++r) if ((rc = slice_step_aggregate(ls[r], k))) return rc;
            for (int r = 0; r < n; ++r) if ((rc = slice_step_unpack(ls[r], k))) return rc;
        }
        for (int r = 0; r < n; ++r) if ((rc = loop_body(ls[r], k, fused[r]))) return rc;
        for (int r = 0; r < n; ++r)
            if ((rc = loop_exchange(ls[r], ls[r]->slice_mode ? -1 : (k & 1) ^ 1, (size_t)(k + 1) * ls[r]->world * GNN_FLAG_WORDS))) return rc;
        if ((k + 1) % GNN_BODY_CHUNK == 0 && k + 1 < max_iter) {
            bool closed = false, c = false;
            for (int r = 0; r < n; ++r) {
                if ((rc = loop_gate_closed(ls[r], k + 1, &c))) return rc;
                if (r == 0) closed = c;
                else if (c != closed) return gnn_fail(GNN_ERR_STATE, "ranks disagree on the gate of body %d", k + 1);
            }
            if (closed) break;
        }
    }
    for (int r = 0; r < n; ++r) {
        if ((rc = loop_finish(ls[r], !small, output_done))) return rc;
        if (ls[r]->profiling) HIPCHK(hipEventRecord(ls[r]->ev_total[1], ls[r]->stream));
    }
    for (int r = 0; r < n; ++r) HIPCHK(hipStreamSynchronize(ls[r]->stream));
    if (small && ls[0]->kfinal_host[1] != 0) {       // a barrier spin gave up (grid not resident?): repeat with one launch per body
        ls[0]->ng_inlaunch = false;
        ls[0]->small_disabled = true;
        ls[0]->small_words_clean = false;
        return run_loops(ls, n, k_out);
    }
    // Certified gate (gnn_common.h): a gate of this impl-2 run was decided by a borderline node and no robust mover - its k is not
    // guaranteed to be the bit-exact chain's.  The Loop is repeated on impl 1 and THAT run's k / state / output are what the caller gets.
    // Every rank reads the same exchanged flag words, so all ranks of a sharded job take this branch together.
    for (int r = 0; r < n; ++r) ls[r]->last_run_rerun = false;
    if (!small && fused[0] && ls[0]->impl_req == 2 && ls[0]->kfinal_host[2] != 0) {
        static bool told = false;       // once per process: the caller gets the exact path's results, at the exact path's price
        if (!told && !getenv("GNN_QUIET")) {
            told = true;
            fprintf(stderr, "libgnn_hip: a gate of a default-path Loop was decided by a borderline node (no robust mover): the Loop is repeated on the "
                            "bit-exact path and its k / state / output are returned (gnn_loop_gate_info counts these; gnn_loop_set_impl(l, 1) avoids the double run)\n");
        }
        for (int r = 0; r < n; ++r) ls[r]->impl_req = 1;
        rc = run_loops(ls, n, k_out);
        for (int r = 0; r < n; ++r) {
            ls[r]->impl_req = 2;
            if (rc == GNN_OK) { ls[r]->last_run_rerun = true; ++ls[r]->certified_reruns; }
        }
        return rc;
    }
    for (int r = 0; r < n; ++r) {
        float k = 0.f;
        if ((rc = loop_collect(ls[r], &k))) return rc;
        if (r && ls[r]->kfinal != ls[0]->kfinal) return gnn_fail(GNN_ERR_STATE, "ranks disagree on the iteration count (%d vs %d)", ls[r]->kfinal, ls[0]->kfinal);
        if (r == 0 && k_out) *k_out = k;
    }
    return GNN_OK;
}

extern "C" int gnn_loop_run(gnn_loop *l, int training, float *k_out)
{
    ARGCHK(l, "loop is NULL");
    if (training) return gnn_fail(GNN_ERR_UNSUPPORTED, "gnn_loop_run is the inference Loop; the training-mode Loop is gnn_loop_train_forward / gnn_loop_train_step");
    if (l->comm && l->comm->grp && l->world > 1)
        return gnn_fail(GNN_ERR_STATE, "this loop belongs to a loopback group of %d ranks: run all of them with gnn_loop_run_group", l->world);
    return run_loops(&l, 1, k_out);
}

// Several INDEPENDENT loops (batches of a dataset: reference GNN_BaseClass.py:165-189 evaluates them one after the other) in one call.
// Small graphs take the persistent one-launch path, and such a launch occupies a few dozen of the 256 CUs: all of them are queued, each on
// its own loop's stream, before the first is waited for, so that they run side by side; the others run one after the other (each
// fills the GPU by itself).  The results are those of n separate gnn_loop_run calls.
extern "C" int gnn_loop_run_many(gnn_loop **loops, int n, float *k_out /* [n] */)
{
    ARGCHK(loops && n >= 1 && k_out, "bad arguments");
    for (int i = 0; i < n; ++i) {
        ARGCHK(loops[i], "loop %d is NULL", i);
        ARGCHK(loops[i]->world == 1, "loop %d is one rank of a sharded job: gnn_loop_run / gnn_loop_run_group", i);
        for (int j = 0; j < i; ++j) ARGCHK(loops[j] != loops[i], "loop %d is listed twice", i);
    }
    std::vector<char> queued((size_t)n, 0), done((size_t)n, 0);
    int rc = 0;
    // (an error part-way: nothing queued is left running behind the caller's back - EVERY early return below goes through drain)
    auto drain = [&](int rc_) { for (int i = 0; i < n; ++i) if (queued[(size_t)i] && !done[(size_t)i]) (void)hipStreamSynchronize(loops[i]->stream); return rc_; };
    // Residency: the persistent launches synchronise through a grid barrier, so every workgroup of every launch in flight must be
    // resident at once.  A launch's workgroup is one wave with 10 - 40 KB of LDS: at least four fit on a CU; launches are queued side by
    // side only while their workgroups sum to no more than three per CU, then the queued ones are collected before the next is queued
    // (the barrier's spin time-out stays as the safety net, it is no longer the mechanism).
    int n_cu = 0;
    if (hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, loops[0]->device) != hipSuccess || n_cu <= 0) n_cu = 64;
    const long cap = 3L * n_cu;
    long in_flight = 0;
    // wait for the queued launches and take their results; one whose barrier gave up runs again, alone, one launch per body
    auto collect = [&]() -> int {
        int first_rc = GNN_OK;
        for (int i = 0; i < n; ++i) {
            if (!queued[(size_t)i] || done[(size_t)i]) continue;
            gnn_loop *l = loops[i];
            hipError_t e = hipSetDevice(l->device);
            if (e == hipSuccess) e = hipStreamSynchronize(l->stream);
            done[(size_t)i] = 1;
            if (e != hipSuccess) { if (!first_rc) first_rc = gnn_fail(GNN_ERR_HIP, "hipStreamSynchronize -> %s", hipGetErrorString(e)); continue; }
            if (first_rc) continue;                  // (keep waiting for the others, report the first error)
            int r = GNN_OK;
            if (l->kfinal_host[1] != 0) {            // a barrier spin gave up: this one again, alone, one launch per body
                l->ng_inlaunch = false;
                l->small_disabled = true;
                l->small_words_clean = false;
                r = run_loops(&loops[i], 1, &k_out[i]);
                l->small_disabled = false;           // (alone it would have been resident: only this call falls back)
            } else r = loop_collect(l, &k_out[i]);
            if (r) first_rc = r;
        }
        in_flight = 0;
        return first_rc;
    };
    for (int i = 0; i < n; ++i) {
        gnn_loop *l = loops[i];
        bool fused = false;
        if ((rc = loop_prepare(l, &fused))) return drain(rc);
        l->last_run_rerun = false;                      // (the persistent path is exact arithmetic and never repeats; run_loops sets it for the others)
        if (!(fused && gnn_small_supported(l))) continue;
        const long wgs = (long)((l->g->n_rows + 15) / 16);      // upper bound of the launch's grid (16- or 32-node tiles)
        if (in_flight && in_flight + wgs > cap && (rc = collect())) return drain(rc);
        bool output_done = false;
        l->kfinal_host[1] = 0;
        if ((rc = gnn_small_run(l, &output_done))) return drain(rc);
        queued[(size_t)i] = 1;
        in_flight += wgs;
        if ((rc = loop_finish(l, false, output_done))) return drain(rc);
    }
    for (int i = 0; i < n; ++i)
        if (!queued[(size_t)i] && (rc = run_loops(&loops[i], 1, &k_out[i]))) return drain(rc);
    if ((rc = collect())) return drain(rc);
    return GNN_OK;
}

extern "C" int gnn_loop_run_group(gnn_loop **loops, int n, float *k_out)
{
    ARGCHK(loops && n >= 1, "bad arguments");
    for (int r = 0; r < n; ++r) {
        ARGCHK(loops[r] && loops[r]->comm && loops[r]->comm->grp, "loops[%d] was not created on a loopback communicator", r);
        ARGCHK(loops[r]->comm->grp == loops[0]->comm->grp && loops[r]->world == n && loops[r]->rank == r, "loops must be the %d ranks of one loopback group, in rank order", n);
        ARGCHK(loops[r]->max_iter == loops[0]->max_iter && loops[r]->thr == loops[0]->thr && loops[r]->Ds == loops[0]->Ds, "ranks were configured differently");
        // the exchange protocol of the whole group follows rank 0: a rank on another layout / arithmetic would skip or misread an exchange
        ARGCHK(loops[r]->slice_mode == loops[0]->slice_mode && loops[r]->Cs == loops[0]->Cs && loops[r]->impl_req == loops[0]->impl_req &&
               (!loops[0]->slice_mode || loops[r]->sl_pipeline == loops[0]->sl_pipeline) &&
               (loops[r]->g->halo_world != 0) == (loops[0]->g->halo_world != 0),
               "rank %d uses another exchange layout or arithmetic (slice %d/%d, impl %d/%d) than rank 0", r, (int)loops[r]->slice_mode, (int)loops[0]->slice_mode,
               loops[r]->impl_req, loops[0]->impl_req);
    }
    return run_loops(loops, n, k_out);
}

extern "C" int gnn_loop_get_state(const gnn_loop *l, float *state_out)
{
    ARGCHK(l && state_out, "bad arguments");
    if (!l->ran) return gnn_fail(GNN_ERR_STATE, "gnn_loop_run has not been called");
    HIPCHK(hipSetDevice(l->device));
    const float *src = l->state[l->kfinal & 1] + (size_t)l->own_off * l->Ds;
    HIPCHK(hipMemcpy(state_out, src, sizeof(float) * (size_t)l->g->n_rows * l->Ds, hipMemcpyDeviceToHost));
    return GNN_OK;
}

extern "C" int gnn_loop_get_output(const gnn_loop *l, float *out, int64_t *n_masked)
{
    ARGCHK(l, "loop is NULL");
    if (!l->ran) return gnn_fail(GNN_ERR_STATE, "gnn_loop_run has not been called");
    const int64_t m = l->edge_mode ? l->n_edge_masked : l->g->n_masked;
    if (n_masked) *n_masked = m;
    if (out && m) {
        HIPCHK(hipSetDevice(l->device));
        HIPCHK(hipMemcpy(out, l->out, sizeof(float) * (size_t)m * l->T, hipMemcpyDeviceToHost));
    }
    return GNN_OK;
}

extern "C" int gnn_loop_set_edge_readout(gnn_loop *l, const int32_t *entry_dst, const float *arc_labels, const uint8_t *arc_mask)
{
    ARGCHK(l && (l->g->E == 0 || (entry_dst && arc_mask && (arc_labels || l->g->AL == 0 || l->g->arc_labels_orig_own))), "bad arguments");
    ARGCHK(l->edge_expected, "net_output input width %d is not the edge-based 2 (NL + D) + AL", l->ou->dims[0]);
    // Sharded loops (round 3): a rank reads out the arcs of its OWN CSR rows - entry_dst are owned-row indices, the source endpoint is in
    // the replica's index space like every adj_src - so the per-rank outputs, in rank order, are the unsharded output.  (Training and
    // the arc-side LGNN relabelling stay single-GPU.)
    const gnn_graph *g = l->g;
    std::vector<int32_t> rows;
    for (int64_t e = 0; e < g->E; ++e) {
        ARGCHK(entry_dst[e] >= 0 && entry_dst[e] < g->n_rows, "entry_dst[%lld]=%d outside the %lld owned rows", (long long)e, entry_dst[e], (long long)g->n_rows);
        if (arc_mask[e]) rows.push_back((int32_t)e);
    }
    HIPCHK(hipSetDevice(l->device));
    (void)hipFree(l->edge_dst); (void)hipFree(l->edge_rows); (void)hipFree(l->edge_labels); (void)hipFree(l->edge_inc_ptr); (void)hipFree(l->edge_inc);
    (void)hipFree(l->feats); (void)hipFree(l->out); (void)hipFree(l->otmp[0]); (void)hipFree(l->otmp[1]);
    l->feats = l->out = l->otmp[0] = l->otmp[1] = nullptr;
    l->edge_dst = l->edge_rows = nullptr; l->edge_labels = nullptr; l->edge_inc_ptr = l->edge_inc = nullptr;
    l->n_edge_masked = (int64_t)rows.size();
    int maxw_o = 1;
    for (int i = 1; i <= l->ou->n_layers; ++i) maxw_o = std::max(maxw_o, l->ou->dims[i]);
    int rc = dev_upload(&l->edge_dst, entry_dst, (size_t)g->E);
    if (!rc) rc = dev_upload(&l->edge_rows, rows.data(), rows.size());
    if (!rc && !g->arc_labels_orig_own) rc = dev_upload(&l->edge_labels, arc_labels, (size_t)g->E * g->AL);   // derived graphs own theirs
    if (!rc && l->world == 1 && g->n_rows == g->N) {      // masked arcs by endpoint (ascending masked-arc index): what the (single-GPU) training backward gathers per node
        std::vector<int32_t> src((size_t)g->E);
        if (g->E) HIPCHK(hipMemcpy(src.data(), g->sh->adj_src, sizeof(int32_t) * (size_t)g->E, hipMemcpyDeviceToHost));
        std::vector<int32_t> ptr((size_t)g->N + 1, 0), inc(2 * rows.size());
        for (int32_t e : rows) { ++ptr[(size_t)entry_dst[e] + 1]; ++ptr[(size_t)src[e] + 1]; }
        for (int64_t i = 0; i < g->N; ++i) ptr[i + 1] += ptr[i];
        std::vector<int32_t> fill(ptr.begin(), ptr.end() - 1);
        for (size_t q = 0; q < rows.size(); ++q) {
            const int32_t e = rows[q];
            inc[fill[entry_dst[e]]++] = (int32_t)(q << 1);
            inc[fill[src[e]]++] = (int32_t)(q << 1) | 1;
        }
        rc = dev_upload(&l->edge_inc_ptr, ptr.data(), ptr.size());
        if (!rc) rc = dev_upload(&l->edge_inc, inc.data(), inc.size());
    }
    if (!rc) rc = dev_alloc(&l->feats, rows.size() * (size_t)l->ou->dims[0]);
    if (!rc) rc = dev_alloc(&l->out, rows.size() * (size_t)l->T);
    for (int b = 0; b < 2 && !rc; ++b) rc = dev_alloc(&l->otmp[b], rows.size() * (size_t)maxw_o);
    if (rc) return rc;
    l->edge_mode = true;
    l->ran = false;
    return GNN_OK;
}

// NodeGraph^T on the device (kept with the loop, re-uploaded only when it changes) + the partial readout of the owned rows
static int readout_partial(gnn_loop *lm, int G, const int32_t *ng_indptr, const int32_t *ng_node, const float *ng_w)
{
    ARGCHK(lm && G > 0 && ng_indptr, "bad arguments");
    if (!lm->ran) return gnn_fail(GNN_ERR_STATE, "gnn_loop_run has not been called");
    ARGCHK(!lm->edge_mode, "graph readout of an edge-based loop");
    const gnn_graph *g = lm->g;
    ARGCHK(g->n_masked == g->n_rows, "graph-based readout needs all-true masks (GNN.py:275-276, :332): %lld of %lld owned rows are masked in",
           (long long)g->n_masked, (long long)g->n_rows);
    const int nnz = ng_indptr[G];
    ARGCHK(ng_indptr[0] == 0 && nnz >= 0 && (nnz == 0 || (ng_node && ng_w)), "bad NodeGraph CSR");
    for (int e = 0; e < nnz; ++e)
        ARGCHK(ng_node[e] >= 0 && ng_node[e] < g->N_global, "NodeGraph row %d but the graph has %lld nodes", ng_node[e], (long long)g->N_global);
    HIPCHK(hipSetDevice(lm->device));
    std::vector<int32_t> key(ng_indptr, ng_indptr + G + 1);
    key.insert(key.end(), ng_node, ng_node + nnz);
    const bool same = lm->ng_key == key && lm->ng_w_host.size() == (size_t)nnz &&
                      (nnz == 0 || memcmp(lm->ng_w_host.data(), ng_w, sizeof(float) * nnz) == 0);
    int rc = GNN_OK;
    if (same && gnn_loop_ng_folded(lm) && lm->ng_G == G) return GNN_OK;      // the persistent launch of this run has already computed it (gnn_small.hip)
    if (!same) {
        lm->ng_inlaunch = false;
        (void)hipFree(lm->ng_ip); (void)hipFree(lm->ng_node); (void)hipFree(lm->ng_w); (void)hipFree(lm->ng_out); (void)hipFree(lm->ng_part);
        lm->ng_ip = lm->ng_node = nullptr; lm->ng_w = lm->ng_out = lm->ng_part = nullptr;
        lm->ng_key.clear();
        rc = dev_upload(&lm->ng_ip, ng_indptr, (size_t)G + 1);
        if (!rc) rc = dev_upload(&lm->ng_node, ng_node, (size_t)nnz);
        if (!rc) rc = dev_upload(&lm->ng_w, ng_w, (size_t)nnz);
        if (!rc) rc = dev_alloc(&lm->ng_out, (size_t)G * lm->T);
        if (!rc && lm->world > 1) rc = dev_alloc(&lm->ng_part, (size_t)lm->world * G * lm->T);
        if (rc) return rc;
        lm->ng_key = key;
        lm->ng_w_host.assign(ng_w, ng_w + nnz);
        lm->ng_G = G;
    }
    float *dst = lm->world > 1 ? lm->ng_part + (size_t)lm->rank * G * lm->T : lm->ng_out;
    hipLaunchKernelGGL(k_readout, cdiv((int64_t)G * lm->T, 64), 64, 0, lm->stream, G, lm->T, lm->ng_ip, lm->ng_node, lm->ng_w, lm->out, g->row_begin, g->n_rows, dst);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

static int readout_combine(gnn_loop *lm, int G, float *out_graph)
{
    if (gnn_loop_ng_folded(lm) && lm->ng_G == G && lm->world == 1) {      // folded into the persistent launch: the result is in pinned host memory
        memcpy(out_graph, lm->ng_host, sizeof(float) * (size_t)G * lm->T);
        return GNN_OK;
    }
    if (lm->world > 1) {
        hipLaunchKernelGGL(k_sum_partials, cdiv((int64_t)G * lm->T, 64), 64, 0, lm->stream, G * lm->T, lm->world, lm->ng_part, lm->ng_out);
        HIPCHK(hipGetLastError());
    }
    HIPCHK(hipMemcpyAsync(out_graph, lm->ng_out, sizeof(float) * (size_t)G * lm->T, hipMemcpyDeviceToHost, lm->stream));
    HIPCHK(hipStreamSynchronize(lm->stream));
    return GNN_OK;
}

extern "C" int gnn_loop_readout(const gnn_loop *l, int G, const int32_t *ng_indptr, const int32_t *ng_node,
                                const float *ng_w, float *out_graph)
{
    ARGCHK(l && out_graph, "bad arguments");
    gnn_loop *lm = const_cast<gnn_loop *>(l);
    if (lm->comm && lm->comm->grp && lm->world > 1) return gnn_fail(GNN_ERR_STATE, "loopback group: use gnn_loop_readout_group");
    int rc = readout_partial(lm, G, ng_indptr, ng_node, ng_w);
    if (rc) return rc;
    if (lm->world > 1) {       // every rank gets every rank's [G, T] partial (a few KB), then adds them in rank order
        const size_t cnt = (size_t)G * lm->T;
        NCCLCHK(g_rccl.AllGather(lm->ng_part + cnt * lm->rank, lm->ng_part, cnt, NCCL_FLOAT32, lm->comm->nccl, lm->stream));
    }
    return readout_combine(lm, G, out_graph);
}

extern "C" int gnn_loop_readout_group(gnn_loop **loops, int n, int G, const int32_t *ng_indptr, const int32_t *ng_node, const float *ng_w,
                                      float *out_graph)
{
    ARGCHK(loops && n >= 1 && out_graph, "bad arguments");
    for (int r = 0; r < n; ++r)
        ARGCHK(loops[r] && loops[r]->comm && loops[r]->comm->grp && loops[r]->comm->grp == loops[0]->comm->grp && loops[r]->world == n && loops[r]->rank == r,
               "loops must be the %d ranks of one loopback group, in rank order", n);
    int rc = 0;
    for (int r = 0; r < n; ++r) if ((rc = readout_partial(loops[r], G, ng_indptr, ng_node, ng_w))) return rc;
    const size_t cnt = (size_t)G * loops[0]->T;
    for (int r = 0; r < n; ++r)
        for (int p = 0; p < n; ++p)
            if (p != r) HIPCHK(hipMemcpyAsync(loops[p]->ng_part + cnt * r, loops[r]->ng_part + cnt * r, sizeof(float) * cnt, hipMemcpyDeviceToDevice, loops[r]->stream));
    std::vector<float> first(cnt), other(cnt);
    for (int r = 0; r < n; ++r) {
        if ((rc = readout_combine(loops[r], G, r ? other.data() : first.data()))) return rc;
        if (r && memcmp(first.data(), other.data(), sizeof(float) * cnt) != 0) return gnn_fail(GNN_ERR_STATE, "ranks disagree on the graph readout");
    }
    memcpy(out_graph, first.data(), sizeof(float) * cnt);
    return GNN_OK;
}

// LGNN.update_graph on the owned rows of `dst` (reference GNN/LGNN.py:227-260); nothing is synchronised here
static int relabel_own(gnn_graph *dst, const gnn_graph *base, const gnn_loop *from, int get_state, int get_output)
{
    ARGCHK(dst && base && from, "bad arguments");
    ARGCHK(dst->sh == base->sh, "dst must be derived from base");
    if (!from->ran) return gnn_fail(GNN_ERR_STATE, "the source loop has not run");
    ARGCHK(from->g->sh == base->sh, "the source loop ran on an unrelated graph");
    ARGCHK(!base->halo_world || (dst->halo_world == base->halo_world && dst->halo_block == base->halo_block), "dst is not a boundary-exchange shard like base");
    // edge-based layers put the output on the ARC labels (LGNN.py:253-254), node/graph-based ones on the node labels (:256)
    const bool arc_side = from->edge_mode;
    ARGCHK(!arc_side || from->world == 1, "edge-based LGNN stacks are single-GPU only");
    const int out_nodes = (get_output && !arc_side) ? from->T : 0, out_arcs = (get_output && arc_side) ? from->T : 0;
    const int extra = (get_state ? from->Ds : 0) + out_nodes;
    ARGCHK(dst->NL == base->base_NL + extra, "dst label width %d != %d + %d", dst->NL, base->base_NL, extra);
    HIPCHK(hipSetDevice(dst->device));
    // base labels are the first base_NL columns of base->nodes only when base is not itself derived
    ARGCHK(base->NL == base->base_NL, "base must be the original (underived) graph (LGNN.py:287)");
    const int64_t rows = base->n_rows, off = base->own_off;
    const int64_t tot = rows * dst->NL;
    int rcw = gnn_graph_wait_ready(dst, from->stream);          // the creation-time zero fill of dst's labels is ordered BEFORE the relabelling
    if (rcw) return rcw;
    if (tot)
        hipLaunchKernelGGL(k_relabel, cdiv(tot, 256), 256, 0, from->stream, rows, base->NL, base->nodes + (size_t)off * base->NL, from->Ds,
                           from->state[0] + (size_t)from->own_off * from->Ds, from->state[1] + (size_t)from->own_off * from->Ds, from->kfinal_dev, get_state, from->T,
                           from->out, base->sh->mask, graph_mask_pos(base), out_nodes ? 1 : 0, dst->nodes + (size_t)off * dst->NL, dst->NL);
    if (arc_side) {
        ARGCHK(dst->arc_labels_own && dst->arc_labels_orig_own && base->sh->arc_id, "dst must come from gnn_graph_derive_edge");
        ARGCHK(dst->AL == base->base_AL + out_arcs, "dst arc label width %d != %d + %d", dst->AL, base->base_AL, out_arcs);
        const int64_t E = dst->E;
        if (E && dst->AL) {
            hipLaunchKernelGGL(k_arc_base, cdiv(E * dst->AL, 256), 256, 0, from->stream, E, base->base_AL, base->sh->arc_labels_orig, dst->AL, dst->arc_labels_orig_own);
            if (out_arcs && from->n_edge_masked)
                hipLaunchKernelGGL(k_arc_scatter, cdiv(from->n_edge_masked * from->T, 256), 256, 0, from->stream, from->n_edge_masked, from->T, from->edge_rows,
                                   from->out, base->base_AL, dst->AL, dst->arc_labels_orig_own);
            hipLaunchKernelGGL(k_arc_permute, cdiv(E * dst->AL, 256), 256, 0, from->stream, E, dst->AL, base->sh->arc_id, dst->arc_labels_orig_own, dst->arc_labels_own);
        }
    }
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

// boundary-exchange shards: the rank's boundary rows of the NEW labels go into its block of the index space (then every rank's block is
// exchanged like the state rows of an iteration: all-gather of blocks / device copies in a loopback group)
static int relabel_pack_boundary(gnn_graph *dst, const gnn_loop *from)
{
    if (!dst->halo_world || !dst->halo_count) return GNN_OK;
    const int64_t tot = dst->halo_count * dst->NL;
    float *block = dst->nodes + ((size_t)from->shard_rows + (size_t)dst->halo_rank * dst->halo_block) * dst->NL;
    hipLaunchKernelGGL(k_pack_rows, cdiv(tot, 256), 256, 0, from->stream, dst->halo_count, dst->NL, dst->halo_send, dst->nodes, block);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

extern "C" int gnn_graph_update_labels(gnn_graph *dst, const gnn_graph *base, const gnn_loop *from, int get_state, int get_output)
{
    ARGCHK(from, "bad arguments");
    if (from->comm && from->comm->grp && from->world > 1) return gnn_fail(GNN_ERR_STATE, "loopback group: use gnn_graph_update_labels_group");
    int rc = relabel_own(dst, base, from, get_state, get_output);
    if (rc) return rc;
    if (from->world > 1 && dst->halo_world) {      // boundary-exchange shards: all-gather of the boundary blocks of the new labels, in place
        if ((rc = relabel_pack_boundary(dst, from))) return rc;
        const size_t cnt = (size_t)dst->halo_block * dst->NL;
        float *blocks = dst->nodes + (size_t)from->shard_rows * dst->NL;
        if (cnt) NCCLCHK(g_rccl.AllGather(blocks + cnt * from->rank, blocks, cnt, NCCL_FLOAT32, from->comm->nccl, from->stream));
    } else if (from->world > 1) {      // every rank relabelled its own rows: all-gather whole shards of the new label rows, in place
        const size_t cnt = (size_t)from->shard_rows * dst->NL;
        ARGCHK((int64_t)from->shard_rows * from->world <= dst->nodes_rows, "derived graph too small for the sharded relabelling");
        NCCLCHK(g_rccl.AllGather(dst->nodes + cnt * from->rank, dst->nodes, cnt, NCCL_FLOAT32, from->comm->nccl, from->stream));
    }
    HIPCHK(hipStreamSynchronize(from->stream));
    dst->label_version++;
    return GNN_OK;
}

extern "C" int gnn_graph_update_labels_group(gnn_graph **dsts, gnn_graph *const *bases, gnn_loop *const *froms, int n, int get_state, int get_output)
{
    ARGCHK(dsts && bases && froms && n >= 1, "bad arguments");
    for (int r = 0; r < n; ++r)
        ARGCHK(froms[r] && froms[r]->comm && froms[r]->comm->grp && froms[r]->comm->grp == froms[0]->comm->grp && froms[r]->world == n && froms[r]->rank == r,
               "froms must be the %d ranks of one loopback group, in rank order", n);
    int rc = 0;
    for (int r = 0; r < n; ++r) if ((rc = relabel_own(dsts[r], bases[r], froms[r], get_state, get_output))) return rc;
    if (dsts[0]->halo_world) {          // boundary-exchange shards: every rank's block of boundary label rows into every other rank's copy
        for (int r = 0; r < n; ++r) {
            ARGCHK(dsts[r]->halo_world == n && dsts[r]->halo_block == dsts[0]->halo_block && dsts[r]->NL == dsts[0]->NL, "ranks hold differently shaped boundary-exchange shards");
            if ((rc = relabel_pack_boundary(dsts[r], froms[r]))) return rc;
        }
        const size_t cnt = (size_t)dsts[0]->halo_block * dsts[0]->NL;
        for (int r = 0; r < n && cnt; ++r) {
            const size_t off = ((size_t)froms[r]->shard_rows + (size_t)r * dsts[r]->halo_block) * dsts[r]->NL;
            for (int p = 0; p < n; ++p)
                if (p != r) HIPCHK(hipMemcpyAsync(dsts[p]->nodes + off, dsts[r]->nodes + off, sizeof(float) * cnt, hipMemcpyDeviceToDevice, froms[r]->stream));
        }
        HIPCHK(hipStreamSynchronize(froms[0]->stream));
        for (int r = 0; r < n; ++r) dsts[r]->label_version++;
        return GNN_OK;
    }
    for (int r = 0; r < n; ++r) {
        const size_t cnt = (size_t)froms[r]->shard_rows * dsts[r]->NL;
        ARGCHK((int64_t)froms[r]->shard_rows * n <= dsts[r]->nodes_rows, "derived graph too small for the sharded relabelling");
        for (int p = 0; p < n; ++p)
            if (p != r) HIPCHK(hipMemcpyAsync(dsts[p]->nodes + cnt * r, dsts[r]->nodes + cnt * r, sizeof(float) * cnt, hipMemcpyDeviceToDevice, froms[r]->stream));
    }
    HIPCHK(hipStreamSynchronize(froms[0]->stream));
    for (int r = 0; r < n; ++r) dsts[r]->label_version++;
    return GNN_OK;
}

extern "C" int gnn_lgnn_run(gnn_loop *const *loops, gnn_graph *const *graphs, int n_layers, int get_state, int get_output, float *k_out)
{
    ARGCHK(loops && graphs && n_layers >= 1 && k_out, "bad arguments");
    for (int i = 0; i < n_layers; ++i) {
        ARGCHK(loops[i] && graphs[i] && loops[i]->g == graphs[i], "loops[%d] was not created on graphs[%d]", i, i);
        ARGCHK(i == 0 || graphs[i]->sh == graphs[0]->sh, "graphs[%d] is not derived from graphs[0]", i);
    }
    ARGCHK(graphs[0]->NL == graphs[0]->base_NL && !graphs[0]->arc_labels_own, "graphs[0] must be the original (underived) graph (LGNN.py:287)");
    for (int i = 0; i < n_layers; ++i) {
        int rc = gnn_loop_run(loops[i], 0, &k_out[i]);
        if (rc) return rc;
        if (i + 1 < n_layers && (rc = gnn_graph_update_labels(graphs[i + 1], graphs[0], loops[i], get_state, get_output))) return rc;
    }
    return GNN_OK;
}

extern "C" int gnn_loop_destroy(gnn_loop *l)
{
    if (!l) return GNN_OK;
    (void)hipSetDevice(l->device);
    if (l->comm && l->comm->grp && l->comm->grp->member[l->rank] == l) l->comm->grp->member[l->rank] = nullptr;
    gnn_train_ctx_free(l);
    gnn_train_arena_free(l);
    for (int b = 0; b < 2; ++b) { (void)hipFree(l->state[b]); (void)hipFree(l->tmp[b]); (void)hipFree(l->otmp[b]); }
    (void)hipFree(l->inp); (void)hipFree(l->inv); (void)hipFree(l->state_init); (void)hipFree(l->feats); (void)hipFree(l->out); (void)hipFree(l->flags); (void)hipFree(l->kfinal_dev); (void)hipFree(l->tile_ctr);
    if (l->kfinal_host) (void)hipHostFree(l->kfinal_host);
    (void)hipFree(l->small_xs);
    for (hipEvent_t e : l->ev) (void)hipEventDestroy(e);
    for (int i = 0; i < 2; ++i) if (l->ev_total[i]) (void)hipEventDestroy(l->ev_total[i]);
    if (l->gate_host) (void)hipHostFree(l->gate_host);
    if (l->ng_host) (void)hipHostFree(l->ng_host);
    (void)hipFree(l->edge_dst); (void)hipFree(l->edge_rows); (void)hipFree(l->edge_labels); (void)hipFree(l->edge_inc_ptr); (void)hipFree(l->edge_inc);
    (void)hipFree(l->sl_send); (void)hipFree(l->sl_state); (void)hipFree(l->sl_agg); (void)hipFree(l->sl_recv); (void)hipFree(l->agg_own);
    for (hipEvent_t ev : l->sl_ev) if (ev) (void)hipEventDestroy(ev);
    if (l->sl_done) (void)hipEventDestroy(l->sl_done);
    (void)hipFree(l->ng_ip); (void)hipFree(l->ng_node); (void)hipFree(l->ng_w); (void)hipFree(l->ng_out); (void)hipFree(l->ng_part);
    if (!l->comm && l->stream) (void)hipStreamDestroy(l->stream);
    gnn_comm *comm = l->comm;
    delete l;
    if (comm && --comm->loops == 0 && comm->closed) return gnn_comm_destroy(comm);
    return GNN_OK;
}
