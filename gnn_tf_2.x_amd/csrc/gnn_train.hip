// One training step on the device: training-mode forward through the unrolled state loop, loss, back-propagation through
// all executed iterations (reference GNN/GNN_BaseClass.py:231-247 around GNN/GNN.py:180-199, :251-280).  The optimizer
// step stays on the host (the weights are a few hundred KB): this file returns the loss, the iteration count, the raw
// gradients and the BatchNormalization batch statistics of every executed body.
//
// Keras training semantics (not in the reference repository; restated in oracle/gnn_train_oracle.py):
//   Dropout: y = x * mask / (1 - rate), fresh mask per call (negative rate: AlphaDropout);  BatchNormalization: batch mean / biased batch variance;
//   categorical_crossentropy(from_logits=False): p = out / sum(out), clip to [1e-7, 1 - 1e-7], -sum t log p.
// This path is built from simple per-op kernels (correctness first; training graphs are small batches); float32 with
// atomically accumulated weight gradients, so it is compared with the oracle to a tolerance, not bit for bit.
#include <math.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gnn_common.h"

namespace {

inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

// rows handled by one block of the column reductions / weight-gradient tiles: about 64 blocks along the rows, so that small
// batches (a few hundred rows) still spread over the chip; a multiple of 16 (k_wgrad's row tile), at most 1024
inline int64_t rows_per_block(int64_t n)
{
    const int64_t r = ((n + 63) / 64 + 15) / 16 * 16;
    const int64_t capped = std::min<int64_t>(1024, std::max<int64_t>(32, r));
    // at most 256 row chunks: every chunk leaves a partial result that a second pass adds up in chunk order
    return std::max<int64_t>(capped, ((n + 255) / 256 + 15) / 16 * 16);
}

__device__ __forceinline__ uint64_t mix64(uint64_t x)
{
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

// AlphaDropout (Keras; reference GNN/MLP.py:59-61 with alphadropout=True) is passed as a NEGATIVE rate: dropped units are set to
// alpha' = -selu_scale * selu_alpha and the result is mapped by a x + b so that mean and variance of selu activations are kept:
//   a = ((1 - r)(1 + r alpha'^2))^-1/2,  b = -a alpha' r,  y = a (x keep + alpha' (1 - keep)) + b,  dy/dx = a keep
__device__ __forceinline__ void alpha_dropout_coeffs(float r, float *a, float *b, float *alpha_p)
{
    const float ap = -1.0507009873554805f * 1.6732632423543772f;
    const float aa = 1.0f / sqrtf((1.0f - r) * (1.0f + r * ap * ap));
    *a = aa; *b = -aa * ap * r; *alpha_p = ap;
}

// Dropout forward: keep[i] = injected mask or own RNG; y = x * keep / (1 - rate); keep bytes are stored for the backward pass
__global__ void k_dropout_fwd(int64_t n, const float *x, const uint8_t *mask_in, float rate, uint64_t seed, uint8_t *keep, float *y)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float r = fabsf(rate);
    uint8_t kp;
    if (mask_in) kp = mask_in[i] != 0;
    else kp = ((mix64(seed ^ mix64((uint64_t)i)) >> 40) * (1.0f / 16777216.0f)) >= r;
    keep[i] = kp;
    if (rate < 0.0f) {
        float a, b, ap;
        alpha_dropout_coeffs(r, &a, &b, &ap);
        y[i] = a * (kp ? x[i] : ap) + b;
    } else
        y[i] = kp ? x[i] / (1.0f - rate) : 0.0f;
}

__global__ void k_dropout_bwd(int64_t n, const uint8_t *keep, float rate, float *d)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (rate < 0.0f) {
        float a, b, ap;
        alpha_dropout_coeffs(-rate, &a, &b, &ap);
        d[i] = keep[i] ? d[i] * a : 0.0f;
    } else
        d[i] = keep[i] ? d[i] / (1.0f - rate) : 0.0f;
}

__global__ void k_act_fwd(int64_t n, int F, const float *z, int act, float *a)
{
    if (act == GNN_ACT_SOFTMAX) {
        const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (r >= n) return;
        const float *zr = z + r * F;
        float *ar = a + r * F;
        float m = zr[0];
        for (int j = 1; j < F; ++j) m = zr[j] > m ? zr[j] : m;
        float s = 0.0f;
        for (int j = 0; j < F; ++j) { const float e = gnn_expf(zr[j] - m); ar[j] = e; s = s + e; }
        for (int j = 0; j < F; ++j) ar[j] = __fdiv_rn(ar[j], s);
    } else {
        const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (i < n * F) a[i] = gnn_act(z[i], act);
    }
}

// dz = da * act'(z) (softmax: a * (da - sum da a) per row); in place on d.  Every derivative is a function of the OUTPUT a
// alone (selu: z > 0 <=> a > 0 and scale * alpha * e^z = a + scale * alpha; elu: e^z = a + 1), so z is not kept.
__global__ void k_act_bwd(int64_t n, int F, float *d, const float *a, int act)
{
    if (act == GNN_ACT_SOFTMAX) {
        const int64_t r = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
        if (r >= n) return;
        float s = 0.0f;
        for (int j = 0; j < F; ++j) s += d[r * F + j] * a[r * F + j];
        for (int j = 0; j < F; ++j) d[r * F + j] = a[r * F + j] * (d[r * F + j] - s);
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * F) return;
    const float aa = a[i];
    float g;
    switch (act) {
    case GNN_ACT_RELU: g = aa > 0.0f ? 1.0f : 0.0f; break;
    case GNN_ACT_SELU: g = aa > 0.0f ? 1.0507009873554805f : aa + 1.0507009873554805f * 1.6732632423543772f; break;
    case GNN_ACT_ELU: g = aa > 0.0f ? 1.0f : aa + 1.0f; break;
    case GNN_ACT_TANH: g = 1.0f - aa * aa; break;
    case GNN_ACT_SIGMOID: g = aa * (1.0f - aa); break;
    default: g = 1.0f; break;
    }
    d[i] = d[i] * g;
}

// column reductions over n rows of [n, F] matrices; one block of 256 threads = 32 columns x 8 row lanes
// mode 0: out0[j] += sum x        mode 1: out0[j] += sum (x - aux0[j])^2        mode 2: out0[j] += sum x * y, out1[j] += sum x
__global__ void k_colreduce(int64_t n, int F, const float *x, const float *y, const float *aux0, int mode, float *out0, float *out1,
                            int64_t rows_per_block)
{
    __shared__ float s0[8][33], s1[8][33];
    const int c = threadIdx.x & 31, ry = threadIdx.x >> 5;
    const int j = blockIdx.x * 32 + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float a0 = 0.0f, a1 = 0.0f;
    if (j < F)
        for (int64_t r = r0 + ry; r < r1; r += 8) {
            const float v = x[r * F + j];
            if (mode == 0) a0 += v;
            else if (mode == 1) { const float dv = v - aux0[j]; a0 += dv * dv; }
            else { a0 += v * y[r * F + j]; a1 += v; }
        }
    s0[ry][c] = a0; s1[ry][c] = a1;
    __syncthreads();
    if (ry == 0 && j < F) {
        for (int t = 1; t < 8; ++t) { a0 += s0[t][c]; a1 += s1[t][c]; }
        // partial of this row chunk; k_sum_parts adds the chunks in a fixed order (run-to-run identical sums, no float atomics)
        out0[(size_t)blockIdx.y * F + j] = a0;
        if (mode == 2) out1[(size_t)blockIdx.y * F + j] = a1;
    }
}

// out[t] += part[0][t] + part[1][t] + ... (ascending chunk index), t < count
__global__ void k_sum_parts(int parts, int64_t count, const float *part, float *out)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= count) return;
    float acc = 0.0f;
    for (int z = 0; z < parts; ++z) acc += part[(size_t)z * count + t];
    out[t] += acc;
}

__global__ void k_scale_vec(int n, float *v, float s)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = v[i] * s;
}

__global__ void k_bn_fwd(int64_t n, int F, const float *h, const float *mean, const float *sqsum, float eps, const float *gamma,
                         const float *beta, float *xhat, float *y, float *stats /* [2][F]: batch mean, biased batch var */)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * F) return;
    const int j = (int)(i % F);
    const float mu = mean[j], var = sqsum[j] / (float)n;
    const float xh = (h[i] - mu) / sqrtf(var + eps);
    xhat[i] = xh;
    y[i] = gamma[j] * xh + beta[j];
    if (i < F) { stats[j] = mu; stats[F + j] = var; }
}

// d x = inv / n * (n * dxh - sum dxh - xhat * sum(dxh * xhat)), dxh = d y * gamma; sums: s_dyx = sum dy*xhat, s_dy = sum dy
__global__ void k_bn_bwd(int64_t n, int F, float *d, const float *xhat, const float *gamma, const float *stats, float eps,
                         const float *s_dyx, const float *s_dy)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n * F) return;
    const int j = (int)(i % F);
    const float inv = 1.0f / sqrtf(stats[F + j] + eps), g = gamma[j], m = (float)n;
    d[i] = inv / m * (m * d[i] * g - g * s_dy[j] - xhat[i] * g * s_dyx[j]);
}

// dW[i, j] += sum_r H[r, i] * DZ[r, j] over the rows of this block's chunk; 16 x 16 output tile per block
__global__ void k_wgrad(int64_t n, int n_in, int n_out, const float *H, const float *DZ, float *dW, int64_t rows_per_block)
{
    __shared__ float sh[16][17], sz[16][17];
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;
    const int i0 = blockIdx.x * 16, j0 = blockIdx.y * 16;
    const int64_t r0 = (int64_t)blockIdx.z * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float acc = 0.0f;
    for (int64_t r = r0; r < r1; r += 16) {
        // tile rows r..r+15: thread (ti, tj) loads H[r + ti][i0 + tj] and DZ[r + ti][j0 + tj]
        sh[ti][tj] = (r + ti < r1 && i0 + tj < n_in) ? H[(r + ti) * n_in + i0 + tj] : 0.0f;
        sz[ti][tj] = (r + ti < r1 && j0 + tj < n_out) ? DZ[(r + ti) * n_out + j0 + tj] : 0.0f;
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 16; ++q) acc += sh[q][ti] * sz[q][tj];
        __syncthreads();
    }
    if (i0 + ti < n_in && j0 + tj < n_out) dW[(size_t)blockIdx.z * n_in * n_out + (size_t)(i0 + ti) * n_out + j0 + tj] = acc;   // partial of chunk z
}

__global__ void k_gather_feats(int64_t m, const int32_t *rows, const float *state, int Ds, const float *nodes, int NL, int NLc, float *feats)
{
    const int wf = Ds + NLc;
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * wf) return;
    const int64_t q = t / wf;
    const int c = (int)(t - q * wf);
    const int64_t row = rows[q];
    feats[t] = c < Ds ? state[row * Ds + c] : nodes[row * NL + (c - Ds)];
}

__global__ void k_scatter_rows(int64_t m, const int32_t *rows, const float *d_feats, int wf, int Ds, float *d_state)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * Ds) return;
    const int64_t q = t / Ds;
    const int c = (int)(t - q * Ds);
    d_state[(int64_t)rows[q] * Ds + c] = d_feats[q * wf + c];
}

// d_state[r, c] = d_inp[r, c] + tmp[r, c]   (own-state columns of the concat + transposed aggregation)
__global__ void k_combine(int64_t n, int Ds, const float *d_inp, int in_s, const float *tmp, float *d_state)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * Ds) return;
    const int64_t r = t / Ds;
    const int c = (int)(t - r * Ds);
    d_state[t] = d_inp[r * in_s + c] + tmp[t];
}

// ---------------------------------------------------------------------------------------------------------------------
// Device scratch of the training step: a bump allocator over slabs that stay with the loop from step to step (a step makes
// a few hundred allocations; hipMalloc / hipFree for each of them dominated the step time).  reset() at the next forward.
struct TrainArena {
    struct Slab { char *p; size_t size; };
    std::vector<Slab> slabs;
    size_t cur = 0, off = 0;
    void reset() { cur = 0; off = 0; }
    void *alloc(size_t bytes)
    {
        bytes = (std::max<size_t>(bytes, 1) + 255) & ~(size_t)255;
        for (; cur < slabs.size(); ++cur, off = 0)
            if (off + bytes <= slabs[cur].size) {
                void *r = slabs[cur].p + off;
                off += bytes;
                return r;
            }
        Slab s{nullptr, std::max<size_t>(bytes, (size_t)32 << 20)};
        if (hipMalloc((void **)&s.p, s.size) != hipSuccess) return nullptr;
        slabs.push_back(s);
        cur = slabs.size() - 1;
        off = bytes;
        return s.p;
    }
    ~TrainArena() { for (Slab &s : slabs) (void)hipFree(s.p); }
};

struct Buf {                      // typed front end of the arena
    TrainArena *arena = nullptr;
    template <typename T>
    int get(T **p, size_t count)
    {
        *p = static_cast<T *>(arena->alloc(count * sizeof(T)));
        if (!*p) return gnn_fail(GNN_ERR_HIP, "hipMalloc of %zu bytes failed", count * sizeof(T));
        return GNN_OK;
    }
};

// out0[j] (+ out1[j]) += column reduction of k_colreduce over all n rows, deterministic: partials per row chunk, then k_sum_parts
static int reduce_cols(hipStream_t st, Buf &buf, int64_t n, int F, const float *x, const float *y, const float *aux0, int mode, float *out0, float *out1)
{
    if (n <= 0 || F <= 0) return GNN_OK;
    const int64_t rpb = rows_per_block(n);
    const int parts = (int)cdiv(n, rpb);
    float *p0 = nullptr, *p1 = nullptr;
    int rc = buf.get(&p0, (size_t)parts * F);
    if (!rc && mode == 2) rc = buf.get(&p1, (size_t)parts * F);
    if (rc) return rc;
    hipLaunchKernelGGL(k_colreduce, dim3(cdiv(F, 32), parts), 256, 0, st, n, F, x, y, aux0, mode, p0, p1, rpb);
    hipLaunchKernelGGL(k_sum_parts, cdiv(F, 64), 64, 0, st, parts, (int64_t)F, p0, out0);
    if (mode == 2) hipLaunchKernelGGL(k_sum_parts, cdiv(F, 64), 64, 0, st, parts, (int64_t)F, p1, out1);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

// dW += H^T . DZ, deterministic in the same way
static int weight_grad(hipStream_t st, Buf &buf, int64_t n, int ni, int no, const float *H, const float *DZ, float *dW)
{
    if (n <= 0) return GNN_OK;
    const int64_t rpb = rows_per_block(n);
    const int parts = (int)cdiv(n, rpb);
    float *part = nullptr;
    int rc = buf.get(&part, (size_t)parts * ni * no);
    if (rc) return rc;
    hipLaunchKernelGGL(k_wgrad, dim3(cdiv(ni, 16), cdiv(no, 16), parts), 256, 0, st, n, ni, no, H, DZ, part, rpb);
    hipLaunchKernelGGL(k_sum_parts, cdiv((int64_t)ni * no, 256), 256, 0, st, parts, (int64_t)ni * no, part, dW);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

__global__ void k_transpose(int ni, int no, const float *W, float *WT)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= ni * no) return;
    const int i = t / no, j = t - i * no;
    WT[(size_t)j * ni + i] = W[t];
}

struct NetCache {                 // what one training-mode forward of a Sequential leaves for the backward pass
    std::vector<float *> hin, a;
    std::vector<uint8_t *> keep;  // per dropout index 0..L (nullptr when no dropout there)
    float *xhat = nullptr, *stats = nullptr;
    int64_t n = 0;
};

struct Net {
    const gnn_mlp *m = nullptr;
    std::vector<float *> WT;      // W^T per layer
    float *gamma = nullptr, *beta = nullptr;
    std::vector<float> rate;      // [L + 1] dropout rate in front of Dense l (index L: in front of BatchNormalization)
    float *zero = nullptr;        // zero "bias" of the d h = d z . W^T products (max layer input width)
    float *grads = nullptr;       // flat: dW1, db1, ..., dgamma, dbeta
    std::vector<size_t> g_off;
    size_t g_total = 0;
};

int net_setup(hipStream_t st, Buf &buf, Net &net, const gnn_mlp *m, const float *rates, const float *bn_gamma_beta_host)
{
    net.m = m;
    const int L = m->n_layers;
    net.rate.assign(rates, rates + L + 1);
    net.WT.assign(L, nullptr);
    size_t off = 0;
    for (int l = 0; l < L; ++l) {
        const int ni = m->dims[l], no = m->dims[l + 1];
        int rc = buf.get(&net.WT[l], (size_t)ni * no);
        if (rc) return rc;
        hipLaunchKernelGGL(k_transpose, cdiv((int64_t)ni * no, 256), 256, 0, st, ni, no, m->W[l], net.WT[l]);
        HIPCHK(hipGetLastError());
        net.g_off.push_back(off); off += (size_t)ni * no;
        net.g_off.push_back(off); off += (size_t)no;
    }
    if (m->has_bn) {
        const int F = m->dims.back();
        int rc = buf.get(&net.gamma, (size_t)2 * F);
        if (rc) return rc;
        net.beta = net.gamma + F;
        HIPCHK(hipMemcpyAsync(net.gamma, bn_gamma_beta_host, sizeof(float) * 2 * F, hipMemcpyHostToDevice, st));
        net.g_off.push_back(off); off += F;
        net.g_off.push_back(off); off += F;
    }
    net.g_total = off;
    int maxw = 1;
    for (int l = 0; l <= L; ++l) maxw = std::max(maxw, m->dims[l]);
    int rc = buf.get(&net.grads, off + maxw);         // gradients, then the zero vector: one memset
    if (rc) return rc;
    net.zero = net.grads + off;
    HIPCHK(hipMemsetAsync(net.grads, 0, (off + maxw) * sizeof(float), st));
    return GNN_OK;
}

// training-mode forward of one Sequential on n rows (x: [n, dims[0]]); *y_out: [n, dims.back()]
int net_forward(hipStream_t st, Buf &buf, const Net &net, int64_t n, float *x, const uint8_t *masks, uint64_t seed, NetCache &c, float **y_out)
{
    const gnn_mlp *m = net.m;
    const int L = m->n_layers;
    c.n = n;
    c.hin.assign(L, nullptr); c.a.assign(L, nullptr); c.keep.assign(L + 1, nullptr);
    float *h = x;
    size_t mask_off = 0;
    int rc;
    for (int l = 0; l <= L; ++l) {
        const int width = m->dims[l];
        if (net.rate[l] != 0.0f) {
            float *hd = nullptr;
            if ((rc = buf.get(&hd, (size_t)n * width)) || (rc = buf.get(&c.keep[l], (size_t)n * width))) return rc;
            if (n > 0) {
                hipLaunchKernelGGL(k_dropout_fwd, cdiv(n * width, 256), 256, 0, st, n * width, h, masks ? masks + mask_off : nullptr, net.rate[l],
                                   seed + 0x9E37ull * (uint64_t)(l + 1), c.keep[l], hd);
                HIPCHK(hipGetLastError());
            }
            mask_off += (size_t)n * width;
            h = hd;
        }
        if (l == L) break;
        const int no = m->dims[l + 1];
        c.hin[l] = h;
        if ((rc = buf.get(&c.a[l], (size_t)n * no))) return rc;
        const bool sm = m->acts[l] == GNN_ACT_SOFTMAX;          // softmax needs the whole row: separate pass, in place
        if ((rc = gnn_launch_dense(st, n, width, no, h, width, m->W[l], m->b[l], sm ? GNN_ACT_LINEAR : m->acts[l], c.a[l], no))) return rc;
        if (n > 0 && sm) {
            hipLaunchKernelGGL(k_act_fwd, cdiv(n, 256), 256, 0, st, n, no, c.a[l], m->acts[l], c.a[l]);
            HIPCHK(hipGetLastError());
        }
        h = c.a[l];
    }
    if (m->has_bn) {
        const int F = m->dims.back();
        float *sums = nullptr, *y = nullptr;
        if ((rc = buf.get(&sums, (size_t)2 * F)) || (rc = buf.get(&c.xhat, (size_t)n * F)) || (rc = buf.get(&c.stats, (size_t)2 * F)) ||
            (rc = buf.get(&y, (size_t)n * F))) return rc;
        HIPCHK(hipMemsetAsync(sums, 0, sizeof(float) * 2 * F, st));
        HIPCHK(hipMemsetAsync(c.stats, 0, sizeof(float) * 2 * F, st));
        if (n > 0) {
            if ((rc = reduce_cols(st, buf, n, F, h, nullptr, nullptr, 0, sums, nullptr))) return rc;
            hipLaunchKernelGGL(k_scale_vec, cdiv(F, 64), 64, 0, st, F, sums, 1.0f / (float)n);                       // sums -> batch mean
            if ((rc = reduce_cols(st, buf, n, F, h, nullptr, sums, 1, sums + F, nullptr))) return rc;
            hipLaunchKernelGGL(k_bn_fwd, cdiv(n * F, 256), 256, 0, st, n, F, h, sums, sums + F, m->eps, net.gamma, net.beta, c.xhat, y, c.stats);
            HIPCHK(hipGetLastError());
        }
        h = y;
    }
    *y_out = h;
    return GNN_OK;
}

// back-propagation through one Sequential: d is d loss / d y on entry ([n, dims.back()], overwritten); on return *dx_out is
// d loss / d x ([n, dims[0]]); weight gradients are ADDED into net.grads
int net_backward(hipStream_t st, Buf &buf, Net &net, const NetCache &c, float *d, float **dx_out)
{
    const gnn_mlp *m = net.m;
    const int L = m->n_layers;
    const int64_t n = c.n;
    int rc;
    if (m->has_bn && n > 0) {
        const int F = m->dims.back();
        float *dgamma = net.grads + net.g_off[2 * L], *dbeta = net.grads + net.g_off[2 * L + 1];
        float *loc = nullptr;                      // this call's own column sums (the grads accumulate over iterations)
        if ((rc = buf.get(&loc, (size_t)2 * F))) return rc;
        HIPCHK(hipMemsetAsync(loc, 0, sizeof(float) * 2 * F, st));
        if ((rc = reduce_cols(st, buf, n, F, d, c.xhat, nullptr, 2, loc, loc + F))) return rc;
        hipLaunchKernelGGL(k_bn_bwd, cdiv(n * F, 256), 256, 0, st, n, F, d, c.xhat, net.gamma, c.stats, m->eps, loc, loc + F);
        // dgamma += loc[0:F], dbeta += loc[F:2F]
        hipLaunchKernelGGL(k_sum_parts, cdiv(F, 64), 64, 0, st, 1, (int64_t)F, loc, dgamma);
        hipLaunchKernelGGL(k_sum_parts, cdiv(F, 64), 64, 0, st, 1, (int64_t)F, loc + F, dbeta);
        HIPCHK(hipGetLastError());
    }
    if (net.rate[L] != 0.0f && n > 0) {
        const int F = m->dims.back();
        hipLaunchKernelGGL(k_dropout_bwd, cdiv(n * F, 256), 256, 0, st, n * F, c.keep[L], net.rate[L], d);
        HIPCHK(hipGetLastError());
    }
    for (int l = L - 1; l >= 0; --l) {
        const int ni = m->dims[l], no = m->dims[l + 1];
        float *dprev = nullptr;
        if ((rc = buf.get(&dprev, (size_t)n * ni))) return rc;
        if (n > 0) {
            const bool sm = m->acts[l] == GNN_ACT_SOFTMAX;
            hipLaunchKernelGGL(k_act_bwd, cdiv(sm ? n : n * no, 256), 256, 0, st, n, no, d, c.a[l], m->acts[l]);
            HIPCHK(hipGetLastError());
            if ((rc = weight_grad(st, buf, n, ni, no, c.hin[l], d, net.grads + net.g_off[2 * l]))) return rc;
            if ((rc = reduce_cols(st, buf, n, no, d, nullptr, nullptr, 0, net.grads + net.g_off[2 * l + 1], nullptr))) return rc;
        }
        // d h_in = d z . W^T  (bias-free: the zero vector behind the gradients)
        if ((rc = gnn_launch_dense(st, n, no, ni, d, no, net.WT[l], net.zero, GNN_ACT_LINEAR, dprev, ni))) return rc;
        if (net.rate[l] != 0.0f && n > 0) {
            hipLaunchKernelGGL(k_dropout_bwd, cdiv(n * ni, 256), 256, 0, st, n * ni, c.keep[l], net.rate[l], dprev);
            HIPCHK(hipGetLastError());
        }
        d = dprev;
    }
    *dx_out = d;
    return GNN_OK;
}

// host side of the loss (rows are few): sum_i w_i L(t_i, o_i) and d / d o
void loss_host(int kind, int64_t n, int T, const float *t, const float *o, const float *w, double *loss, std::vector<float> &d_o)
{
    d_o.assign((size_t)n * T, 0.0f);
    double total = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const float *ti = t + i * T, *oi = o + i * T;
        if (kind == 0) {                           // categorical_crossentropy, from_logits=False
            double s = 0.0;
            for (int j = 0; j < T; ++j) s += oi[j];
            std::vector<double> p(T), g(T);
            double li = 0.0, gp = 0.0;
            for (int j = 0; j < T; ++j) {
                p[j] = oi[j] / s;
                const bool in = p[j] >= 1e-7 && p[j] <= 1.0 - 1e-7;
                const double pc = std::min(std::max(p[j], 1e-7), 1.0 - 1e-7);
                li -= ti[j] * log(pc);
                g[j] = in ? -ti[j] / pc : 0.0;
                gp += g[j] * p[j];
            }
            for (int j = 0; j < T; ++j) d_o[i * T + j] = (float)(w[i] * (g[j] - gp) / s);
            total += w[i] * li;
        } else if (kind == 2) {                    // categorical_crossentropy, from_logits=True: softmax inside the loss, no clipping
            double mx = oi[0], s = 0.0, st = 0.0, li = 0.0;
            for (int j = 1; j < T; ++j) mx = std::max(mx, (double)oi[j]);
            for (int j = 0; j < T; ++j) { s += exp(oi[j] - mx); st += ti[j]; }
            for (int j = 0; j < T; ++j) {
                const double logp = (oi[j] - mx) - log(s);
                li -= ti[j] * logp;
                d_o[i * T + j] = (float)(w[i] * (exp(logp) * st - ti[j]));
            }
            total += w[i] * li;
        } else {                                   // mean_squared_error
            double li = 0.0;
            for (int j = 0; j < T; ++j) { const double e = (double)oi[j] - ti[j]; li += e * e; d_o[i * T + j] = (float)(w[i] * 2.0 * e / T); }
            total += w[i] * li / T;
        }
    }
    *loss = total;
}

// d_nodes[r, c] += d_inp[r, c_nodes + c] + via[r, c]   (direct label columns of the concat + transposed aggregated_nodes)
__global__ void k_nodes_grad(int64_t n, int NL, const float *d_inp, int in_s, int c_nodes, const float *via, float *d_nodes)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * NL) return;
    const int64_t r = t / NL;
    const int c = (int)(t - r * NL);
    d_nodes[t] += d_inp[r * in_s + c_nodes + c] + via[t];
}

// d_nodes[rows[q], c] += d_feats[q, Ds + c]   (label columns of net_output's input; rows are unique)
__global__ void k_scatter_label_grad(int64_t m, const int32_t *rows, const float *d_feats, int wf, int Ds, int NL, float *d_nodes)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * NL) return;
    const int64_t q = t / NL;
    const int c = (int)(t - q * NL);
    d_nodes[(int64_t)rows[q] * NL + c] += d_feats[q * wf + Ds + c];
}

// GNNedgeBased backward: row q of d_feats = d [F[dst(e)] | F[src(e)] | arc label], e = rows[q]; F = [state | labels?].
// Both endpoints receive their half (several arcs share a node: atomics); the arc-label columns are data.
__global__ void k_scatter_edge_grad(int64_t m, const int32_t *rows, const int32_t *entry_dst, const int32_t *adj_src, const float *d_feats,
                                    int we, int wn, int Ds, int NL, float *d_state, float *d_nodes)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * 2 * wn) return;
    const int64_t q = t / (2 * wn);
    int c = (int)(t - q * 2 * wn);
    const int64_t e = rows[q];
    const float v = d_feats[q * we + c];
    const int64_t node = c < wn ? entry_dst[e] : adj_src[e];
    if (c >= wn) c -= wn;
    if (c < Ds) atomicAdd(d_state + node * Ds + c, v);
    else if (d_nodes) atomicAdd(d_nodes + node * NL + (c - Ds), v);
}

// acc[r, c] += d_inp[r, col0 + c]   (the loop-invariant aggregated arc labels receive gradient from every body)
__global__ void k_add_cols(int64_t n, int width, const float *d_inp, int in_s, int col0, float *acc)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * width) return;
    const int64_t r = t / width;
    const int c = (int)(t - r * width);
    acc[t] += d_inp[r * in_s + col0 + c];
}

// d arc labels, ORIGINAL arc order.  (a) label columns of the per-arc readout rows: row m <-> arc position rows[m];
// (b) ArcNode^T . arc labels: entry q of destination dst carries arc arc_id[q] with weight arc_w[q].  Targets are unique.
__global__ void k_arc_grad_readout(int64_t m, int AL, const int32_t *rows, const float *d_feats, int we, int col0, float *d_arcs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= m * AL) return;
    const int64_t q = t / AL;
    const int c = (int)(t - q * AL);
    d_arcs[(int64_t)rows[q] * AL + c] += d_feats[q * we + col0 + c];
}

__global__ void k_arc_grad_agg(int64_t e, int AL, const int32_t *entry_dst, const int32_t *arc_id, const float *arc_w, const float *d_agg, float *d_arcs)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= e * AL) return;
    const int64_t q = t / AL;
    const int c = (int)(t - q * AL);
    d_arcs[(int64_t)arc_id[q] * AL + c] += arc_w[q] * d_agg[(int64_t)entry_dst[q] * AL + c];
}

__global__ void k_axpy1(int64_t n, const float *x, float *y)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < n) y[t] += x[t];
}

// What gnn_loop_train_forward leaves for gnn_loop_train_backward (owned by the loop; replaced by the next forward)
struct TrainCtx {
    Buf buf;
    Net ns, no_;
    std::vector<NetCache> caches;
    NetCache co;
    int32_t *d_sip = nullptr, *d_sdst = nullptr;
    float *d_sw = nullptr;
    float *state = nullptr, *out_nodes = nullptr;
    int k = 0;
};

}   // namespace

void gnn_train_ctx_free(gnn_loop *l)
{
    if (l && l->train_ctx) {
        delete static_cast<TrainCtx *>(l->train_ctx);
        l->train_ctx = nullptr;
    }
}

void gnn_train_arena_free(gnn_loop *l)
{
    if (l && l->train_arena) {
        delete static_cast<TrainArena *>(l->train_arena);
        l->train_arena = nullptr;
    }
}

extern "C" int gnn_loss_grad(int loss_kind, int64_t n_rows, int n_out, const float *targets, const float *out, const float *sample_weights,
                             double *loss, float *d_out)
{
    ARGCHK((n_rows == 0 || (targets && out && sample_weights)) && loss && n_out > 0 && n_rows >= 0, "bad arguments");
    ARGCHK(loss_kind >= 0 && loss_kind <= 2, "loss_kind: 0 categorical_crossentropy, 1 mean_squared_error, 2 categorical_crossentropy(from_logits=True)");
    std::vector<float> d;
    loss_host(loss_kind, n_rows, n_out, targets, out, sample_weights, loss, d);
    if (d_out && n_rows) memcpy(d_out, d.data(), sizeof(float) * d.size());
    return GNN_OK;
}

extern "C" int gnn_loop_train_forward(gnn_loop *l, const int32_t *src_indptr, const int32_t *src_dst, const float *src_w,
                                      const float *dropout_state, const float *dropout_output, const uint8_t *masks_state,
                                      const uint8_t *masks_output, uint64_t seed, const float *bn_state, const float *bn_output,
                                      float *k_out, float *out_nodes_host)
{
    ARGCHK(l && dropout_state && dropout_output && k_out, "bad arguments");
    ARGCHK(l->world == 1, "training is single-GPU");
    ARGCHK(l->edge_mode == l->edge_expected, "edge-based net_output: call gnn_loop_set_edge_readout first");
    ARGCHK(!l->st->has_bn || bn_state, "net_state ends with BatchNormalization: gamma|beta required");
    ARGCHK(!l->ou->has_bn || bn_output, "net_output ends with BatchNormalization: gamma|beta required");
    if (!l->have_state0 && l->D) return gnn_fail(GNN_ERR_STATE, "state_vect_dim > 0: call gnn_loop_set_state0 first");
    gnn_graph *g = l->g;
    const int64_t N = g->n_rows, M = l->edge_mode ? l->n_edge_masked : g->n_masked, E = g->E;
    const int Ds = l->Ds, NLc = l->NLc, in_s = l->in_s, T = l->T, wf = l->ou->dims[0];
    HIPCHK(hipSetDevice(l->device));
    hipStream_t st = l->stream;
    gnn_train_ctx_free(l);
    if (!l->train_arena) l->train_arena = new TrainArena();
    static_cast<TrainArena *>(l->train_arena)->reset();
    TrainCtx *cx = new TrainCtx();
    l->train_ctx = cx;
    cx->buf.arena = static_cast<TrainArena *>(l->train_arena);
    Buf &buf = cx->buf;
    Net &ns = cx->ns, &no_ = cx->no_;
    int rc;
    if ((rc = net_setup(st, buf, ns, l->st, dropout_state, bn_state)) || (rc = net_setup(st, buf, no_, l->ou, dropout_output, bn_output))) return rc;
    // Adjacency by source for the transposed aggregation of the backward pass: the caller's arrays, or (NULL) the graph's
    // own copy, built once from its CSR by destination (a stable counting sort by source keeps destinations ascending)
    if (src_indptr) {
        if ((rc = buf.get(&cx->d_sip, (size_t)N + 1)) || (rc = buf.get(&cx->d_sdst, (size_t)E)) || (rc = buf.get(&cx->d_sw, (size_t)E))) return rc;
        ARGCHK(src_indptr[0] == 0 && src_indptr[N] == E && (E == 0 || (src_dst && src_w)), "bad by-source CSR");
        HIPCHK(hipMemcpy(cx->d_sip, src_indptr, sizeof(int32_t) * (N + 1), hipMemcpyHostToDevice));
        if (E) { HIPCHK(hipMemcpy(cx->d_sdst, src_dst, sizeof(int32_t) * E, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(cx->d_sw, src_w, sizeof(float) * E, hipMemcpyHostToDevice)); }
    } else {
        gnn_graph_shared *sh = g->sh;
        if (!sh->src_indptr) {
            std::vector<int32_t> ip((size_t)N + 1), src((size_t)E), sip((size_t)N + 1, 0), sdst((size_t)E);
            std::vector<float> w((size_t)E), sw((size_t)E);
            HIPCHK(hipMemcpy(ip.data(), sh->indptr, sizeof(int32_t) * (N + 1), hipMemcpyDeviceToHost));
            if (E) { HIPCHK(hipMemcpy(src.data(), sh->adj_src, sizeof(int32_t) * E, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(w.data(), sh->adj_w, sizeof(float) * E, hipMemcpyDeviceToHost)); }
            for (int64_t e = 0; e < E; ++e) ++sip[(size_t)src[e] + 1];
            for (int64_t i = 0; i < N; ++i) sip[i + 1] += sip[i];
            std::vector<int32_t> fill(sip.begin(), sip.end() - 1);
            for (int64_t d = 0; d < N; ++d)
                for (int32_t e = ip[d]; e < ip[d + 1]; ++e) { const int32_t q = fill[src[e]]++; sdst[q] = (int32_t)d; sw[q] = w[e]; }
            if (hipMalloc((void **)&sh->src_indptr, sizeof(int32_t) * (N + 1)) != hipSuccess || hipMalloc((void **)&sh->src_dst, sizeof(int32_t) * std::max<int64_t>(E, 1)) != hipSuccess ||
                hipMalloc((void **)&sh->src_w, sizeof(float) * std::max<int64_t>(E, 1)) != hipSuccess)
                return gnn_fail(GNN_ERR_HIP, "hipMalloc of the by-source adjacency failed");
            HIPCHK(hipMemcpy(sh->src_indptr, sip.data(), sizeof(int32_t) * (N + 1), hipMemcpyHostToDevice));
            if (E) { HIPCHK(hipMemcpy(sh->src_dst, sdst.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(sh->src_w, sw.data(), sizeof(float) * E, hipMemcpyHostToDevice)); }
        }
        cx->d_sip = sh->src_indptr; cx->d_sdst = sh->src_dst; cx->d_sw = sh->src_w;
    }

    // template of the concat with the loop-invariant columns filled in (GNN.py:259, :263)
    float *tmpl = nullptr;
    if ((rc = buf.get(&tmpl, (size_t)N * in_s))) return rc;
    HIPCHK(hipMemsetAsync(tmpl, 0, sizeof(float) * (size_t)N * in_s, st));
    const int c_nodes = Ds, c_aggs = Ds + NLc, c_aggn = c_aggs + Ds, c_agga = c_aggn + NLc;
    if ((rc = gnn_launch_spmm(st, N, g->sh->indptr, nullptr, g->sh->arc_w, gnn_graph_arc_labels(g), g->AL, g->AL, tmpl + c_agga, in_s, nullptr, 1))) return rc;
    if (l->D) {
        if ((rc = gnn_launch_spmm(st, N, g->sh->indptr, g->sh->adj_src, g->sh->adj_w, g->nodes, g->NL, g->NL, tmpl + c_aggn, in_s, nullptr, 1))) return rc;
        if ((rc = gnn_launch_copy_cols(st, N, g->NL, g->nodes, g->NL, tmpl + c_nodes, in_s, nullptr, 1))) return rc;
    }
    // state, condition flags
    float *state = nullptr, *state_old = nullptr;
    int *flag = nullptr;
    if ((rc = buf.get(&state, (size_t)N * Ds)) || (rc = buf.get(&state_old, (size_t)N * Ds)) || (rc = buf.get(&flag, (size_t)GNN_FLAG_WORDS))) return rc;
    if (N) HIPCHK(hipMemcpyAsync(state, l->D ? l->state_init : g->nodes, sizeof(float) * (size_t)N * Ds, hipMemcpyDeviceToDevice, st));
    std::vector<int> hflag(GNN_FLAG_WORDS);
    auto not_converged = [&](const float *s, const float *so, bool *go) -> int {
        HIPCHK(hipMemsetAsync(flag, 0, sizeof(int) * GNN_FLAG_WORDS, st));
        int r = gnn_launch_check(st, N, Ds, s, so, l->thr, flag);
        if (r) return r;
        HIPCHK(hipMemcpyAsync(hflag.data(), flag, sizeof(int) * GNN_FLAG_WORDS, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        int any = 0;
        for (int i = 0; i < GNN_FLAG_WORDS; i += GNN_FLAG_STRIDE) any |= hflag[i];
        *go = any != 0;
        return GNN_OK;
    };
    // masks of one iteration of net_state: sum over the dropout positions of N * width bytes
    size_t mask_iter_bytes = 0;
    for (int i = 0; i <= l->st->n_layers; ++i) if (dropout_state[i] != 0.0f) mask_iter_bytes += (size_t)N * l->st->dims[i];
    size_t mask_out_bytes = 0;
    for (int i = 0; i <= l->ou->n_layers; ++i) if (dropout_output[i] != 0.0f) mask_out_bytes += (size_t)M * l->ou->dims[i];
    uint8_t *d_masks_s = nullptr, *d_masks_o = nullptr;
    if (masks_state && mask_iter_bytes) {
        if ((rc = buf.get(&d_masks_s, mask_iter_bytes * (size_t)l->max_iter))) return rc;
        HIPCHK(hipMemcpy(d_masks_s, masks_state, mask_iter_bytes * (size_t)l->max_iter, hipMemcpyHostToDevice));
    }
    if (masks_output && mask_out_bytes) {
        if ((rc = buf.get(&d_masks_o, mask_out_bytes))) return rc;
        HIPCHK(hipMemcpy(d_masks_o, masks_output, mask_out_bytes, hipMemcpyHostToDevice));
    }

    // ---- while condition: state <- net_state(concat), training mode (GNN.py:271 with training=True) ----------------------
    bool go = false;
    if ((rc = not_converged(state, nullptr, &go))) return rc;
    int k = 0;
    while (go && k < l->max_iter) {
        float *inp = nullptr, *y = nullptr;
        if ((rc = buf.get(&inp, (size_t)N * in_s))) return rc;
        HIPCHK(hipMemcpyAsync(inp, tmpl, sizeof(float) * (size_t)N * in_s, hipMemcpyDeviceToDevice, st));
        if ((rc = gnn_launch_copy_cols(st, N, Ds, state, Ds, inp, in_s, nullptr, 1))) return rc;
        if ((rc = gnn_launch_spmm(st, N, g->sh->indptr, g->sh->adj_src, g->sh->adj_w, state, Ds, Ds, inp + c_aggs, in_s, nullptr, 1))) return rc;
        cx->caches.emplace_back();
        if ((rc = net_forward(st, buf, ns, N, inp, d_masks_s ? d_masks_s + mask_iter_bytes * (size_t)k : nullptr, seed + 7919ull * (uint64_t)(k + 1),
                              cx->caches.back(), &y))) return rc;
        std::swap(state, state_old);
        if (N) HIPCHK(hipMemcpyAsync(state, y, sizeof(float) * (size_t)N * Ds, hipMemcpyDeviceToDevice, st));
        ++k;
        if ((rc = not_converged(state, state_old, &go))) return rc;
    }
    // ---- net_output on the masked rows --------------------------------------------------------------------------------------
    float *feats = nullptr;
    if ((rc = buf.get(&feats, (size_t)M * wf))) return rc;
    if (l->edge_mode) {
        if ((rc = gnn_launch_feats_edge(st, l, state, feats))) return rc;
    } else if (M) {
        hipLaunchKernelGGL(k_gather_feats, cdiv(M * wf, 256), 256, 0, st, M, g->sh->masked_rows, state, Ds, g->nodes, g->NL, NLc, feats);
        HIPCHK(hipGetLastError());
    }
    if ((rc = net_forward(st, buf, no_, M, feats, d_masks_o, seed + 104729ull, cx->co, &cx->out_nodes))) return rc;
    cx->state = state;
    cx->k = k;
    // publish the training-mode state / outputs as the loop's result: gnn_loop_get_state / get_output / readout and
    // gnn_graph_update_labels (LGNN stacking) read them exactly like an inference run's
    const int zero = 0;
    if (N) HIPCHK(hipMemcpyAsync(l->state[0], state, sizeof(float) * (size_t)N * Ds, hipMemcpyDeviceToDevice, st));
    if (M) HIPCHK(hipMemcpyAsync(l->out, cx->out_nodes, sizeof(float) * (size_t)M * T, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(l->kfinal_dev, &zero, sizeof(int), hipMemcpyHostToDevice, st));
    if (out_nodes_host && M) HIPCHK(hipMemcpyAsync(out_nodes_host, cx->out_nodes, sizeof(float) * (size_t)M * T, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    l->kfinal = 0;
    *l->kfinal_host = 0;
    l->ran = true;
    *k_out = (float)k;
    return GNN_OK;
}

extern "C" int gnn_loop_train_backward(gnn_loop *l, const float *d_out_nodes, const float *d_state_extra, float *grads_state,
                                       float *grads_output, float *bn_batch_state, float *bn_batch_output, float *d_nodes_host,
                                       float *d_arcs_host)
{
    ARGCHK(l && grads_state && grads_output, "bad arguments");
    TrainCtx *cx = static_cast<TrainCtx *>(l->train_ctx);
    if (!cx) return gnn_fail(GNN_ERR_STATE, "gnn_loop_train_forward has not been called");
    gnn_graph *g = l->g;
    const int64_t N = g->n_rows, M = l->edge_mode ? l->n_edge_masked : g->n_masked;
    const int Ds = l->Ds, NLc = l->NLc, in_s = l->in_s, T = l->T, wf = l->ou->dims[0], NL = g->NL, k = cx->k;
    ARGCHK(M == 0 || d_out_nodes, "d_out_nodes is NULL");
    HIPCHK(hipSetDevice(l->device));
    hipStream_t st = l->stream;
    Buf &buf = cx->buf;
    Net &ns = cx->ns, &no_ = cx->no_;
    const int c_nodes = Ds, c_aggs = Ds + NLc, c_aggn = c_aggs + Ds;
    int rc;
    float *d_out = nullptr, *d_feats = nullptr, *d_state = nullptr, *tmp = nullptr, *d_nodes = nullptr, *via = nullptr;
    if ((rc = buf.get(&d_out, (size_t)M * T)) || (rc = buf.get(&d_state, (size_t)N * Ds)) || (rc = buf.get(&tmp, (size_t)N * Ds))) return rc;
    if (M) HIPCHK(hipMemcpyAsync(d_out, d_out_nodes, sizeof(float) * (size_t)M * T, hipMemcpyHostToDevice, st));
    if ((rc = net_backward(st, buf, no_, cx->co, d_out, &d_feats))) return rc;
    if (d_state_extra) { if (N) HIPCHK(hipMemcpyAsync(d_state, d_state_extra, sizeof(float) * (size_t)N * Ds, hipMemcpyHostToDevice, st)); }
    else HIPCHK(hipMemsetAsync(d_state, 0, sizeof(float) * (size_t)N * Ds, st));
    const bool want_nodes = d_nodes_host != nullptr;
    const bool want_arcs = d_arcs_host != nullptr && g->AL > 0;
    const int AL = g->AL, c_agga = c_aggn + NLc;
    float *d_arcs = nullptr, *d_aa = nullptr;
    if (want_arcs) {
        ARGCHK(l->edge_mode && g->sh->arc_id, "d_arc_labels: edge-based loop on a graph with gnn_graph_set_arc_order required");
        if ((rc = buf.get(&d_arcs, (size_t)g->E * AL)) || (rc = buf.get(&d_aa, (size_t)N * AL))) return rc;
        HIPCHK(hipMemsetAsync(d_arcs, 0, sizeof(float) * std::max<size_t>(1, (size_t)g->E * AL), st));
        HIPCHK(hipMemsetAsync(d_aa, 0, sizeof(float) * std::max<size_t>(1, (size_t)N * AL), st));
    }
    if (want_nodes && l->D) {
        if ((rc = buf.get(&d_nodes, (size_t)N * NL)) || (rc = buf.get(&via, (size_t)N * NL))) return rc;
        HIPCHK(hipMemsetAsync(d_nodes, 0, sizeof(float) * (size_t)N * NL, st));
    }
    if (l->edge_mode) {
        if (M) {
            const int wn = Ds + NLc;
            hipLaunchKernelGGL(k_scatter_edge_grad, cdiv(M * 2 * wn, 256), 256, 0, st, M, l->edge_rows, l->edge_dst, g->sh->adj_src, d_feats, wf, wn,
                               Ds, NL, d_state, d_nodes);
            HIPCHK(hipGetLastError());
        }
    } else if (M) {
        if (d_state_extra) {        // extra + scattered rows: scatter into a zero buffer, then add
            HIPCHK(hipMemsetAsync(tmp, 0, sizeof(float) * (size_t)N * Ds, st));
            hipLaunchKernelGGL(k_scatter_rows, cdiv(M * Ds, 256), 256, 0, st, M, g->sh->masked_rows, d_feats, wf, Ds, tmp);
            hipLaunchKernelGGL(k_axpy1, cdiv(N * Ds, 256), 256, 0, st, N * Ds, tmp, d_state);
        } else
            hipLaunchKernelGGL(k_scatter_rows, cdiv(M * Ds, 256), 256, 0, st, M, g->sh->masked_rows, d_feats, wf, Ds, d_state);
        HIPCHK(hipGetLastError());
    }
    if (d_nodes && !l->edge_mode) {
        if (M) {
            hipLaunchKernelGGL(k_scatter_label_grad, cdiv(M * NL, 256), 256, 0, st, M, g->sh->masked_rows, d_feats, wf, Ds, NL, d_nodes);
            HIPCHK(hipGetLastError());
        }
    }
    for (int it = k - 1; it >= 0; --it) {
        float *d_inp = nullptr, *dy = nullptr;
        if ((rc = buf.get(&dy, (size_t)N * Ds))) return rc;                       // net_backward overwrites its input
        if (N) HIPCHK(hipMemcpyAsync(dy, d_state, sizeof(float) * (size_t)N * Ds, hipMemcpyDeviceToDevice, st));
        if ((rc = net_backward(st, buf, ns, cx->caches[it], dy, &d_inp))) return rc;
        // aggregated_states = Adjacency^T . state  =>  d state[src] = d inp[src, :Ds] + sum over arcs (src -> dst) of w * d agg[dst]
        if ((rc = gnn_launch_spmm(st, N, cx->d_sip, cx->d_sdst, cx->d_sw, d_inp + c_aggs, Ds, in_s, tmp, Ds, nullptr, 1))) return rc;
        if (N) {
            hipLaunchKernelGGL(k_combine, cdiv(N * Ds, 256), 256, 0, st, N, Ds, d_inp, in_s, tmp, d_state);
            HIPCHK(hipGetLastError());
        }
        if (want_arcs && N) {
            hipLaunchKernelGGL(k_add_cols, cdiv(N * AL, 256), 256, 0, st, N, AL, d_inp, in_s, c_agga, d_aa);
            HIPCHK(hipGetLastError());
        }
        if (want_nodes && l->D && N) {    // labels enter each body directly and through aggregated_nodes (GNN.py:228, :263)
            if ((rc = gnn_launch_spmm(st, N, cx->d_sip, cx->d_sdst, cx->d_sw, d_inp + c_aggn, NL, in_s, via, NL, nullptr, 1))) return rc;
            hipLaunchKernelGGL(k_nodes_grad, cdiv(N * NL, 256), 256, 0, st, N, NL, d_inp, in_s, c_nodes, via, d_nodes);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipMemcpyAsync(grads_state, ns.grads, sizeof(float) * ns.g_total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(grads_output, no_.grads, sizeof(float) * no_.g_total, hipMemcpyDeviceToHost, st));
    if (bn_batch_state && l->st->has_bn)
        for (int it = 0; it < k; ++it)
            HIPCHK(hipMemcpyAsync(bn_batch_state + (size_t)it * 2 * Ds, cx->caches[it].stats, sizeof(float) * 2 * Ds, hipMemcpyDeviceToHost, st));
    if (bn_batch_output && l->ou->has_bn && M) HIPCHK(hipMemcpyAsync(bn_batch_output, cx->co.stats, sizeof(float) * 2 * T, hipMemcpyDeviceToHost, st));
    if (want_arcs && g->E) {
        if (M) hipLaunchKernelGGL(k_arc_grad_readout, cdiv(M * AL, 256), 256, 0, st, M, AL, l->edge_rows, d_feats, wf, 2 * (Ds + NLc), d_arcs);
        hipLaunchKernelGGL(k_arc_grad_agg, cdiv(g->E * AL, 256), 256, 0, st, g->E, AL, l->edge_dst, g->sh->arc_id, g->sh->arc_w, d_aa, d_arcs);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d_arcs_host, d_arcs, sizeof(float) * (size_t)g->E * AL, hipMemcpyDeviceToHost, st));
    }
    if (want_nodes && N)        // D == 0: state_0 = nodes (GNN.py:265), so the gradient of the initial state IS the label gradient
        HIPCHK(hipMemcpyAsync(d_nodes_host, l->D ? d_nodes : d_state, sizeof(float) * (size_t)N * NL, hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    gnn_train_ctx_free(l);          // one backward per forward: the weight-gradient accumulators are spent
    return GNN_OK;
}

extern "C" int gnn_loop_train_step(gnn_loop *l, const int32_t *src_indptr, const int32_t *src_dst, const float *src_w,
                                   const float *targets, const float *sample_weights, int64_t n_targets, int loss_kind,
                                   int n_graphs, const int32_t *ng_indptr, const int32_t *ng_node, const float *ng_w,
                                   const float *dropout_state, const float *dropout_output, const uint8_t *masks_state,
                                   const uint8_t *masks_output, uint64_t seed, const float *bn_state, const float *bn_output,
                                   float *loss_out, float *k_out, float *grads_state, float *grads_output,
                                   float *bn_batch_state, float *bn_batch_output)
{
    ARGCHK(l && targets && sample_weights && loss_out && k_out && grads_state && grads_output, "bad arguments");
    ARGCHK(loss_kind >= 0 && loss_kind <= 2, "loss_kind: 0 categorical_crossentropy, 1 mean_squared_error, 2 categorical_crossentropy(from_logits=True)");
    const int64_t M = l->edge_mode ? l->n_edge_masked : l->g->n_masked;
    const int T = l->T;
    ARGCHK(!(l->edge_mode && n_graphs > 0), "an edge-based loop has no graph readout");
    ARGCHK(n_targets == (n_graphs > 0 ? n_graphs : M), "%lld target rows but %lld outputs", (long long)n_targets, (long long)(n_graphs > 0 ? n_graphs : M));
    ARGCHK(n_graphs <= 0 || (ng_indptr && ng_node && ng_w), "NodeGraph^T CSR required for a graph-based step");
    std::vector<float> h_out((size_t)M * T), h_dnodes((size_t)M * T, 0.0f), d_o;
    int rc = gnn_loop_train_forward(l, src_indptr, src_dst, src_w, dropout_state, dropout_output, masks_state, masks_output, seed,
                                    bn_state, bn_output, k_out, h_out.data());
    if (rc) return rc;
    double loss = 0.0;
    if (n_graphs > 0) {                            // GNNgraphBased: out = NodeGraph^T . out_nodes (GNN.py:331-332)
        std::vector<float> og((size_t)n_graphs * T, 0.0f);
        for (int gi = 0; gi < n_graphs; ++gi)
            for (int e = ng_indptr[gi]; e < ng_indptr[gi + 1]; ++e)
                for (int t = 0; t < T; ++t) og[(size_t)gi * T + t] += ng_w[e] * h_out[(size_t)ng_node[e] * T + t];
        loss_host(loss_kind, n_graphs, T, targets, og.data(), sample_weights, &loss, d_o);
        for (int gi = 0; gi < n_graphs; ++gi)
            for (int e = ng_indptr[gi]; e < ng_indptr[gi + 1]; ++e)
                for (int t = 0; t < T; ++t) h_dnodes[(size_t)ng_node[e] * T + t] += ng_w[e] * d_o[(size_t)gi * T + t];
    } else {
        loss_host(loss_kind, M, T, targets, h_out.data(), sample_weights, &loss, h_dnodes);
    }
    rc = gnn_loop_train_backward(l, h_dnodes.data(), nullptr, grads_state, grads_output, bn_batch_state, bn_batch_output, nullptr, nullptr);
    if (rc) return rc;
    *loss_out = (float)loss;
    return GNN_OK;
}
