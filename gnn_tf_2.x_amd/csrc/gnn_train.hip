// One training step on the device: training-mode forward through the unrolled state loop, loss, back-propagation through
// all executed iterations, optimizer update (reference GNN/GNN_BaseClass.py:231-247 around GNN/GNN.py:180-199, :251-280).
// The step returns the loss, the iteration count, the raw gradients and the BatchNormalization batch statistics of every
// executed body; with gnn_loop_arm_optimizer the Adam / SGD update and the moving statistics are applied on the device too
// (weights, optimizer slots and gradients never leave HBM) and the host waits for the device once per TRAIN_CHUNK bodies
// of the forward pass and once at the end of the step.
//
// Keras training semantics (not in the reference repository; restated in oracle/gnn_train_oracle.py):
//   Dropout: y = x * mask / (1 - rate), fresh mask per call (negative rate: AlphaDropout);  BatchNormalization: batch mean / biased batch variance;
//   categorical_crossentropy(from_logits=False): p = out / sum(out), clip to [1e-7, 1 - 1e-7], -sum t log p.
// Launch structure (the graphs of a training batch are small: the step is bound by the NUMBER of launches, so every body is few,
// fused kernels): forward body = concat + gather + Dropout + gate (k_train_input), one k_dense_fwd per layer, two
// BatchNormalization kernels; backward body = two BatchNormalization kernels, one k_layer_bwd per layer (weight / bias gradient
// tiles beside d h_in, with the Dropout / activation derivative as its epilogue), one k_state_grad_sum.  float32; every
// reduction over rows leaves per-chunk partials that are added in a fixed order (no float atomics, run-to-run identical), so
// results are compared with the float64 oracle to a tolerance, not bit for bit.
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gnn_common.h"
#include "gnn_fused_kernel.h"     // layer_from_lds / f32x16: the f32-MFMA K-step pipeline of the exact fused path, reused by the wide dense products

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

// act'(z) as a function of the OUTPUT a alone (selu: z > 0 <=> a > 0 and scale * alpha * e^z = a + scale * alpha; elu: e^z = a + 1),
// so z is not kept.  Softmax is not elementwise: k_act_bwd.
__device__ __forceinline__ float act_grad(float aa, int act)
{
    switch (act) {
    case GNN_ACT_RELU: return aa > 0.0f ? 1.0f : 0.0f;
    case GNN_ACT_SELU: return aa > 0.0f ? 1.0507009873554805f : aa + 1.0507009873554805f * 1.6732632423543772f;
    case GNN_ACT_ELU: return aa > 0.0f ? 1.0f : aa + 1.0f;
    case GNN_ACT_TANH: return 1.0f - aa * aa;
    case GNN_ACT_SIGMOID: return aa * (1.0f - aa);
    default: return 1.0f;
    }
}

__device__ __forceinline__ float dropout_grad(float d, uint8_t keep, float rate)
{
    if (rate < 0.0f) {
        float a, b, ap;
        alpha_dropout_coeffs(-rate, &a, &b, &ap);
        return keep ? d * a : 0.0f;
    }
    return keep ? d / (1.0f - rate) : 0.0f;
}

// dz = da * act'(z) (softmax: a * (da - sum da a) per row); in place on d
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
    d[i] = d[i] * act_grad(a[i], act);
}

// Row chunks.  Every reduction over the rows of a matrix (BatchNormalization statistics, bias / weight / gamma / beta gradients)
// is done per chunk of rows_per_block(n) rows; a chunk leaves a partial result and the partials are added in a fixed order
// (k_sum_parts, or by the consumer itself): run-to-run identical sums without float atomics.
// Thread layout of the column reductions: 256 threads = CW columns x (256 / CW) row lanes, CW = 2^cw_shift >= min(F, 32), so that
// narrow matrices (F = 14, 16) still use the whole block; rows are read four at a time (independent loads in flight).
inline int column_shift(int F) { int s = 0; while ((1 << s) < F && s < 5) ++s; return s; }

// partial sums of x * y and x over the rows of chunk blockIdx.y: out0 / out1 [chunk * ostride + j]
__global__ void __launch_bounds__(256) k_colreduce2(int64_t n, int F, int cw_shift, const float *__restrict__ x, const float *__restrict__ y, float *out0,
                                                    float *out1, int64_t ostride, int64_t rows_per_block)
{
    __shared__ float s0[256], s1[256];
    const int CW = 1 << cw_shift, RL = 256 >> cw_shift;
    const int c = threadIdx.x & (CW - 1), ry = threadIdx.x >> cw_shift;
    const int j = blockIdx.x * CW + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float a0 = 0.0f, a1 = 0.0f;
    if (j < F) {
        int64_t r = r0 + ry;
        for (; r + 3 * RL < r1; r += 4 * RL) {
            float xv[4], yv[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { xv[q] = x[(r + q * RL) * F + j]; yv[q] = y[(r + q * RL) * F + j]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) { a0 += xv[q] * yv[q]; a1 += xv[q]; }
        }
        for (; r < r1; r += RL) { const float v = x[r * F + j]; a0 += v * y[r * F + j]; a1 += v; }
    }
    s0[threadIdx.x] = a0; s1[threadIdx.x] = a1;
    __syncthreads();
    if (ry == 0 && j < F) {
        for (int t = 1; t < RL; ++t) { a0 += s0[t * CW + c]; a1 += s1[t * CW + c]; }
        out0[(size_t)blockIdx.y * ostride + j] = a0;
        out1[(size_t)blockIdx.y * ostride + j] = a1;
    }
}

// out[t] += part[0][t] + part[1][t] + ... for the 64 columns of block `bid`: four lanes per column take every fourth chunk, their
// sums are added in lane order
__device__ __forceinline__ void sum_parts_block(int bid, int parts, int64_t count, const float *__restrict__ part, float *out, float *sp /* [256] */)
{
    const int c = threadIdx.x & 63, zl = threadIdx.x >> 6;
    const int64_t t = (int64_t)bid * 64 + c;
    float acc = 0.0f;
    if (t < count) {
#pragma unroll 8
        for (int z = zl; z < parts; z += 4) acc += part[(size_t)z * count + t];
    }
    sp[threadIdx.x] = acc;
    __syncthreads();
    if (zl == 0 && t < count) out[t] += ((acc + sp[64 + c]) + sp[128 + c]) + sp[192 + c];
}

__global__ void __launch_bounds__(256) k_sum_parts(int parts, int64_t count, const float *part, float *out)
{
    __shared__ float sp[256];
    sum_parts_block(blockIdx.x, parts, count, part, out, sp);
}

// out[j] = sum over the chunks z (ascending) of base[z * stride + j], j < count: a rank's own share of sums that the sharded backward
// pass exchanges (BatchNormalization: sum d y xhat | sum d y)
__global__ void __launch_bounds__(256) k_sum_strided(int parts, int64_t stride, const float *__restrict__ base, int count, float *out)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    float acc = 0.0f;
    for (int z = 0; z < parts; ++z) acc += base[(size_t)z * stride + j];
    out[j] = acc;
}

// BatchNormalization, training mode, forward statistics of one row chunk: part[chunk][j] = chunk mean, part[chunk][F + j] =
// sum over the chunk of (x - chunk mean)^2 (two passes over the chunk's rows)
__global__ void __launch_bounds__(256) k_bn_stats(int64_t n, int F, int cw_shift, const float *__restrict__ h, float *part, int64_t rows_per_block)
{
    __shared__ float s0[256];
    __shared__ float mu[32];
    const int CW = 1 << cw_shift, RL = 256 >> cw_shift;
    const int c = threadIdx.x & (CW - 1), ry = threadIdx.x >> cw_shift;
    const int j = blockIdx.x * CW + c;
    const int64_t r0 = (int64_t)blockIdx.y * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float a0 = 0.0f;
    if (j < F) {
        int64_t r = r0 + ry;
        for (; r + 3 * RL < r1; r += 4 * RL) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = h[(r + q * RL) * F + j];
#pragma unroll
            for (int q = 0; q < 4; ++q) a0 += v[q];
        }
        for (; r < r1; r += RL) a0 += h[r * F + j];
    }
    s0[threadIdx.x] = a0;
    __syncthreads();
    if (ry == 0) {
        for (int t = 1; t < RL; ++t) a0 += s0[t * CW + c];
        mu[c] = a0 / (float)(r1 - r0);
    }
    __syncthreads();
    const float m = mu[c];
    a0 = 0.0f;
    if (j < F) {
        int64_t r = r0 + ry;
        for (; r + 3 * RL < r1; r += 4 * RL) {
            float v[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = h[(r + q * RL) * F + j] - m;
#pragma unroll
            for (int q = 0; q < 4; ++q) a0 += v[q] * v[q];
        }
        for (; r < r1; r += RL) { const float dv = h[r * F + j] - m; a0 += dv * dv; }
    }
    __syncthreads();
    s0[threadIdx.x] = a0;
    __syncthreads();
    if (ry == 0 && j < F) {
        for (int t = 1; t < RL; ++t) a0 += s0[t * CW + c];
        part[(size_t)blockIdx.y * 2 * F + j] = m;
        part[(size_t)blockIdx.y * 2 * F + F + j] = a0;
    }
}

// (count, mean, M2) of two disjoint sets of rows -> of their union (exact in real arithmetic; the order of the calls is fixed)
__device__ __forceinline__ void stats_merge(float &cnt, float &mean, float &m2, float cb, float mb, float qb)
{
    if (cb == 0.0f) return;
    const float delta = mb - mean, tot = cnt + cb;
    mean = mean + delta * (cb / tot);
    m2 = m2 + qb + delta * delta * (cnt * cb / tot);
    cnt = tot;
}

// batch mean / biased batch variance from the chunk statistics, then xhat = (h - mean) / sqrt(var + eps), y = gamma xhat + beta.
// Every block combines the chunks itself (256 / CW lanes per column take every (256 / CW)-th chunk, the lanes are merged in order);
// block 0 leaves [mean | var] in stats for the backward pass and the moving statistics.  Dynamic LDS: 2 F floats.
__global__ void __launch_bounds__(256) k_bn_apply(int64_t n, int F, int cw_shift, const float *__restrict__ h, const float *__restrict__ part, int parts,
                                                  int64_t rows_per_block, float eps, const float *gamma, const float *beta, float *xhat, float *y, float *stats)
{
    extern __shared__ float bsh[];
    __shared__ float sc[3][256];
    float *sm = bsh, *sinv = bsh + F;
    const int CW = 1 << cw_shift, ZL = 256 >> cw_shift;
    const int c = threadIdx.x & (CW - 1), zl = threadIdx.x >> cw_shift;
    for (int jb = 0; jb < F; jb += CW) {
        const int j = jb + c;
        float cnt = 0.0f, mean = 0.0f, m2 = 0.0f;
        if (j < F) {
#pragma unroll 4
            for (int z = zl; z < parts; z += ZL) {
                const int64_t r0 = (int64_t)z * rows_per_block;
                const float nz = (float)((r0 + rows_per_block < n ? r0 + rows_per_block : n) - r0);
                stats_merge(cnt, mean, m2, nz, part[(size_t)z * 2 * F + j], part[(size_t)z * 2 * F + F + j]);
            }
        }
        sc[0][threadIdx.x] = cnt; sc[1][threadIdx.x] = mean; sc[2][threadIdx.x] = m2;
        __syncthreads();
        if (zl == 0 && j < F) {
            for (int t = 1; t < ZL; ++t) stats_merge(cnt, mean, m2, sc[0][t * CW + c], sc[1][t * CW + c], sc[2][t * CW + c]);
            const float var = m2 / (float)n;
            sm[j] = mean;
            sinv[j] = 1.0f / sqrtf(var + eps);
            if (blockIdx.x == 0) { stats[j] = mean; stats[F + j] = var; }
        }
        __syncthreads();
    }
    const int64_t total = n * F, step = (int64_t)gridDim.x * blockDim.x;
    const bool small = total < ((int64_t)1 << 31);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int j = small ? (int)((unsigned)i % (unsigned)F) : (int)(i % F);
        const float xh = (h[i] - sm[j]) * sinv[j];
        xhat[i] = xh;
        y[i] = gamma[j] * xh + beta[j];
    }
}

// (Round 3 tried BatchNormalization of a small batch as ONE single-block launch per direction - statistics + apply, column sums + apply,
//  matrices staged in LDS, tree-reduced column sums - to save two launches per call: SLOWER than the two multi-block kernels at MUTAG size,
//  0.85 against 0.76 ms per 10-body step and 3.0 against 2.6 ms per 50-body step: one workgroup's latency chain against a few microseconds
//  of launch.  Removed.)
// ---- BatchNormalization statistics over the rows of ALL ranks (sharded training forward) ----------------------------------------------
// k_bn_local: the rank's chunk statistics merged in chunk order into ONE triple per feature, tri = [count | mean | M2] (3 F floats);
// the triples of all ranks are all-gathered (3 F floats per rank and call - the review's "2 H floats" plus the count) and
// k_bn_apply_ext merges them in RANK order - every rank the same numbers - before it normalises its own rows.
__global__ void __launch_bounds__(256) k_bn_local(int64_t n, int F, const float *__restrict__ part, int parts, int64_t rows_per_block, float *tri)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F) return;
    float cnt = 0.0f, mean = 0.0f, m2 = 0.0f;
    for (int z = 0; z < parts; ++z) {
        const int64_t r0 = (int64_t)z * rows_per_block;
        const float nz = (float)((r0 + rows_per_block < n ? r0 + rows_per_block : n) - r0);
        stats_merge(cnt, mean, m2, nz, part[(size_t)z * 2 * F + j], part[(size_t)z * 2 * F + F + j]);
    }
    tri[j] = cnt; tri[F + j] = mean; tri[2 * F + j] = m2;
}

__global__ void __launch_bounds__(256) k_bn_apply_ext(int64_t n, int F, const float *__restrict__ h, const float *__restrict__ tri_all, int world, float eps,
                                                      const float *gamma, const float *beta, float *xhat, float *y, float *stats)
{
    extern __shared__ float bsh[];
    float *sm = bsh, *sinv = bsh + F;
    for (int j = threadIdx.x; j < F; j += blockDim.x) {
        float cnt = 0.0f, mean = 0.0f, m2 = 0.0f;
        for (int p = 0; p < world; ++p) stats_merge(cnt, mean, m2, tri_all[(size_t)p * 3 * F + j], tri_all[(size_t)p * 3 * F + F + j], tri_all[(size_t)p * 3 * F + 2 * F + j]);
        const float var = cnt > 0.0f ? m2 / cnt : 0.0f;
        sm[j] = mean;
        sinv[j] = 1.0f / sqrtf(var + eps);
        if (blockIdx.x == 0) { stats[j] = mean; stats[F + j] = var; }
    }
    __syncthreads();
    const int64_t total = n * F, step = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int j = (int)(i % F);
        const float xh = (h[i] - sm[j]) * sinv[j];
        xhat[i] = xh;
        y[i] = gamma[j] * xh + beta[j];
    }
}

// d x = inv / n * (n * dxh - sum dxh - xhat * sum(dxh * xhat)), dxh = d y * gamma, with sum d y * xhat / sum d y added up from the
// chunk partials p_dyx / p_dy [chunk * pstride + j] (the same numbers k_sum_parts adds into the gamma / beta gradients);
// then, fused, the derivative of the layer's activation: d <- d x * act'(a) (act < 0: none).  Dynamic LDS: 2 F floats.
__global__ void __launch_bounds__(256) k_bn_bwd_apply(int64_t n, int F, int cw_shift, float *d, const float *__restrict__ xhat, const float *gamma,
                                                      const float *stats, float eps, const float *__restrict__ p_dyx, const float *__restrict__ p_dy,
                                                      int64_t pstride, int parts, const float *__restrict__ a, int act, int64_t n_stat = 0)
{
    // n_stat: rows the batch statistics were taken over (sharded backward: the rows of ALL ranks, the partial sums are then one pair per rank); 0: n
    extern __shared__ float bsh[];
    __shared__ float sc[2][256];
    float *s_dyx = bsh, *s_dy = bsh + F;
    const int CW = 1 << cw_shift, ZL = 256 >> cw_shift;
    const int c = threadIdx.x & (CW - 1), zl = threadIdx.x >> cw_shift;
    for (int jb = 0; jb < F; jb += CW) {
        const int j = jb + c;
        float a0 = 0.0f, a1 = 0.0f;
        if (j < F) {
#pragma unroll 4
            for (int z = zl; z < parts; z += ZL) { a0 += p_dyx[(size_t)z * pstride + j]; a1 += p_dy[(size_t)z * pstride + j]; }
        }
        sc[0][threadIdx.x] = a0; sc[1][threadIdx.x] = a1;
        __syncthreads();
        if (zl == 0 && j < F) {
            for (int t = 1; t < ZL; ++t) { a0 += sc[0][t * CW + c]; a1 += sc[1][t * CW + c]; }
            s_dyx[j] = a0; s_dy[j] = a1;
        }
        __syncthreads();
    }
    const int64_t total = n * F, step = (int64_t)gridDim.x * blockDim.x;
    const float m = (float)(n_stat > 0 ? n_stat : n);
    const bool small = total < ((int64_t)1 << 31);
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += step) {
        const int j = small ? (int)((unsigned)i % (unsigned)F) : (int)(i % F);
        const float inv = 1.0f / sqrtf(stats[F + j] + eps), g = gamma[j];
        float v = inv / m * (m * d[i] * g - g * s_dy[j] - xhat[i] * g * s_dyx[j]);
        if (act >= 0) v = v * act_grad(a[i], act);
        d[i] = v;
    }
}

// Weight and bias gradient of one Dense layer over one row chunk: part[chunk * pstride + i * n_out + j] = sum_r H'[r, i] DZ[r, j]
// with H' = [H | 1] (row i = n_in is the bias gradient: dW and db are adjacent in the gradient vector).  32 x 32 outputs per
// block, 2 x 2 per thread (one 8-byte LDS read of each operand per four products), 64 rows staged per step, the next step's
// rows fetched into registers while this step's products run.  lds: 2 x 64 x 34 floats.
#define GNN_WG_TILE 32
#define GNN_WG_LD 34
__device__ __forceinline__ void wgrad_block(int bx, int by, int bz, int64_t n, int n_in, int n_out, const float *__restrict__ H,
                                            const float *__restrict__ DZ, float *part, int64_t pstride, int64_t rows_per_block, float *lds)
{
    float *sh = lds, *sz = lds + 64 * GNN_WG_LD;
    const int ti = threadIdx.x >> 4, tj = threadIdx.x & 15;            // outputs (i0 + 2 ti + {0, 1}, j0 + 2 tj + {0, 1})
    const int lr = threadIdx.x >> 5, lc = threadIdx.x & 31;            // loader: rows lr + 8 q, column lc
    const int i0 = bx * GNN_WG_TILE, j0 = by * GNN_WG_TILE;
    const int64_t r0 = (int64_t)bz * rows_per_block, r1 = r0 + rows_per_block < n ? r0 + rows_per_block : n;
    float a00 = 0.0f, a01 = 0.0f, a10 = 0.0f, a11 = 0.0f;
    float hv[8], zv[8];
    auto fetch = [&](int64_t r) {                  // this thread's share of the 64-row step at r, into registers
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int64_t row = r + lr + 8 * q;
            const bool in = row < r1;
            hv[q] = !in ? 0.0f : (i0 + lc < n_in ? H[row * n_in + i0 + lc] : (i0 + lc == n_in ? 1.0f : 0.0f));
            zv[q] = (in && j0 + lc < n_out) ? DZ[row * n_out + j0 + lc] : 0.0f;
        }
    };
    if (r0 < r1) fetch(r0);
    for (int64_t r = r0; r < r1; r += 64) {
#pragma unroll
        for (int q = 0; q < 8; ++q) { sh[(lr + 8 * q) * GNN_WG_LD + lc] = hv[q]; sz[(lr + 8 * q) * GNN_WG_LD + lc] = zv[q]; }
        __syncthreads();
        if (r + 64 < r1) fetch(r + 64);            // the next step's loads fly during this step's products
#pragma unroll 16
        for (int q = 0; q < 64; ++q) {
            const float2 hq = *reinterpret_cast<const float2 *>(sh + q * GNN_WG_LD + 2 * ti);
            const float2 zq = *reinterpret_cast<const float2 *>(sz + q * GNN_WG_LD + 2 * tj);
            a00 = __builtin_fmaf(hq.x, zq.x, a00); a01 = __builtin_fmaf(hq.x, zq.y, a01);
            a10 = __builtin_fmaf(hq.y, zq.x, a10); a11 = __builtin_fmaf(hq.y, zq.y, a11);
        }
        __syncthreads();
    }
    float *out = part + (size_t)bz * pstride;
    const int i = i0 + 2 * ti, j = j0 + 2 * tj;
    if (i <= n_in) {
        if (j < n_out) out[(size_t)i * n_out + j] = a00;
        if (j + 1 < n_out) out[(size_t)i * n_out + j + 1] = a01;
    }
    if (i + 1 <= n_in) {
        if (j < n_out) out[(size_t)(i + 1) * n_out + j] = a10;
        if (j + 1 < n_out) out[(size_t)(i + 1) * n_out + j + 1] = a11;
    }
}

// Dense products of the training step, Y[r, j] = sum_k X[r, k] M[k, j] on R rows per block: 256 threads = CW output columns x KG
// slices of the k range (CW = 2^cshift >= min(columns, 64)); every thread runs the fmaf chain of its slice (one to a few iterations
// even for narrow layers: the loop over k is a chain of L2 round trips), the KG partial sums of an output are added in slice order
// through LDS.  xs: R rows of X, padded to a multiple of 4 (zeros); ps: [KG][R][CW] partials.  fin(r, j, value) stores an output.
template <int R, class Fin>
__device__ __forceinline__ void dense_rows(int n_k, int n_k_pad, int n_cols, int cshift, const float *__restrict__ M, const float *xs, float *ps, Fin fin)
{
    const int CW = 1 << cshift, KG = 256 >> cshift;
    const int c = threadIdx.x & (CW - 1), kg = threadIdx.x >> cshift;
    const int slice = ((n_k + KG - 1) / KG + 3) & ~3;
    const int k0 = kg * slice, k1 = k0 + slice < n_k ? k0 + slice : n_k;
    for (int jb = 0; jb < n_cols; jb += CW) {
        const int j = jb + c;
        float acc[R];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = 0.0f;
        if (j < n_cols) {
            int k = k0;
            for (; k + 4 <= k1; k += 4) {
                const float w0 = M[(size_t)(k + 0) * n_cols + j], w1 = M[(size_t)(k + 1) * n_cols + j];
                const float w2 = M[(size_t)(k + 2) * n_cols + j], w3 = M[(size_t)(k + 3) * n_cols + j];
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const float4 x = *reinterpret_cast<const float4 *>(&xs[r * n_k_pad + k]);
                    acc[r] = __builtin_fmaf(x.x, w0, acc[r]);
                    acc[r] = __builtin_fmaf(x.y, w1, acc[r]);
                    acc[r] = __builtin_fmaf(x.z, w2, acc[r]);
                    acc[r] = __builtin_fmaf(x.w, w3, acc[r]);
                }
            }
            for (; k < k1; ++k) {
                const float wk = M[(size_t)k * n_cols + j];
#pragma unroll
                for (int r = 0; r < R; ++r) acc[r] = __builtin_fmaf(xs[r * n_k_pad + k], wk, acc[r]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) ps[(kg * R + r) * CW + c] = acc[r];
        __syncthreads();
        if (j < n_cols)
            for (int r = kg; r < R; r += KG) {
                float v = ps[r * CW + c];
                for (int g = 1; g < KG; ++g) v += ps[(g * R + r) * CW + c];
                fin(r, j, v);
            }
        __syncthreads();
    }
}

inline int dense_cshift(int cols) { int s = 2; while ((1 << s) < cols && s < 6) ++s; return s; }     // CW = 4 .. 64
inline size_t dense_lds_bytes(int R, int k_pad) { return sizeof(float) * ((size_t)R * k_pad + (size_t)256 * R); }

// d h_in = d z . W^T, then (fused) the way back through what produced h_in: Dropout (keep != NULL) and the previous layer's
// activation (act >= 0: d <- d * act'(a_prev))
template <int R>
__device__ __forceinline__ void dense_bwd_block(int64_t bid, int64_t n, int n_out, int n_out_pad, int n_in, int cshift, const float *__restrict__ DZ,
                                                const float *__restrict__ WT, const uint8_t *__restrict__ keep, float rate,
                                                const float *__restrict__ a_prev, int act, float *__restrict__ dprev, float *xs)
{
    const int64_t i0 = bid * R;
    for (int t = threadIdx.x; t < R * n_out_pad; t += blockDim.x) {
        const int r = t / n_out_pad, k = t - r * n_out_pad;
        xs[t] = (k < n_out && i0 + r < n) ? DZ[(i0 + r) * n_out + k] : 0.0f;
    }
    __syncthreads();
    dense_rows<R>(n_out, n_out_pad, n_in, cshift, WT, xs, xs + R * n_out_pad, [&](int r, int j, float v) {
        if (i0 + r >= n) return;
        const int64_t o = (i0 + r) * n_in + j;
        if (keep) v = dropout_grad(v, keep[o], rate);
        if (act >= 0) v = v * act_grad(a_prev[o], act);
        dprev[o] = v;
    });
}

// a = act(h . W + b), training-mode forward of one Dense layer (softmax is applied by the caller)
template <int R>
__global__ void __launch_bounds__(256) k_dense_fwd(int64_t n, int n_in, int n_in_pad, int n_out, int cshift, const float *__restrict__ X,
                                                   const float *__restrict__ W, const float *__restrict__ b, int act, float *__restrict__ Y)
{
    extern __shared__ __attribute__((aligned(16))) float xs[];
    const int64_t i0 = (int64_t)blockIdx.x * R;
    for (int t = threadIdx.x; t < R * n_in_pad; t += blockDim.x) {
        const int r = t / n_in_pad, k = t - r * n_in_pad;
        xs[t] = (k < n_in && i0 + r < n) ? X[(i0 + r) * n_in + k] : 0.0f;
    }
    __syncthreads();
    dense_rows<R>(n_in, n_in_pad, n_out, cshift, W, xs, xs + R * n_in_pad, [&](int r, int j, float v) {
        if (i0 + r >= n) return;
        v = v + b[j];
        if (act != GNN_ACT_SOFTMAX) v = gnn_act(v, act);
        Y[(i0 + r) * n_out + j] = v;
    });
}

// ALL Dense layers of a Sequential on a row tile in one launch (few rows: a MUTAG-sized step is bound by the number of launches): the
// tile's activations go from layer to layer through LDS, every layer's output is also written out (the backward pass reads it); the
// arithmetic per layer is that of k_dense_fwd (same dense_rows chains: identical bits).  No Dropout between the layers, softmax only as
// the last activation (applied by the caller).  LDS: 2 R maxpad + 256 R floats.
struct MlpFwd {
    int64_t n;
    int L, maxpad;
    int dims[GNN_FUSED_MAXL + 2], pad[GNN_FUSED_MAXL + 2], cshift[GNN_FUSED_MAXL + 1], act[GNN_FUSED_MAXL + 1];
    const float *W[GNN_FUSED_MAXL + 1], *b[GNN_FUSED_MAXL + 1];
    const float *X;
    float *Y[GNN_FUSED_MAXL + 1];
    // build != 0 (net_state of a loop body, no Dropout in front of the first layer): the input rows are not read from X but BUILT here -
    // the concat of k_train_input (GNN.py:223-239: own state | template columns | aggregated neighbour states, the fmaf chain over the
    // arcs in stored order) - written to X_out for the backward pass, and the body's gate (GNN.py:202-220) is evaluated per row
    int build, Ds, c_aggs;
    const float *tmpl, *state, *own, *own_prev;
    const int32_t *indptr, *adj_src;
    const float *adj_w;
    float *X_out;
    float thr;
    int *flag;
};
template <int R>
__global__ void __launch_bounds__(256) k_mlp_fwd(const MlpFwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *in = lds, *out = lds + (size_t)R * p.maxpad, *ps = lds + (size_t)2 * R * p.maxpad;
    const int64_t i0 = (int64_t)blockIdx.x * R;
    if (p.build) {
        const int in_s = p.dims[0], Ds = p.Ds, c_aggs = p.c_aggs;
        int moved = 0;
        for (int t = threadIdx.x; t < R * p.pad[0]; t += blockDim.x) {
            const int r = t / p.pad[0], c = t - r * p.pad[0];
            const int64_t row = i0 + r;
            float v = 0.0f;
            if (c < in_s && row < p.n) {
                if (c < Ds) v = p.own[row * Ds + c];
                else if (c >= c_aggs && c < c_aggs + Ds) {
                    const int cc = c - c_aggs;
                    const int32_t e1 = p.indptr[row + 1];
                    for (int32_t e = p.indptr[row]; e < e1; e += 4) {          // four arcs per step: their loads are in flight together
                        float w[4], x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool in_ = e + u < e1;
                            w[u] = in_ ? p.adj_w[e + u] : 0.0f;
                            x[u] = in_ ? p.state[(int64_t)p.adj_src[e + u] * Ds + cc] : 0.0f;
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u) if (e + u < e1) v = __builtin_fmaf(w[u], x[u], v);
                    }
                } else
                    v = p.tmpl[row * in_s + c];
                p.X_out[row * in_s + c] = v;
                if (c == 0) {                              // the while-condition of this body for the row (k_train_input's chain)
                    float dist = 0.0f, nrm = 0.0f;
                    for (int q = 0; q < Ds; ++q) {
                        const float o = p.own_prev ? p.own_prev[row * Ds + q] : 1.0f;
                        const float df = p.own[row * Ds + q] - o;
                        dist = dist + df * df;
                        nrm = nrm + o * o;
                    }
                    moved |= sqrtf(dist) > p.thr * sqrtf(nrm) ? 1 : 0;
                }
            }
            in[t] = v;
        }
        if (__any(moved) && (threadIdx.x & 63) == 0) gnn_flag_raise(p.flag);
    } else {
        for (int t = threadIdx.x; t < R * p.pad[0]; t += blockDim.x) {
            const int r = t / p.pad[0], k = t - r * p.pad[0];
            in[t] = (k < p.dims[0] && i0 + r < p.n) ? p.X[(i0 + r) * p.dims[0] + k] : 0.0f;
        }
    }
    for (int l = 0; l < p.L; ++l) {
        const int no = p.dims[l + 1], npad = p.pad[l + 1], act = p.act[l];
        for (int t = threadIdx.x; t < R * npad; t += blockDim.x) out[t] = 0.0f;          // (the padding columns of the next input)
        __syncthreads();
        const float *bl = p.b[l];
        float *Yl = p.Y[l];
        dense_rows<R>(p.dims[l], p.pad[l], no, p.cshift[l], p.W[l], in, ps, [&](int r, int j, float v) {
            v = v + bl[j];
            if (act != GNN_ACT_SOFTMAX) v = gnn_act(v, act);
            out[r * npad + j] = v;
            if (i0 + r < p.n) Yl[(i0 + r) * no + j] = v;
        });
        float *t_ = in; in = out; out = t_;          // (dense_rows ends with a barrier)
    }
}

// One Dense layer of the backward pass in one launch: the blocks of the weight / bias gradient (first wg_blocks ids: the heavier
// ones) and the blocks of d h_in run side by side; both read d z, neither reads the other's result.
struct LayerBwd {
    int64_t n, rows_per_block, pstride;
    int n_in, n_out, n_out_pad, act, wg_bx, wg_by, wg_blocks, cshift;
    float rate;
    const float *H, *DZ, *WT, *a_prev;
    const uint8_t *keep;
    float *part, *dprev;
};

template <int R>
__global__ void __launch_bounds__(256) k_layer_bwd(const LayerBwd p)
{
    extern __shared__ __attribute__((aligned(16))) float lds[];
    if ((int)blockIdx.x < p.wg_blocks) {
        const int id = blockIdx.x, bx = id % p.wg_bx, by = (id / p.wg_bx) % p.wg_by, bz = id / (p.wg_bx * p.wg_by);
        wgrad_block(bx, by, bz, p.n, p.n_in, p.n_out, p.H, p.DZ, p.part, p.pstride, p.rows_per_block, lds);
    } else
        dense_bwd_block<R>((int64_t)blockIdx.x - p.wg_blocks, p.n, p.n_out, p.n_out_pad, p.n_in, p.cshift, p.DZ, p.WT, p.keep, p.rate, p.a_prev, p.act, p.dprev, lds);
}

// ---------------------------------------------------------------------------------------------------------------------
// Wide layers (round 3): the three dense products of a Dense layer on the matrix cores (v_mfma_f32_32x32x2_f32: f32 in, f32
// accumulate, every product exact, the k-ordered fmaf chain of the per-op kernels - so results do not depend on the grid).
//   forward      a      = act(h . W + b)                        k_gemm_f32, weights as the A operand, 32 rows of h per wave as B
//   backward     d h_in = d z . W^T (x Dropout / act' epilogue)   k_gemm_f32 on W^T
//                [dW; db] = [h | 1]^T . d z                      k_wgrad_f32: rows are the K dimension; per-chunk partials, added in
//                                                               chunk order afterwards like every other reduction of the step
// BASELINE configs[2] shape (1 M rows, 135 -> 128 -> 128 -> 64): k_dense_fwd 1.08 ms and k_layer_bwd 2.02 ms per layer on the FP32
// ALUs before (profiles/r03_train_c3.txt).
// ---------------------------------------------------------------------------------------------------------------------
constexpr int TG_WAVES = 8;
constexpr int64_t GNN_TRAIN_MFMA_MIN_ROWS = 4096;      // below that a step is launch-bound (MUTAG batches: 570 rows) and the per-op kernels are as fast

// packed A operand of layer_from_lds for output columns [col0, col0 + 32 NO) of M [K, n_cols]: wp[(kk 64 + lane) NO + j] =
// M[2 kk + (lane >> 5)][col0 + 32 j + (lane & 31)], zero outside the matrix (K-steps up to kk_total: the pipeline's look-ahead)
__global__ void k_pack_exact(int K, int n_cols, int col0, int NO, int kk_total, const float *__restrict__ M, float *__restrict__ wp)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= kk_total * 64 * NO) return;
    const int j = t % NO, lane = (t / NO) & 63, kk = t / (64 * NO);
    const int k = 2 * kk + (lane >> 5), c = col0 + 32 * j + (lane & 31);
    wp[t] = (k < K && c < n_cols) ? M[(size_t)k * n_cols + c] : 0.0f;
}

struct GemmArgs {
    int64_t n;
    int K, KP, kk, n_cols, col0, act, mode, spread;       // mode 0: forward (bias + activation); 1: backward (Dropout / act' of the producer)
    float rate;
    const float *X, *wp, *bias, *a_prev;
    const uint8_t *keep;
    float *Y;
};

// Y[r, col0 .. col0 + 32 NO) = epilogue(X[r, :] . M[:, col0 ..]) for all rows; X dense [n, K], Y dense [n, n_cols].
// One wave = 32 rows: rows staged in LDS (odd row stride: conflict-free column reads), K-steps through layer_from_lds.  All pieces of a
// tile (up to 18 x 16 B per lane) are requested before the first is written to LDS: one round trip per tile, covered by the SIMD's other
// wave.  (Measured, profiles/r03_train_c3.txt: a staging loop with a load per iteration - 17 dependent round trips - 0.89 ms per
// 1 M x 135 x 128 product; the NEXT tile's rows held in registers across the K-steps: 256 VGPRs + 163 spilled, 1.06 ms.)
constexpr int TG_MAXQ = 18;                        // 16-byte pieces per lane of a 32 x 144 tile

template <int ACT, int NO>
__device__ __forceinline__ void gemm_store_fwd(const GemmArgs &p, f32x16 (&acc)[NO], int64_t row, int half, bool vec)
{
    using namespace gnn_fused_dev;
#pragma unroll
    for (int jt = 0; jt < NO; ++jt)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int f0 = p.col0 + 32 * jt + 8 * q + 4 * half;
            const int64_t o = row * p.n_cols + f0;
            float v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = f0 + u < p.n_cols ? act_t<ACT>(acc[jt][4 * q + u] + p.bias[f0 + u]) : 0.0f;
            if (vec && f0 + 4 <= p.n_cols) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(p.Y) + o) = v4f{v[0], v[1], v[2], v[3]};
            else
                for (int u = 0; u < 4; ++u) if (f0 + u < p.n_cols) gptr_w(p.Y)[o + u] = v[u];
        }
}

template <int NO>
__global__ void __launch_bounds__(64 * TG_WAVES, 2) k_gemm_f32(const GemmArgs p)
{
    using namespace gnn_fused_dev;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = p.K, KP = p.KP;
    float *X = lds + (size_t)wave * 32 * KP;
    for (int t = lane; t < 32 * (KP - K); t += 64) X[(t / (KP - K)) * KP + K + t % (KP - K)] = 0.0f;      // columns >= K: zero, once
    const int64_t n_tiles = (p.n + 31) / 32;
    const float inv_k = 1.0f / (float)K;
    const int half = lane >> 5, node = lane & 31;
    const int64_t stride = (int64_t)gridDim.x * TG_WAVES;
    const bool vec = (p.n_cols & 3) == 0;
    v4f nxt[TG_MAXQ];
    auto request = [&](int64_t tile) {                                   // rows of `tile` -> registers (zeros past the matrix)
        const int64_t i0 = tile * 32;
        const int total = tile < n_tiles ? (int)((p.n - i0) < 32 ? (p.n - i0) : 32) * K : 0;
        const float *src = p.X + i0 * K;
#pragma unroll
        for (int q = 0; q < TG_MAXQ; ++q) {
            const int e = lane * 4 + 256 * q;
            nxt[q] = v4f{0.f, 0.f, 0.f, 0.f};
            if (e + 4 <= total) nxt[q] = gload4(src + e);
            else if (e < total) {                                        // tail of a partial last tile
                float t4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int u = 0; u < 4; ++u) if (e + u < total) t4[u] = gload1(src + e + u);
                nxt[q] = v4f{t4[0], t4[1], t4[2], t4[3]};
            }
        }
    };
    for (int64_t tile = (int64_t)blockIdx.x * TG_WAVES + wave; tile < n_tiles; tile += stride) {
        const int64_t i0 = tile * 32;
        const int nvalid = (int)((p.n - i0) < 32 ? (p.n - i0) : 32);
        request(tile);
        int lane_o = lane;                      // opaque per tile: the 72 (row, column) pairs below are loop-invariant and would otherwise be
        asm volatile("" : "+v"(lane_o));        // hoisted out of the tile loop and kept in registers across the K-steps (163 spills)
#pragma unroll
        for (int q = 0; q < TG_MAXQ; ++q) {
            const int e = lane_o * 4 + 256 * q;
            if (e < 32 * K) {
                const float v[4] = {nxt[q].x, nxt[q].y, nxt[q].z, nxt[q].w};
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ee = e + u, r = (int)(((float)ee + 0.5f) * inv_k), c = ee - r * K;
                    X[r * KP + c] = v[u];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        f32x16 acc[NO];
        zero_acc<NO>(acc);
        layer_from_lds<NO>(X + node * KP + half, p.wp + (size_t)lane * NO, p.kk, acc, 1);
        int half_o = half;                      // (opaque per tile as well: the 64 bias values / output offsets of the epilogue are loop-invariant too)
        asm volatile("" : "+v"(half_o));
        if (node < nvalid) {
            const int64_t row = i0 + node;
            const int half = half_o;
            if (p.mode == 0) {
                switch (p.act) {
                case GNN_ACT_RELU: gemm_store_fwd<GNN_ACT_RELU, NO>(p, acc, row, half, vec); break;
                case GNN_ACT_SELU: gemm_store_fwd<GNN_ACT_SELU, NO>(p, acc, row, half, vec); break;
                case GNN_ACT_ELU: gemm_store_fwd<GNN_ACT_ELU, NO>(p, acc, row, half, vec); break;
                case GNN_ACT_TANH: gemm_store_fwd<GNN_ACT_TANH, NO>(p, acc, row, half, vec); break;
                case GNN_ACT_SIGMOID: gemm_store_fwd<GNN_ACT_SIGMOID, NO>(p, acc, row, half, vec); break;
                default: gemm_store_fwd<GNN_ACT_LINEAR, NO>(p, acc, row, half, vec); break;       // (softmax is applied by the caller)
                }
            } else {
#pragma unroll
                for (int jt = 0; jt < NO; ++jt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int f0 = p.col0 + 32 * jt + 8 * q + 4 * half;
                        const int64_t o = row * p.n_cols + f0;
                        float v[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            float x = acc[jt][4 * q + u];
                            if (f0 + u < p.n_cols) {
                                if (p.keep) x = dropout_grad(x, p.keep[o + u], p.rate);
                                if (p.act >= 0) x = x * act_grad(p.a_prev[o + u], p.act);
                            }
                            v[u] = x;
                        }
                        if (vec && f0 + 4 <= p.n_cols) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(p.Y) + o) = v4f{v[0], v[1], v[2], v[3]};
                        else
                            for (int u = 0; u < 4; ++u) if (f0 + u < p.n_cols) gptr_w(p.Y)[o + u] = v[u];
                    }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // the next tile re-uses this wave's LDS region
    }
}

// The same product in the split arithmetic of the fused inference kernel (gnn_fused_kernel.h: every fp32 operand cut into three exact
// bf16 pieces, six piece products per term on v_mfma_f32_32x32x16_bf16, fp32 accumulation): 2.7 x fewer matrix-pipe cycles than the f32
// MFMA, and the bf16 MFMA overlaps the wave's VALU work.  Packed operand: [K = 16 chunk][out tile][piece][lane][8 bf16] + two zero chunks.
// hidden: the k order of a layer whose input is the previous layer's accumulators (gnn_fused_kernel.h: chunk c, element i of k half h is
// feature 32 (c >> 1) + (r & 3) + 8 (r >> 2) + 4 h, r = 8 (c & 1) + i); fold: factor on every weight (the folded SELU of the fused chain)
__global__ void k_pack_split(int K, int n_cols, int col0, int NO, int chunks_img, const float *__restrict__ M, uint32_t *__restrict__ out, int hidden = 0,
                             float fold = 1.0f)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;                 // one thread per (chunk, tile, lane, element pair): three dwords (pieces)
    if (t >= chunks_img * NO * 64 * 4) return;
    const int j2 = t & 3, lane = (t >> 2) & 63, jt = (t >> 8) % NO, c = (t >> 8) / NO;
    uint32_t d[3] = {0u, 0u, 0u};
    for (int e = 0; e < 2; ++e) {
        const int i = 2 * j2 + e, h = lane >> 5, r = 8 * (c & 1) + i, col = col0 + 32 * jt + (lane & 31);
        const int k = hidden ? 32 * (c >> 1) + (r & 3) + 8 * (r >> 2) + 4 * h : 16 * c + 8 * h + i;
        float v = (k < K && col < n_cols) ? M[(size_t)k * n_cols + col] * fold : 0.0f;
        for (int pc = 0; pc < 3; ++pc) {                                 // truncation split: v == p0 + p1 + p2 exactly
            const uint32_t hi = __float_as_uint(v) & 0xffff0000u;
            v = v - __uint_as_float(hi);
            d[pc] |= e ? hi : (hi >> 16);
        }
    }
    for (int pc = 0; pc < 3; ++pc) out[((((size_t)c * NO + jt) * 3 + pc) * 64 + lane) * 4 + j2] = d[pc];
}

template <int NO>
__global__ void __launch_bounds__(64 * TG_WAVES, 2) k_gemm_split(const GemmArgs p)
{
    using namespace gnn_fused_dev;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = p.K, KP = p.KP;                                        // KP: a multiple of 4 with KP / 4 odd (16-byte rows, conflict-free b128 column reads)
    float *X = lds + (size_t)wave * 32 * KP;
    float *bias_lds = lds + (size_t)TG_WAVES * 32 * KP + 32;             // [32 NO]: the accumulators start from it (zeros in backward mode)
    for (int t = threadIdx.x; t < 32 * NO; t += blockDim.x) bias_lds[t] = (p.mode == 0 && p.col0 + t < p.n_cols) ? p.bias[p.col0 + t] : 0.0f;
    for (int t = lane; t < 32 * (KP - K); t += 64) X[(t / (KP - K)) * KP + K + t % (KP - K)] = 0.0f;      // columns >= K: zero, once
    __syncthreads();
    const int64_t n_tiles = (p.n + 31) / 32;
    const float inv_k = 1.0f / (float)K;
    const int half = lane >> 5, node = lane & 31;
    const int64_t stride = (int64_t)gridDim.x * TG_WAVES;
    const bool vec = (p.n_cols & 3) == 0, kvec = (K & 3) == 0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(p.wp), 0, p.kk, 0x00020000);      // (kk: bytes of the packed image here)
    // start-up spread (as k_fused): all waves run the same phases - load, K-steps, store - on tiles of equal cost; started together they
    // would load together and compute together.  Every wave waits a different fraction of about one tile period first.
    if (n_tiles >= 4 * stride) {
        const int rounds = (int)((((unsigned)blockIdx.x * TG_WAVES + (unsigned)wave) * 0x9E3779B1u) >> 16) % (unsigned)(p.spread + 1);
        for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
    }
    v4f nxt[TG_MAXQ];
    for (int64_t tile = (int64_t)blockIdx.x * TG_WAVES + wave; tile < n_tiles; tile += stride) {
        const int64_t i0 = tile * 32;
        const int nvalid = (int)((p.n - i0) < 32 ? (p.n - i0) : 32);
        const int total = nvalid * K;
        const float *src = p.X + i0 * K;
#pragma unroll
        for (int q = 0; q < TG_MAXQ; ++q) {                              // all pieces of the tile requested before the first is used
            const int e = lane * 4 + 256 * q;
            nxt[q] = v4f{0.f, 0.f, 0.f, 0.f};
            if (e + 4 <= total) nxt[q] = gload4(src + e);
            else if (e < total) {                                        // tail of a partial last tile
                float t4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int u = 0; u < 4; ++u) if (e + u < total) t4[u] = gload1(src + e + u);
                nxt[q] = v4f{t4[0], t4[1], t4[2], t4[3]};
            }
        }
        int lane_o = lane;                      // opaque per tile (see k_gemm_f32)
        asm volatile("" : "+v"(lane_o));
        if (K < 32 * NO) {                      // the previous tile's output pass left values in columns [K, 32 NO): zero again (0 x Inf would poison the sums)
            const int zw = 32 * NO - K;
            for (int t = lane_o; t < 32 * zw; t += 64) X[(t / zw) * KP + K + t % zw] = 0.0f;
        }
#pragma unroll
        for (int q = 0; q < TG_MAXQ; ++q) {
            const int e = lane_o * 4 + 256 * q;
            if (e < 32 * K) {
                if (kvec) {                                              // rows are whole 16-byte pieces
                    const int r = (int)(((float)e + 0.5f) * inv_k), c = e - r * K;
                    *reinterpret_cast<v4f *>(X + r * KP + c) = nxt[q];
                } else {
                    const float v[4] = {nxt[q].x, nxt[q].y, nxt[q].z, nxt[q].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ee = e + u, r = (int)(((float)ee + 0.5f) * inv_k), c = ee - r * K;
                        X[r * KP + c] = v[u];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        f32x16 acc[NO];
        layer0_split<NO, true>(X + node * KP + 8 * half, wrs, lane * 16, 0, (K + 15) / 16, acc, bias_lds, half);
        // Epilogue in two steps, so that every global access is a whole row piece: (1) the accumulators (feature on the register, row on
        // the lane) go to the wave's LDS tile as [row][column] (16-byte pieces, row stride KP: KP / 4 odd, conflict-free); (2) lanes take
        // consecutive 16-byte pieces of consecutive rows - 512 contiguous bytes per 32 lanes for a 128-wide pass - read the matching
        // pieces of the producer's activation / Dropout mask, apply bias-included activation or the derivative, and store.  (Stores of
        // 16-byte pieces straight from the accumulator layout touch 32 rows per instruction: 0.63 ms per product whatever K.)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // all B-operand reads of the tile are done: its LDS region is free
        int half_o = half;
        asm volatile("" : "+v"(half_o));
#pragma unroll
        for (int jt = 0; jt < NO; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<v4f *>(X + node * KP + 32 * jt + 8 * q + 4 * half_o) = v4f{acc[jt][4 * q], acc[jt][4 * q + 1], acc[jt][4 * q + 2], acc[jt][4 * q + 3]};
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        constexpr int PPR = 8 * NO;                                      // 16-byte pieces per row of this pass
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
#pragma unroll
        for (int u = 0; u < 32 * PPR / 64; ++u) {
            const int idx = lane_p + 64 * u, r = idx / PPR, c = (idx % PPR) * 4;
            const int f0 = p.col0 + c;
            if (r < nvalid && f0 < p.n_cols) {
                const v4f a4 = *reinterpret_cast<const v4f *>(X + r * KP + c);
                float v[4] = {a4.x, a4.y, a4.z, a4.w};
                const int64_t o = (i0 + r) * p.n_cols + f0;
                const bool full = vec && f0 + 4 <= p.n_cols;
                if (p.mode == 0) {
                    // hardware transcendentals (v_exp_f32 / v_rcp_f32, 1 ulp: act_fast of the fused inference path); the backward pass
                    // differentiates from the stored activation, so forward and backward stay consistent
                    switch (p.act) {
                    case GNN_ACT_RELU: for (int t = 0; t < 4; ++t) v[t] = act_fast<GNN_ACT_RELU>(v[t]); break;
                    case GNN_ACT_SELU: for (int t = 0; t < 4; ++t) v[t] = act_fast<GNN_ACT_SELU>(v[t]); break;
                    case GNN_ACT_ELU: for (int t = 0; t < 4; ++t) v[t] = act_fast<GNN_ACT_ELU>(v[t]); break;
                    case GNN_ACT_TANH: for (int t = 0; t < 4; ++t) v[t] = act_fast<GNN_ACT_TANH>(v[t]); break;
                    case GNN_ACT_SIGMOID: for (int t = 0; t < 4; ++t) v[t] = act_fast<GNN_ACT_SIGMOID>(v[t]); break;
                    default: break;
                    }
                } else {
                    if (p.keep)
                        for (int t = 0; t < 4; ++t) if (f0 + t < p.n_cols) v[t] = dropout_grad(v[t], p.keep[o + t], p.rate);
                    if (p.act >= 0) {
                        if (full) {
                            const v4f ap = gload4(p.a_prev + o);
                            v[0] *= act_grad(ap.x, p.act); v[1] *= act_grad(ap.y, p.act); v[2] *= act_grad(ap.z, p.act); v[3] *= act_grad(ap.w, p.act);
                        } else
                            for (int t = 0; t < 4; ++t) if (f0 + t < p.n_cols) v[t] *= act_grad(p.a_prev[o + t], p.act);
                    }
                }
                if (full) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(p.Y) + o) = v4f{v[0], v[1], v[2], v[3]};
                else
                    for (int t = 0; t < 4; ++t) if (f0 + t < p.n_cols) gptr_w(p.Y)[o + t] = v[t];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");           // the next tile re-uses this wave's LDS region
    }
}

// The three Dense layers of a 3-layer net_state in ONE pass over the rows (round 3): the chain of the fused inference kernel - layer 0 from
// the LDS tile, the hidden layers from the previous accumulators without leaving registers (layer_split_from_regs, folded SELU) - with
// the activations the backward pass differentiates written out on the way (a0, a1 after the layer that consumes them has cut them into
// pieces, a2 at the end), each through the wave's LDS tile as whole row pieces.  Saves re-reading a0 and a1 (2 x 512 MB at 1 M rows) and
// two stagings.  Shape: hidden width <= 128 (four 32-feature tiles), last width <= 64 (two).
struct Fwd3Args {
    int64_t n;
    int K, KP, chunks0, w1, w2, w3, act;
    int img_bytes, off1, off2;               // one packed image for the three layers: byte offsets of layers 1 and 2
    const float *X, *b0, *b1, *b2;
    const uint32_t *img;
    float *A0, *A1, *A2;
};

template <int ACT>
__global__ void __launch_bounds__(64 * TG_WAVES, 2) k_fwd3_split(const Fwd3Args p)
{
    using namespace gnn_fused_dev;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr bool FOLD = ACT == GNN_ACT_SELU;
    constexpr float LOG2E = 1.44269504088896341f, UNFOLD = FOLD ? 1.0507009873554805f / LOG2E : 1.0f;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int K = p.K, KP = p.KP;
    float *X = lds + (size_t)wave * 32 * KP;
    float *hb = lds + (size_t)TG_WAVES * 32 * KP + 32;                   // biases: layer 0 [128] | layer 1 [128] | layer 2 [64]
    for (int t = threadIdx.x; t < 320; t += blockDim.x) {
        float v = 0.0f;
        if (t < 128) v = t < p.w1 ? p.b0[t] * (FOLD ? LOG2E : 1.0f) : 0.0f;
        else if (t < 256) v = t - 128 < p.w2 ? p.b1[t - 128] * (FOLD ? LOG2E : 1.0f) : 0.0f;
        else v = t - 256 < p.w3 ? p.b2[t - 256] : 0.0f;
        hb[t] = v;
    }
    __syncthreads();
    const int64_t n_tiles = (p.n + 31) / 32, stride = (int64_t)gridDim.x * TG_WAVES;
    const float inv_k = 1.0f / (float)K;
    const int half = lane >> 5, node = lane & 31;
    const bool kvec = (K & 3) == 0;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.img), 0, p.img_bytes, 0x00020000);
    // rows of one activation array through the LDS tile: accumulator layout -> [row][column] -> whole row pieces to memory
    auto store_rows = [&](auto &h, auto NTc, float *dst, int width, int nvalid, int64_t i0, float scale, bool activate) {
        constexpr int NTT = decltype(NTc)::value;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        int half_o = half;
        asm volatile("" : "+v"(half_o));
#pragma unroll
        for (int jt = 0; jt < NTT; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                v4f v = {h[jt][4 * q], h[jt][4 * q + 1], h[jt][4 * q + 2], h[jt][4 * q + 3]};
                if (activate) v = v4f{act_fast<ACT>(v.x), act_fast<ACT>(v.y), act_fast<ACT>(v.z), act_fast<ACT>(v.w)};
                *reinterpret_cast<v4f *>(X + node * KP + 32 * jt + 8 * q + 4 * half_o) = v * scale;
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        constexpr int PPR = 8 * NTT;
        const bool vec = (width & 3) == 0;
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
#pragma unroll
        for (int u = 0; u < 32 * PPR / 64; ++u) {
            const int idx = lane_p + 64 * u, r = idx / PPR, c = (idx % PPR) * 4;
            if (r < nvalid && c < width) {
                const v4f a4 = *reinterpret_cast<const v4f *>(X + r * KP + c);
                const int64_t o = (i0 + r) * width + c;
                if (vec) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(dst) + o) = a4;
                else {
                    const float v[4] = {a4.x, a4.y, a4.z, a4.w};
                    for (int t = 0; t < 4; ++t) if (c + t < width) gptr_w(dst)[o + t] = v[t];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    v4f nxt[TG_MAXQ];
    for (int64_t tile = (int64_t)blockIdx.x * TG_WAVES + wave; tile < n_tiles; tile += stride) {
        const int64_t i0 = tile * 32;
        const int nvalid = (int)((p.n - i0) < 32 ? (p.n - i0) : 32);
        const int total = nvalid * K;
        const float *src = p.X + i0 * K;
#pragma unroll
        for (int q = 0; q < TG_MAXQ; ++q) {
            const int e = lane * 4 + 256 * q;
            nxt[q] = v4f{0.f, 0.f, 0.f, 0.f};
            if (e + 4 <= total) nxt[q] = gload4(src + e);
            else if (e < total) {
                float t4[4] = {0.f, 0.f, 0.f, 0.f};
                for (int u = 0; u < 4; ++u) if (e + u < total) t4[u] = gload1(src + e + u);
                nxt[q] = v4f{t4[0], t4[1], t4[2], t4[3]};
            }
        }
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        // (the stores of the previous tile left values in columns [K, KP) of the tile region: zero them again - 0 x Inf would poison the sums)
        for (int t = lane_o; t < 32 * (KP - K); t += 64) X[(t / (KP - K)) * KP + K + t % (KP - K)] = 0.0f;
#pragma unroll
        for (int q = 0; q < TG_MAXQ; ++q) {
            const int e = lane_o * 4 + 256 * q;
            if (e < 32 * K) {
                if (kvec) {
                    const int r = (int)(((float)e + 0.5f) * inv_k), c = e - r * K;
                    *reinterpret_cast<v4f *>(X + r * KP + c) = nxt[q];
                } else {
                    const float v[4] = {nxt[q].x, nxt[q].y, nxt[q].z, nxt[q].w};
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const int ee = e + u, r = (int)(((float)ee + 0.5f) * inv_k), c = ee - r * K;
                        X[r * KP + c] = v[u];
                    }
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        f32x16 h1[4], h2[4], out[2];
        layer0_split<4, true>(X + node * KP + 8 * half, wrs, lane * 16, 0, p.chunks0, h1, hb, half);
        layer_split_from_regs<4, 4, ACT>(h1, hb + 128, half, h2, wrs, lane * 16, p.off1);       // h1 now holds the (folded) activations of layer 0
        store_rows(h1, std::integral_constant<int, 4>{}, p.A0, p.w1, nvalid, i0, UNFOLD, false);
        layer_split_from_regs<4, 2, ACT>(h2, hb + 256, half, out, wrs, lane * 16, p.off2);
        store_rows(h2, std::integral_constant<int, 4>{}, p.A1, p.w2, nvalid, i0, UNFOLD, false);
        store_rows(out, std::integral_constant<int, 2>{}, p.A2, p.w3, nvalid, i0, 1.0f, true);
    }
}

// K-steps of the wide products: a multiple of 12 (layer_from_lds consumes groups of 3 x 4) plus its look-ahead of 8
inline int tg_kk(int K) { return ((K + 1) / 2 + 11) / 12 * 12; }
inline int tg_kp(int K) { return std::max(2 * tg_kk(K), (K + 15) / 16 * 16) + 1; }
inline bool tg_many_rows(int64_t n)
{
#ifdef GNN_DIAG      // GNN_TRAIN_MFMA=0: the round-2 kernels everywhere (accuracy / timing comparison)
    static const bool off = getenv("GNN_TRAIN_MFMA") && atoi(getenv("GNN_TRAIN_MFMA")) == 0;
    if (off) return false;
#endif
    return n >= GNN_TRAIN_MFMA_MIN_ROWS;
}
inline bool tg_wide(int n_in, int n_out) { return n_in >= 64 && n_out >= 32 && n_in <= 144; }      // (TG_MAXQ pieces of a 32-row tile per lane)

// Y = epilogue(X . M) over all column passes of M [K, n_cols]; scratch for the packed operand comes from the step's arena
inline int tg_kps(int K) { int kp = ((K + 15) / 16 * 16 + 3) / 4 * 4; if ((kp / 4) % 2 == 0) kp += 4; return kp; }

template <class BufT>      // (Buf is defined further down with the arena)
int launch_gemm_f32(hipStream_t st, BufT &buf, int64_t n, int K, int n_cols, const float *X, const float *M, const float *bias, int act, int mode,
                    const uint8_t *keep, float rate, const float *a_prev, float *Y)
{
    static bool raised = false;
    if (!raised) {
        const void *ks[6] = {reinterpret_cast<const void *>(&k_gemm_f32<4>), reinterpret_cast<const void *>(&k_gemm_f32<2>), reinterpret_cast<const void *>(&k_gemm_f32<1>),
                             reinterpret_cast<const void *>(&k_gemm_split<4>), reinterpret_cast<const void *>(&k_gemm_split<2>), reinterpret_cast<const void *>(&k_gemm_split<1>)};
        for (const void *k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    bool split = true;                             // shipped: the split-bf16 products; the f32-MFMA form stays as the exact-chain cross-check
#ifdef GNN_DIAG
    static const bool f32_env = getenv("GNN_TRAIN_GEMM_F32") != nullptr;
    split = !f32_env;
#endif
    GemmArgs p{};
    p.n = n; p.K = K; p.n_cols = n_cols; p.act = act; p.mode = mode; p.rate = rate;
    p.X = X; p.bias = bias; p.a_prev = a_prev; p.keep = keep; p.Y = Y;
    p.spread = 0;                                  // (measured: 0 .. 8 rounds of start-up spread change nothing here, profiles/r03_train_c3.txt)
#ifdef GNN_DIAG
    static const int spread_env = getenv("GNN_TRAIN_SPREAD") ? atoi(getenv("GNN_TRAIN_SPREAD")) : 0;
    p.spread = spread_env;
#endif
    p.KP = split ? std::max(tg_kps(K), tg_kps(std::min(128, (n_cols + 31) / 32 * 32))) : tg_kp(K);      // (split: the tile is re-used for the pass's output columns)
    const size_t lds = sizeof(float) * ((size_t)TG_WAVES * 32 * p.KP + 32 + 128) + 16;
    if (lds > 160 * 1024) return gnn_fail(GNN_ERR_UNSUPPORTED, "layer input width %d too large for the matrix-core path", K);
    const int64_t n_tiles = (n + 31) / 32;
    const unsigned grid = (unsigned)std::min<int64_t>(256, (n_tiles + TG_WAVES - 1) / TG_WAVES);
    for (int col0 = 0; col0 < n_cols;) {
        const int left = (n_cols - col0 + 31) / 32, NO = left >= 4 ? 4 : (left >= 2 ? 2 : 1);
        int rc;
        p.col0 = col0;
        if (split) {
            const int chunks_img = (K + 15) / 16 + 2;
            uint32_t *img = nullptr;
            if ((rc = buf.get(&img, (size_t)chunks_img * NO * 3 * 256))) return rc;
            hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)chunks_img * NO * 256, 256), 256, 0, st, K, n_cols, col0, NO, chunks_img, M, img);
            p.wp = reinterpret_cast<const float *>(img);
            p.kk = (int)((size_t)chunks_img * NO * 3 * 256 * sizeof(uint32_t));      // bytes of the image (buffer descriptor)
            if (NO == 4) hipLaunchKernelGGL((k_gemm_split<4>), grid, 64 * TG_WAVES, lds, st, p);
            else if (NO == 2) hipLaunchKernelGGL((k_gemm_split<2>), grid, 64 * TG_WAVES, lds, st, p);
            else hipLaunchKernelGGL((k_gemm_split<1>), grid, 64 * TG_WAVES, lds, st, p);
        } else {
            p.kk = tg_kk(K);
            float *wp = nullptr;
            const int kk_img = p.kk + 8;
            if ((rc = buf.get(&wp, (size_t)kk_img * 64 * NO))) return rc;
            hipLaunchKernelGGL(k_pack_exact, cdiv((int64_t)kk_img * 64 * NO, 256), 256, 0, st, K, n_cols, col0, NO, kk_img, M, wp);
            p.wp = wp;
            if (NO == 4) hipLaunchKernelGGL((k_gemm_f32<4>), grid, 64 * TG_WAVES, lds, st, p);
            else if (NO == 2) hipLaunchKernelGGL((k_gemm_f32<2>), grid, 64 * TG_WAVES, lds, st, p);
            else hipLaunchKernelGGL((k_gemm_f32<1>), grid, 64 * TG_WAVES, lds, st, p);
        }
        HIPCHK(hipGetLastError());
        col0 += 32 * NO;
    }
    return GNN_OK;
}

inline bool fwd3_covers(const gnn_mlp *m)
{
    if (m->n_layers != 3) return false;
    const int a = m->acts[0];
    if (a == GNN_ACT_SOFTMAX || m->acts[1] != a || m->acts[2] != a) return false;
    return m->dims[0] >= 64 && m->dims[0] <= 144 && m->dims[1] > 64 && m->dims[1] <= 128 && m->dims[2] > 64 && m->dims[2] <= 128 && m->dims[3] > 32 && m->dims[3] <= 64;
}

template <class BufT>
int launch_fwd3(hipStream_t st, BufT &buf, const gnn_mlp *m, int64_t n, const float *x, float *a0, float *a1, float *a2)
{
    static bool raised = false;
    const void *ks[6] = {reinterpret_cast<const void *>(&k_fwd3_split<GNN_ACT_LINEAR>), reinterpret_cast<const void *>(&k_fwd3_split<GNN_ACT_RELU>),
                         reinterpret_cast<const void *>(&k_fwd3_split<GNN_ACT_SELU>), reinterpret_cast<const void *>(&k_fwd3_split<GNN_ACT_ELU>),
                         reinterpret_cast<const void *>(&k_fwd3_split<GNN_ACT_TANH>), reinterpret_cast<const void *>(&k_fwd3_split<GNN_ACT_SIGMOID>)};
    if (!raised) {
        for (const void *k : ks) (void)hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    Fwd3Args p{};
    p.n = n; p.K = m->dims[0]; p.w1 = m->dims[1]; p.w2 = m->dims[2]; p.w3 = m->dims[3]; p.act = m->acts[0];
    p.KP = std::max(tg_kps(p.K), tg_kps(128));
    p.chunks0 = (p.K + 15) / 16;
    const size_t blk = 3 * 256;                                          // dwords per (chunk, tile)
    const size_t d0 = (size_t)(p.chunks0 + 2) * 4 * blk, d1 = (size_t)8 * 4 * blk, d2 = (size_t)8 * 2 * blk;
    uint32_t *img = nullptr;
    int rc = buf.get(&img, d0 + d1 + d2);
    if (rc) return rc;
    const bool fold = p.act == GNN_ACT_SELU;
    const float LOG2E = 1.44269504088896341f, SCALE = 1.0507009873554805f;
    hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)(p.chunks0 + 2) * 4 * 256, 256), 256, 0, st, p.K, p.w1, 0, 4, p.chunks0 + 2, m->W[0], img, 0, fold ? LOG2E : 1.0f);
    hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)8 * 4 * 256, 256), 256, 0, st, p.w1, p.w2, 0, 4, 8, m->W[1], img + d0, 1, fold ? SCALE : 1.0f);
    hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)8 * 2 * 256, 256), 256, 0, st, p.w2, p.w3, 0, 2, 8, m->W[2], img + d0 + d1, 1, fold ? SCALE / LOG2E : 1.0f);
    p.img = img; p.img_bytes = (int)((d0 + d1 + d2) * sizeof(uint32_t)); p.off1 = (int)(d0 * sizeof(uint32_t)); p.off2 = (int)((d0 + d1) * sizeof(uint32_t));
    p.X = x; p.b0 = m->b[0]; p.b1 = m->b[1]; p.b2 = m->b[2]; p.A0 = a0; p.A1 = a1; p.A2 = a2;
    const size_t lds = sizeof(float) * ((size_t)TG_WAVES * 32 * p.KP + 32 + 320) + 16;
    if (lds > 160 * 1024) return gnn_fail(GNN_ERR_UNSUPPORTED, "fused forward: LDS");
    const int64_t n_tiles = (n + 31) / 32;
    const unsigned grid = (unsigned)std::min<int64_t>(256, (n_tiles + TG_WAVES - 1) / TG_WAVES);
    switch (p.act) {
    case GNN_ACT_LINEAR: hipLaunchKernelGGL((k_fwd3_split<GNN_ACT_LINEAR>), grid, 64 * TG_WAVES, lds, st, p); break;
    case GNN_ACT_RELU: hipLaunchKernelGGL((k_fwd3_split<GNN_ACT_RELU>), grid, 64 * TG_WAVES, lds, st, p); break;
    case GNN_ACT_SELU: hipLaunchKernelGGL((k_fwd3_split<GNN_ACT_SELU>), grid, 64 * TG_WAVES, lds, st, p); break;
    case GNN_ACT_ELU: hipLaunchKernelGGL((k_fwd3_split<GNN_ACT_ELU>), grid, 64 * TG_WAVES, lds, st, p); break;
    case GNN_ACT_TANH: hipLaunchKernelGGL((k_fwd3_split<GNN_ACT_TANH>), grid, 64 * TG_WAVES, lds, st, p); break;
    default: hipLaunchKernelGGL((k_fwd3_split<GNN_ACT_SIGMOID>), grid, 64 * TG_WAVES, lds, st, p); break;
    }
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

// The backward chain of the same 3-layer net in ONE pass over the rows (round 5): d z2 (the gradient at the last layer's pre-activation) ->
//     d z1 = (d z2 . W2^T) * act'(a1)  ->  d z0 = (d z1 . W1^T) * act'(a0)  ->  d inp = d z0 . W0^T
// with the chain of the fused kernels - the first product from the wave's LDS tile, the following ones from the previous accumulators without
// leaving registers (layer_split_from_regs with the identity in place of the activation) - and d z1, d z0 (operands of the weight gradients)
// and d inp written out on the way, each through the LDS tile as whole row pieces.  The stored activations a1 / a0 pass through the same tile
// (coalesced rows in, accumulator layout out).  Replaces three k_gemm_split passes (+ the narrow fourth for columns >= 128 of d inp): d z1 and
// d z0 are no longer re-read and re-staged (2 x 512 MB at 1 M rows), one launch instead of four.
struct Bwd3Args {
    int64_t n;
    int K0, w1, w2, w3, KP, chunksA, act;
    int img_bytes, offB, offC, offD;         // one packed image: W2^T (from LDS, 4 tiles) | W1^T (from registers, 4 tiles) | W0^T columns [0, 128) | [128, K0)
    const float *DZ2, *A1, *A0;
    const uint32_t *img;
    float *DZ1, *DZ0, *DINP;
    // optional (DSG != nullptr): the two column blocks of d inp the state gradient reads - own state [0, Ds) and aggregated state [c_aggs, c_aggs + Ds) -
    // once more as 16-byte aligned rows [n, 2 Ds] (the concat's rows are 135 floats long and its aggregate block starts at column 67: k_state_grad_rows
    // gathers ten rows of it per node with 4-byte loads; from the aligned copy with 16-byte loads)
    float *DSG;
    int Ds, c_aggs;
};

__global__ void __launch_bounds__(64 * TG_WAVES, 2) k_bwd3_split(const Bwd3Args p)
{
    using namespace gnn_fused_dev;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int KP = p.KP;
    float *X = lds + (size_t)wave * 32 * KP;
    float *zb = lds + (size_t)TG_WAVES * 32 * KP + 32;                   // [128] zeros: the accumulators start from it (no bias in a backward product)
    for (int t = threadIdx.x; t < 128; t += blockDim.x) zb[t] = 0.0f;
    __syncthreads();
    const int64_t n_tiles = (p.n + 31) / 32, stride = (int64_t)gridDim.x * TG_WAVES;
    const int half = lane >> 5, node = lane & 31;
    const __amdgpu_buffer_rsrc_t wrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint32_t *>(p.img), 0, p.img_bytes, 0x00020000);
    // 32 rows of a dense [n, width] array (width a multiple of 4, <= 128) into the tile as [row][column]: all pieces requested before the first is
    // written (one round trip, covered by the SIMD's other wave); columns [width, zero_to) are zeroed
    auto stage_rows = [&](const float *src_all, int width, int zero_to, int nvalid, int64_t i0) {
        constexpr int MAXQ = 16;                                         // 32 x 128 floats = 16 pieces of 16 bytes per lane
        v4f nxt[MAXQ];
        const int total = nvalid * width;
        const float *src = src_all + i0 * width;
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int e = lane * 4 + 256 * q;
            nxt[q] = v4f{0.f, 0.f, 0.f, 0.f};
            if (e < total) nxt[q] = gload4(src + e);
        }
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const float inv_w = 1.0f / (float)width;
        if (zero_to > width) {
            const int zw = zero_to - width;
            for (int t = lane_o; t < 32 * zw; t += 64) X[(t / zw) * KP + width + t % zw] = 0.0f;
        }
#pragma unroll
        for (int q = 0; q < MAXQ; ++q) {
            const int e = lane_o * 4 + 256 * q;
            if (e < 32 * width) {
                const int r = (int)(((float)e + 0.5f) * inv_w), c = e - r * width;
                *reinterpret_cast<v4f *>(X + r * KP + c) = nxt[q];
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // accumulators (feature on the register, row on the lane) times act'(stored activation), the activations read from the tile in the same layout
    auto times_act_grad = [&](f32x16 (&h)[4]) {
        int half_o = half;
        asm volatile("" : "+v"(half_o));
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const v4f a4 = *reinterpret_cast<const v4f *>(X + node * KP + 32 * jt + 8 * q + 4 * half_o);
                h[jt][4 * q] *= act_grad(a4.x, p.act); h[jt][4 * q + 1] *= act_grad(a4.y, p.act);
                h[jt][4 * q + 2] *= act_grad(a4.z, p.act); h[jt][4 * q + 3] *= act_grad(a4.w, p.act);
            }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // NTT accumulator tiles -> [row][column] in the tile -> whole row pieces to columns [col0, col0 + 32 NTT) of dst [n, width]
    auto store_rows = [&](auto &h, auto NTc, float *dst, int width, int col0, int nvalid, int64_t i0) {
        constexpr int NTT = decltype(NTc)::value;
        int half_o = half;
        asm volatile("" : "+v"(half_o));
#pragma unroll
        for (int jt = 0; jt < NTT; ++jt)
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *reinterpret_cast<v4f *>(X + node * KP + 32 * jt + 8 * q + 4 * half_o) = v4f{h[jt][4 * q], h[jt][4 * q + 1], h[jt][4 * q + 2], h[jt][4 * q + 3]};
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        constexpr int PPR = 8 * NTT;
        const bool vec = (width & 3) == 0 && (col0 & 3) == 0;
        int lane_p = lane;
        asm volatile("" : "+v"(lane_p));
#pragma unroll
        for (int u = 0; u < 32 * PPR / 64; ++u) {
            const int idx = lane_p + 64 * u, r = idx / PPR, c = (idx % PPR) * 4;
            if (r < nvalid && col0 + c < width) {
                const v4f a4 = *reinterpret_cast<const v4f *>(X + r * KP + c);
                const int64_t o = (i0 + r) * width + col0 + c;
                if (vec && col0 + c + 4 <= width) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(dst) + o) = a4;
                else {
                    const float v[4] = {a4.x, a4.y, a4.z, a4.w};
                    for (int t = 0; t < 4; ++t) if (col0 + c + t < width) gptr_w(dst)[o + t] = v[t];
                }
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    for (int64_t tile = (int64_t)blockIdx.x * TG_WAVES + wave; tile < n_tiles; tile += stride) {
        const int64_t i0 = tile * 32;
        const int nvalid = (int)((p.n - i0) < 32 ? (p.n - i0) : 32);
        f32x16 g[4], acc[4];
        // d z2 tile -> d h2 = d z2 . W2^T
        stage_rows(p.DZ2, p.w3, 16 * p.chunksA, nvalid, i0);
        layer0_split<4, true>(X + node * KP + 8 * half, wrs, lane * 16, 0, p.chunksA, g, zb, half);
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        stage_rows(p.A1, p.w2, 128, nvalid, i0);
        times_act_grad(g);                                               // g = d z1
        store_rows(g, std::integral_constant<int, 4>{}, p.DZ1, p.w2, 0, nvalid, i0);
        layer_split_from_regs<4, 4, GNN_ACT_LINEAR>(g, zb, half, acc, wrs, lane * 16, p.offB);
        stage_rows(p.A0, p.w1, 128, nvalid, i0);
        times_act_grad(acc);                                             // acc = d z0
        store_rows(acc, std::integral_constant<int, 4>{}, p.DZ0, p.w1, 0, nvalid, i0);
        layer_split_from_regs<4, 4, GNN_ACT_LINEAR>(acc, zb, half, g, wrs, lane * 16, p.offC);
        f32x16 tail[1];
        if (p.K0 > 128) layer_split_from_regs<4, 1, GNN_ACT_LINEAR>(acc, zb, half, tail, wrs, lane * 16, p.offD);
        // d inp: all K0 columns into the tile, then the tile's rows as ONE flat run of nvalid x K0 floats - it starts on a 16-byte boundary whatever
        // K0 is (32 K0 floats per tile), so memory is written in aligned 16-byte pieces even for K0 = 135 (a piece may straddle two tile rows)
        {
            int half_o = half;
            asm volatile("" : "+v"(half_o));
#pragma unroll
            for (int jt = 0; jt < 4; ++jt)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    *reinterpret_cast<v4f *>(X + node * KP + 32 * jt + 8 * q + 4 * half_o) = v4f{g[jt][4 * q], g[jt][4 * q + 1], g[jt][4 * q + 2], g[jt][4 * q + 3]};
            if (p.K0 > 128) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if (128 + 8 * q + 4 * half_o + 4 <= KP)
                        *reinterpret_cast<v4f *>(X + node * KP + 128 + 8 * q + 4 * half_o) = v4f{tail[0][4 * q], tail[0][4 * q + 1], tail[0][4 * q + 2], tail[0][4 * q + 3]};
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            const int K0 = p.K0, total = nvalid * K0;
            const float inv_k = 1.0f / (float)K0;
            float *dst = p.DINP + i0 * K0;
            int lane_p = lane;
            asm volatile("" : "+v"(lane_p));
            for (int e = 4 * lane_p; e < total; e += 256) {
                float v[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const int ee = e + t < total ? e + t : total - 1;
                    const int r = (int)(((float)ee + 0.5f) * inv_k), c = ee - r * K0;
                    v[t] = X[r * KP + c];
                }
                if (e + 4 <= total) *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(dst) + e) = v4f{v[0], v[1], v[2], v[3]};
                else
                    for (int t = 0; t < 4; ++t) if (e + t < total) gptr_w(dst)[e + t] = v[t];
            }
            if (p.DSG) {                                                 // [own | aggregate] column blocks as aligned rows
                const int Ds = p.Ds, ppr = Ds >> 1;                      // 16-byte pieces per row of the copy (2 Ds floats)
                float *sg = p.DSG + i0 * 2 * Ds;
                for (int idx = lane_p; idx < nvalid * ppr; idx += 64) {
                    const int r = idx / ppr, q = idx - r * ppr, c = 4 * q < Ds ? 4 * q : p.c_aggs + (4 * q - Ds);
                    const float *x = X + r * KP + c;
                    *reinterpret_cast<GNN_GLOBAL v4f *>(gptr_w(sg) + r * 2 * Ds + 4 * q) = v4f{x[0], x[1], x[2], x[3]};
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        }
    }
}

inline bool bwd3_covers(const gnn_mlp *m)
{
    return fwd3_covers(m) && (m->dims[1] & 3) == 0 && (m->dims[2] & 3) == 0 && (m->dims[3] & 3) == 0 && m->dims[0] <= 160;
}

// WT[l]: the transposed kernels [n_out, n_in] of the three layers (Net::WT)
template <class BufT>
int launch_bwd3(hipStream_t st, BufT &buf, const gnn_mlp *m, float *const *WT, int64_t n, const float *dz2, const float *a1, const float *a0, float *dz1,
                float *dz0, float *dinp, float *dsg = nullptr, int Ds = 0, int c_aggs = 0)
{
    static bool raised = false;
    if (!raised) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&k_bwd3_split), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised = true;
    }
    Bwd3Args p{};
    p.n = n; p.K0 = m->dims[0]; p.w1 = m->dims[1]; p.w2 = m->dims[2]; p.w3 = m->dims[3]; p.act = m->acts[0];
    p.KP = std::max(tg_kps(128), tg_kps(p.K0));
    p.chunksA = (p.w3 + 15) / 16;
    const size_t blk = 3 * 256;                                          // dwords per (chunk, tile)
    const size_t dA = (size_t)(p.chunksA + 2) * 4 * blk, dB = (size_t)8 * 4 * blk, dC = (size_t)8 * 4 * blk, dD = (size_t)8 * 1 * blk;
    uint32_t *img = nullptr;
    int rc = buf.get(&img, dA + dB + dC + dD);
    if (rc) return rc;
    // W2^T: [K = w3, n_cols = w2] from the tile (plain k order); W1^T: [w2, w1] and W0^T: [w1, K0] from the accumulators (hidden k order)
    hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)(p.chunksA + 2) * 4 * 256, 256), 256, 0, st, p.w3, p.w2, 0, 4, p.chunksA + 2, WT[2], img, 0, 1.0f);
    hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)8 * 4 * 256, 256), 256, 0, st, p.w2, p.w1, 0, 4, 8, WT[1], img + dA, 1, 1.0f);
    hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)8 * 4 * 256, 256), 256, 0, st, p.w1, p.K0, 0, 4, 8, WT[0], img + dA + dB, 1, 1.0f);
    hipLaunchKernelGGL(k_pack_split, cdiv((int64_t)8 * 1 * 256, 256), 256, 0, st, p.w1, p.K0, 128, 1, 8, WT[0], img + dA + dB + dC, 1, 1.0f);
    p.img = img; p.img_bytes = (int)((dA + dB + dC + dD) * sizeof(uint32_t));
    p.offB = (int)(dA * sizeof(uint32_t)); p.offC = (int)((dA + dB) * sizeof(uint32_t)); p.offD = (int)((dA + dB + dC) * sizeof(uint32_t));
    p.DZ2 = dz2; p.A1 = a1; p.A0 = a0; p.DZ1 = dz1; p.DZ0 = dz0; p.DINP = dinp;
    p.DSG = dsg; p.Ds = Ds; p.c_aggs = c_aggs;
    const size_t lds = sizeof(float) * ((size_t)TG_WAVES * 32 * p.KP + 32 + 128) + 16;
    if (lds > 160 * 1024) return gnn_fail(GNN_ERR_UNSUPPORTED, "fused backward: LDS");
    const int64_t n_tiles = (n + 31) / 32;
    const unsigned grid = (unsigned)std::min<int64_t>(256, (n_tiles + TG_WAVES - 1) / TG_WAVES);
    hipLaunchKernelGGL(k_bwd3_split, grid, 64 * TG_WAVES, lds, st, p);
    HIPCHK(hipGetLastError());
    return GNN_OK;
}

// [dW; db] partials of one row chunk: D[hf, zf] = sum over the chunk's rows of [H | 1][r, hf] d z[r, zf].  Rows are the K dimension of
// the 32x32x2 MFMA: lane (m, k half) loads H[r0 + 2 kk + k half][32 mt + m] and d z[..][32 nt + m] - whole 128-byte row pieces per
// half-wave, straight from memory, no staging.  Block = 4 waves, each a quarter of the chunk's rows, MT tiles of [H | 1] columns x up to
// two tiles of d z columns; the four partial tiles are added in wave order through LDS (fixed order: run-to-run identical).
struct WgradArgs {
    int64_t n, rows_per_block, pstride;
    int n_in, n_out;
    const float *H, *DZ;
    float *part;
};

template <int MT, int NT2>
__global__ void __launch_bounds__(256, 2) k_wgrad_f32(const WgradArgs p)
{
    using namespace gnn_fused_dev;
    __shared__ float red[3][1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * p.rows_per_block, c1 = c0 + p.rows_per_block < p.n ? c0 + p.rows_per_block : p.n;
    const int64_t quarter = ((c1 - c0 + 3) / 4 + 1) & ~(int64_t)1;                 // even: K-steps are row pairs
    const int64_t r0 = c0 + wave * quarter, r1 = r0 + quarter < c1 ? r0 + quarter : c1;
    const int nt0 = blockIdx.y * NT2;
    f32x16 acc[MT][NT2];
#pragma unroll
    for (int a = 0; a < MT; ++a) zero_acc<NT2>(acc[a]);
    constexpr int PF = 4;
    float av[PF][MT], bv[PF][NT2];
    auto load = [&](int slot, int64_t r) {
        const int64_t rr = r + kh;
        const bool in = rr < r1;
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            const int hf = 32 * a + m;
            av[slot][a] = in ? (hf < p.n_in ? gload1(p.H + rr * p.n_in + hf) : (hf == p.n_in ? 1.0f : 0.0f)) : 0.0f;
        }
#pragma unroll
        for (int b = 0; b < NT2; ++b) {
            const int zf = 32 * (nt0 + b) + m;
            bv[slot][b] = (in && zf < p.n_out) ? gload1(p.DZ + rr * p.n_out + zf) : 0.0f;
        }
    };
#pragma unroll
    for (int s = 0; s < PF; ++s) load(s, r0 + 2 * s);
    for (int64_t r = r0; r < r1; r += 2 * PF) {
#pragma unroll
        for (int s = 0; s < PF; ++s) {
            float a_[MT], b_[NT2];
#pragma unroll
            for (int a = 0; a < MT; ++a) a_[a] = av[s][a];
#pragma unroll
            for (int b = 0; b < NT2; ++b) b_[b] = bv[s][b];
            load(s, r + 2 * (s + PF));                                               // rows past r1 load zeros
#pragma unroll
            for (int a = 0; a < MT; ++a)
#pragma unroll
                for (int b = 0; b < NT2; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_[a], b_[b], acc[a][b], 0, 0, 0);
        }
    }
    // D[row hf = 32 a + (r & 3) + 8 (r >> 2) + 4 kh][col zf = 32 (nt0 + b) + m]
    float *out = p.part + (size_t)blockIdx.x * p.pstride;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT2; ++b) {
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][r * 64 + lane] = acc[a][b][r];
            }
            __syncthreads();
            if (wave == 0) {
                const int zf = 32 * (nt0 + b) + m;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[a][b][r];
                    v = v + red[0][r * 64 + lane]; v = v + red[1][r * 64 + lane]; v = v + red[2][r * 64 + lane];
                    const int hf = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (hf <= p.n_in && zf < p.n_out) out[(size_t)hf * p.n_out + zf] = v;
                }
            }
            __syncthreads();
        }
}

// The same partials in the split arithmetic of the dense layers (round 5): every fp32 operand cut into three exact bf16 pieces, six piece
// products per term on v_mfma_f32_32x32x16_bf16, fp32 accumulation (error per product <= 3 * 2^-24: fp32-class, run-to-run identical).  Rows are
// the K dimension, 16 per step: lane (m, k half) takes H[r + 8 k half + i][32 a + m], i < 8 - eight coalesced 128-byte row pieces per operand
// tile, no transposition - and cuts them in registers.  What the round-3 experiment of this (below, 0.92 ms against 0.42) lacked: its 5 x 2
// accumulator tiles (160 registers) left no room to have the next step's rows in flight.  Here a block is EIGHT waves: wave w owns d z tile
// w % NT for ALL tiles of [H | 1] (MT x 16 accumulator registers) on rows part w / NT of the chunk, the next step's 8 (MT + 1) row pieces are
// requested before the current step's products, and the 8 / NT partial tiles of an output are added in part order through LDS.
template <int MT, int NT>
__global__ void __launch_bounds__(512, 2) k_wgrad_bf(const WgradArgs p)
{
    using namespace gnn_fused_dev;
    __shared__ float red[8][1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, kh = lane >> 5;
    constexpr int NP = 8 / NT;                                                     // row parts of a chunk
    const int b = wave % NT, part = wave / NT;
    const int64_t c0 = (int64_t)blockIdx.x * p.rows_per_block, c1 = c0 + p.rows_per_block < p.n ? c0 + p.rows_per_block : p.n;
    const int64_t span = ((c1 - c0 + NP - 1) / NP + 15) & ~(int64_t)15;             // whole K = 16 steps
    const int64_t r0 = c0 + part * span, r1 = r0 + span < c1 ? r0 + span : c1;
    const int zf = 32 * b + m;
    const bool zok = zf < p.n_out;
    f32x16 acc[MT];
    zero_acc<MT>(acc);
    // ONE register set: a tile's eight row pieces are requested again for the NEXT step as soon as this step has cut them into pieces, i.e. a
    // whole step (6 MT MFMAs) ahead of their use (two sets, loaded a step ahead as a block: 28 registers spilled at MT = 5)
    float hv[MT][8], zv[8];
    auto load_z = [&](int64_t r) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = r + 8 * kh + i;
            zv[i] = (rr < r1 && zok) ? gload1(p.DZ + rr * p.n_out + zf) : 0.0f;
        }
    };
    auto load_h = [&](int a, int64_t r) {
        const int hf = 32 * a + m;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = r + 8 * kh + i;
            hv[a][i] = rr < r1 ? (hf < p.n_in ? gload1(p.H + rr * p.n_in + hf) : (hf == p.n_in ? 1.0f : 0.0f)) : 0.0f;
        }
    };
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    if (r0 < r1) {
        load_z(r0);
#pragma unroll
        for (int a = 0; a < MT; ++a) load_h(a, r0);
    }
    for (int64_t r = r0; r < r1; r += 16) {
        v4i pb[3];
        split8(zv, pb[0], pb[1], pb[2]);
        load_z(r + 16);                                                             // (rows past r1 load zeros: no guard around the requests)
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            v4i pa[3];
            split8(hv[a], pa[0], pa[1], pa[2]);
            load_h(a, r + 16);
#pragma unroll
            for (int term = 0; term < 6; ++term) acc[a] = mfma_bf16(pa[PA[term]], pb[PB[term]], acc[a]);
        }
    }
    // D[row hf = 32 a + (r & 3) + 8 (r >> 2) + 4 kh][col zf]
    float *out = p.part + (size_t)blockIdx.x * p.pstride;
#pragma unroll
    for (int a = 0; a < MT; ++a) {
        if (part > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[wave][r * 64 + lane] = acc[a][r];
        }
        __syncthreads();
        if (part == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[a][r];
#pragma unroll
                for (int q = 1; q < NP; ++q) v = v + red[b + NT * q][r * 64 + lane];
                const int hf = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                if (hf <= p.n_in && zok) out[(size_t)hf * p.n_out + zf] = v;
            }
        }
        __syncthreads();
    }
}

#ifdef GNN_DIAG
// EXPERIMENT (diagnostic build, GNN_TRAIN_WGRAD_SPLIT=1; round 3): the same partials in split arithmetic (three exact bf16 pieces per
// operand, six piece products on v_mfma_f32_32x32x16_bf16): rows are the K dimension, 16 per step - lane (m, k half) holds
// H[r + 8 k half + i][32 a + m], i < 8, eight coalesced row pieces per operand tile, cut into pieces in registers.  60 bf16 MFMAs
// (1,920 matrix-pipe cycles) per 16 rows instead of 80 f32 MFMAs (5,120) - and measured SLOWER: 0.92 ms against 0.42 ms per
// 1 M x 129 x 128 gradient (256 VGPRs + 33 spilled; fifty-six dependent row-piece loads per K-step).  Correct (the training tests pass with it).
template <int MT, int NT2>
__global__ void __launch_bounds__(256, 2) k_wgrad_split(const WgradArgs p)
{
    using namespace gnn_fused_dev;
    __shared__ float red[3][1024];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int m = lane & 31, kh = lane >> 5;
    const int64_t c0 = (int64_t)blockIdx.x * p.rows_per_block, c1 = c0 + p.rows_per_block < p.n ? c0 + p.rows_per_block : p.n;
    const int64_t quarter = ((c1 - c0 + 3) / 4 + 15) & ~(int64_t)15;                // whole K = 16 steps
    const int64_t r0 = c0 + wave * quarter, r1 = r0 + quarter < c1 ? r0 + quarter : c1;
    const int nt0 = blockIdx.y * NT2;
    f32x16 acc[MT][NT2];
#pragma unroll
    for (int a = 0; a < MT; ++a) zero_acc<NT2>(acc[a]);
    constexpr int PA[6] = {0, 2, 1, 0, 1, 0}, PB[6] = {2, 0, 1, 1, 0, 0};
    auto load_a = [&](int a, int64_t r, float (&v)[8]) {
        const int hf = 32 * a + m;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int64_t rr = r + 8 * kh + i;
            v[i] = rr < r1 ? (hf < p.n_in ? gload1(p.H + rr * p.n_in + hf) : (hf == p.n_in ? 1.0f : 0.0f)) : 0.0f;
        }
    };
    for (int64_t r = r0; r < r1; r += 16) {
        v4i pb[NT2][3];
#pragma unroll
        for (int b = 0; b < NT2; ++b) {
            const int zf = 32 * (nt0 + b) + m;
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int64_t rr = r + 8 * kh + i;
                v[i] = (rr < r1 && zf < p.n_out) ? gload1(p.DZ + rr * p.n_out + zf) : 0.0f;
            }
            split8(v, pb[b][0], pb[b][1], pb[b][2]);
        }
        float va[8], vn[8];
        load_a(0, r, va);
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            if (a + 1 < MT) load_a(a + 1, r, vn);                                    // the next tile's rows are on their way during these MFMAs
            v4i pa[3];
            split8(va, pa[0], pa[1], pa[2]);
#pragma unroll
            for (int term = 0; term < 6; ++term)
#pragma unroll
                for (int b = 0; b < NT2; ++b) acc[a][b] = mfma_bf16(pa[PA[term]], pb[b][PB[term]], acc[a][b]);
#pragma unroll
            for (int i = 0; i < 8; ++i) va[i] = vn[i];
        }
    }
    float *out = p.part + (size_t)blockIdx.x * p.pstride;
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int b = 0; b < NT2; ++b) {
            if (wave > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[wave - 1][r * 64 + lane] = acc[a][b][r];
            }
            __syncthreads();
            if (wave == 0) {
                const int zf = 32 * (nt0 + b) + m;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[a][b][r];
                    v = v + red[0][r * 64 + lane]; v = v + red[1][r * 64 + lane]; v = v + red[2][r * 64 + lane];
                    const int hf = 32 * a + (r & 3) + 8 * (r >> 2) + 4 * kh;
                    if (hf <= p.n_in && zf < p.n_out) out[(size_t)hf * p.n_out + zf] = v;
                }
            }
            __syncthreads();
        }
}

#endif

int launch_wgrad_f32(hipStream_t st, int64_t n, int64_t rpb, int parts, int64_t pstride, int n_in, int n_out, const float *H, const float *DZ, float *part)
{
    WgradArgs p{n, rpb, pstride, n_in, n_out, H, DZ, part};
    const int mt = (n_in + 1 + 31) / 32, nt = (n_out + 31) / 32;
    const int nt2 = nt >= 2 ? 2 : 1;
    {   // split-bf16 form (k_wgrad_bf) when the d z tiles divide the eight waves of a block; the f32-MFMA form (k_wgrad_f32) otherwise
        bool bf = nt == 1 || nt == 2 || nt == 4;
#ifdef GNN_DIAG
        static const bool bf_off = getenv("GNN_TRAIN_WGRAD_BF") && atoi(getenv("GNN_TRAIN_WGRAD_BF")) == 0;
        if (bf_off) bf = false;
#endif
        if (bf) {
#define GNN_WGB_CASE(M_, N_) if (mt == M_ && nt == N_) { hipLaunchKernelGGL((k_wgrad_bf<M_, N_>), dim3((unsigned)parts), 512, 0, st, p); HIPCHK(hipGetLastError()); return GNN_OK; }
            GNN_WGB_CASE(3, 1) GNN_WGB_CASE(3, 2) GNN_WGB_CASE(3, 4) GNN_WGB_CASE(4, 1) GNN_WGB_CASE(4, 2) GNN_WGB_CASE(4, 4) GNN_WGB_CASE(5, 1) GNN_WGB_CASE(5, 2) GNN_WGB_CASE(5, 4)
#undef GNN_WGB_CASE
        }
    }
    const dim3 grid((unsigned)parts, (unsigned)((nt + nt2 - 1) / nt2));
#ifdef GNN_DIAG
    static const bool split = getenv("GNN_TRAIN_WGRAD_SPLIT") != nullptr;
#define GNN_WG_LAUNCH(M_, N_) if (split) hipLaunchKernelGGL((k_wgrad_split<M_, N_>), grid, 256, 0, st, p); else hipLaunchKernelGGL((k_wgrad_f32<M_, N_>), grid, 256, 0, st, p);
#else
#define GNN_WG_LAUNCH(M_, N_) hipLaunchKernelGGL((k_wgrad_f32<M_, N_>), grid, 256, 0, st, p);
#endif
#define GNN_WG_CASE(M_, N_)                                                                         \
    if (mt == M_ && nt2 == N_) {                                                                    \
        GNN_WG_LAUNCH(M_, N_)                                                                       \
        HIPCHK(hipGetLastError());                                                                  \
        return GNN_OK;                                                                              \
    }
    GNN_WG_CASE(3, 1) GNN_WG_CASE(3, 2) GNN_WG_CASE(4, 1) GNN_WG_CASE(4, 2) GNN_WG_CASE(5, 1) GNN_WG_CASE(5, 2)
#undef GNN_WG_CASE
#undef GNN_WG_LAUNCH
    return gnn_fail(GNN_ERR_UNSUPPORTED, "no matrix-core weight-gradient instantiation for %d x %d tiles", mt, nt2);
}
inline bool tg_wgrad_covers(int n_in, int n_out) { const int mt = (n_in + 1 + 31) / 32; return mt >= 3 && mt <= 5 && n_out >= 32; }

// The concat of one body (reference GNN/GNN.py:223-239) in one pass: [state | node labels | aggregated states | aggregated labels |
// aggregated arc labels].  Everything but the state columns and their aggregate is loop-invariant and comes from the template.
// Dropout in front of the first Dense layer (rate != 0) is applied on the way out.  The thread of column 0 also evaluates the
// while-condition of THIS body for its node (reference GNN/GNN.py:202-220: condition(state, state_old), ascending-feature sums as
// k_check; so == NULL: ones) and raises the body's gate.
// state: row 0 of the state REPLICA (the sources of the arcs are replica rows); own: the first OWNED row of it (== state on one GPU)
__global__ void __launch_bounds__(256) k_train_input(int64_t n, int in_s, int Ds, int c_aggs, const float *__restrict__ tmpl, const float *__restrict__ state,
                                                     const float *__restrict__ own,
                                                     const int32_t *__restrict__ indptr, const int32_t *__restrict__ adj_src,
                                                     const float *__restrict__ adj_w, float rate, const uint8_t *mask_in, uint64_t seed, uint8_t *keep,
                                                     float *__restrict__ inp, const float *__restrict__ so, float thr, int *flag)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int f = 0;
    if (i < n * in_s) {
        // (a 64-bit division is some forty instructions: the 32-bit one when the matrix has fewer than 2^31 elements)
        const int64_t r = n * in_s < ((int64_t)1 << 31) ? (int64_t)((unsigned)i / (unsigned)in_s) : i / in_s;
        const int c = (int)(i - r * in_s);
        float v;
        bool skip = false;                         // Ds % 4 == 0: the thread of every fourth aggregate column gathers and writes four
        if (c < Ds) v = own[r * Ds + c];
        else if (c >= c_aggs && c < c_aggs + Ds) {
            const int cc = c - c_aggs;
            v = 0.0f;
            if ((Ds & 3) == 0 && rate == 0.0f) {
                skip = true;
                if ((cc & 3) == 0) {
                    float4 a4 = {0.0f, 0.0f, 0.0f, 0.0f};
                    const int32_t e1 = indptr[r + 1];
                    for (int32_t e = indptr[r]; e < e1; e += 4) {          // four arcs per step: their loads are in flight together
                        float w[4];
                        float4 x[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) {
                            const bool in = e + u < e1;
                            w[u] = in ? adj_w[e + u] : 0.0f;
                            x[u] = in ? *reinterpret_cast<const float4 *>(state + (int64_t)adj_src[e + u] * Ds + cc) : float4{0.0f, 0.0f, 0.0f, 0.0f};
                        }
#pragma unroll
                        for (int u = 0; u < 4; ++u)
                            if (e + u < e1) {
                                a4.x = __builtin_fmaf(w[u], x[u].x, a4.x); a4.y = __builtin_fmaf(w[u], x[u].y, a4.y);
                                a4.z = __builtin_fmaf(w[u], x[u].z, a4.z); a4.w = __builtin_fmaf(w[u], x[u].w, a4.w);
                            }
                    }
                    inp[i] = a4.x; inp[i + 1] = a4.y; inp[i + 2] = a4.z; inp[i + 3] = a4.w;
                }
            } else {
                const int32_t e1 = indptr[r + 1];
                for (int32_t e = indptr[r]; e < e1; e += 4) {
                    float w[4], x[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const bool in = e + u < e1;
                        w[u] = in ? adj_w[e + u] : 0.0f;
                        x[u] = in ? state[(int64_t)adj_src[e + u] * Ds + cc] : 0.0f;
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (e + u < e1) v = __builtin_fmaf(w[u], x[u], v);
                }
            }
        } else
            v = tmpl[i];
        if (rate != 0.0f) {
            const float rr = fabsf(rate);
            uint8_t kp;
            if (mask_in) kp = mask_in[i] != 0;
            else kp = ((mix64(seed ^ mix64((uint64_t)i)) >> 40) * (1.0f / 16777216.0f)) >= rr;
            keep[i] = kp;
            if (rate < 0.0f) {
                float a, b, ap;
                alpha_dropout_coeffs(rr, &a, &b, &ap);
                v = a * (kp ? v : ap) + b;
            } else
                v = kp ? v / (1.0f - rate) : 0.0f;
        }
        if (!skip) inp[i] = v;
        if (c == 0) {
            float dist = 0.0f, nrm = 0.0f;
            if ((Ds & 3) == 0) {
#pragma unroll 4
                for (int q = 0; q < Ds; q += 4) {
                    const float4 sv = *reinterpret_cast<const float4 *>(own + r * Ds + q);
                    const float4 ov = so ? *reinterpret_cast<const float4 *>(so + r * Ds + q) : float4{1.0f, 1.0f, 1.0f, 1.0f};
                    const float d0 = sv.x - ov.x, d1 = sv.y - ov.y, d2 = sv.z - ov.z, d3 = sv.w - ov.w;
                    dist = dist + d0 * d0; nrm = nrm + ov.x * ov.x;
                    dist = dist + d1 * d1; nrm = nrm + ov.y * ov.y;
                    dist = dist + d2 * d2; nrm = nrm + ov.z * ov.z;
                    dist = dist + d3 * d3; nrm = nrm + ov.w * ov.w;
                }
            } else
                for (int q = 0; q < Ds; ++q) {
                    const float o = so ? so[r * Ds + q] : 1.0f;
                    const float df = own[r * Ds + q] - o;
                    dist = dist + df * df;
                    nrm = nrm + o * o;
                }
            f = sqrtf(dist) > thr * sqrtf(nrm);
        }
    }
    if (__any(f) && (threadIdx.x & 63) == 0) gnn_flag_raise(flag);
}

// The same concat for many rows without Dropout in front of the first layer (state width a multiple of 4, <= 64): 16 lanes per row,
// each gathering four aggregate columns (four arcs in flight, the fmaf chain in stored order) and copying four state columns and the
// template columns; the while-condition of the body is evaluated by k_check (gnn_launch_check: same ascending-feature sums) beside it.
// k_train_input gave the condition to the thread of column 0 - a 64-step chain that the other 63 lanes of its wave waited for - and ran
// one thread per element: 3.2 ms per body at 1 M rows x 135 columns, this one about a quarter of that (profiles/r03_train_c3.txt).
__global__ void __launch_bounds__(256) k_train_input_rows(int64_t n, int in_s, int Ds, int c_aggs, const float *__restrict__ tmpl, const float *__restrict__ state,
                                                          const float *__restrict__ own_rows,
                                                          const int32_t *__restrict__ indptr, const int32_t *__restrict__ adj_src,
                                                          const float *__restrict__ adj_w, float *__restrict__ inp)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> 4;
    const int j = (int)(t & 15);
    if (r >= n) return;
    float *row = inp + r * in_s;
    const int cc = 4 * j;
    if (cc < Ds) {
        const float4 own = *reinterpret_cast<const float4 *>(own_rows + r * Ds + cc);
        float4 a4 = {0.0f, 0.0f, 0.0f, 0.0f};
        const int32_t e1 = indptr[r + 1];
        for (int32_t e = indptr[r]; e < e1; e += 4) {
            float w[4];
            float4 x[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = e + u < e1;
                w[u] = in ? adj_w[e + u] : 0.0f;
                x[u] = in ? *reinterpret_cast<const float4 *>(state + (int64_t)adj_src[e + u] * Ds + cc) : float4{0.0f, 0.0f, 0.0f, 0.0f};
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u < e1) {
                    a4.x = __builtin_fmaf(w[u], x[u].x, a4.x); a4.y = __builtin_fmaf(w[u], x[u].y, a4.y);
                    a4.z = __builtin_fmaf(w[u], x[u].z, a4.z); a4.w = __builtin_fmaf(w[u], x[u].w, a4.w);
                }
        }
        row[cc] = own.x; row[cc + 1] = own.y; row[cc + 2] = own.z; row[cc + 3] = own.w;
        float *ag = row + c_aggs + cc;
        ag[0] = a4.x; ag[1] = a4.y; ag[2] = a4.z; ag[3] = a4.w;
    }
    // template columns: [Ds, c_aggs) and [c_aggs + Ds, in_s)
    const int n1 = c_aggs - Ds, nt = n1 + (in_s - c_aggs - Ds);
    for (int q = j; q < nt; q += 16) {
        const int c = q < n1 ? Ds + q : c_aggs + Ds + (q - n1);
        row[c] = tmpl[r * in_s + c];
    }
}

// End of one body of the backward pass in one launch.  Blocks < sg_blocks: aggregated_states = Adjacency^T . state  =>
// d state[r] = d inp[r, :Ds] + sum over arcs (r -> dst) of w * d inp[dst, c_aggs:] (own-state columns of the concat + the transposed
// aggregation over the by-source CSR).  The other blocks: the net's gradient vector += this call's chunk partials (sum_parts_block).
__global__ void __launch_bounds__(256) k_state_grad_sum(int sg_blocks, int64_t n, int Ds, int in_s, int c_aggs, const float *__restrict__ d_inp,
                                                        const int32_t *__restrict__ sip, const int32_t *__restrict__ sdst, const float *__restrict__ sw,
                                                        float *__restrict__ d_state, int parts, int64_t count, const float *part, float *out)
{
    __shared__ float sp[256];
    if ((int)blockIdx.x >= sg_blocks) {
        sum_parts_block(blockIdx.x - sg_blocks, parts, count, part, out, sp);
        return;
    }
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * Ds) return;
    const int64_t r = n * Ds < ((int64_t)1 << 31) ? (int64_t)((unsigned)t / (unsigned)Ds) : t / Ds;
    const int c = (int)(t - r * Ds);
    float acc = 0.0f;
    const int32_t e1 = sip[r + 1];
    for (int32_t e = sip[r]; e < e1; e += 4) {                  // four arcs per step: their loads are in flight together
        float w[4], x[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = e + u < e1;
            w[u] = in ? sw[e + u] : 0.0f;
            x[u] = in ? d_inp[(int64_t)sdst[e + u] * in_s + c_aggs + c] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) if (e + u < e1) acc = __builtin_fmaf(w[u], x[u], acc);
    }
    d_state[t] = d_inp[r * in_s + c] + acc;
}


// The transposed aggregation of k_state_grad_sum for many rows (state width a multiple of 4, <= 64): 16 lanes per source row, four columns
// per lane, four arcs in flight per lane (sixteen dependent-free loads), same fmaf chain per element; the sum of the chunk partials runs as
// its own launch then.  (One thread per element: 1.14 ms per body at 1 M rows x 64, profiles/r03_train_c3.txt.)
__global__ void __launch_bounds__(256) k_state_grad_rows(int64_t n, int Ds, int in_s, int c_aggs, const float *__restrict__ d_inp, const int32_t *__restrict__ sip,
                                                         const int32_t *__restrict__ sdst, const float *__restrict__ sw, float *__restrict__ d_state)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> 4;
    const int cc = 4 * (int)(t & 15);
    if (r >= n || cc >= Ds) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int32_t e1 = sip[r + 1];
    for (int32_t e = sip[r]; e < e1; e += 4) {
        float w[4], x[4][4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const bool in = e + u < e1;
            w[u] = in ? sw[e + u] : 0.0f;
            const float *q = d_inp + (int64_t)(in ? sdst[e + u] : 0) * in_s + c_aggs + cc;
#pragma unroll
            for (int v = 0; v < 4; ++v) x[u][v] = in ? q[v] : 0.0f;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (e + u < e1) {
#pragma unroll
                for (int v = 0; v < 4; ++v) acc[v] = __builtin_fmaf(w[u], x[u][v], acc[v]);
            }
    }
    const float *own = d_inp + r * in_s + cc;
    *reinterpret_cast<float4 *>(d_state + r * Ds + cc) = float4{own[0] + acc[0], own[1] + acc[1], own[2] + acc[2], own[3] + acc[3]};
}

// The same from the aligned copy k_bwd3_split leaves (dsg [n, 2 Ds] = [d inp[:, :Ds] | d inp[:, c_aggs : c_aggs + Ds]]): 16-byte loads, eight arcs in
// flight per lane, same fmaf chain per element.
__global__ void __launch_bounds__(256) k_state_grad_rows_al(int64_t n, int Ds, const float *__restrict__ dsg, const int32_t *__restrict__ sip,
                                                            const int32_t *__restrict__ sdst, const float *__restrict__ sw, float *__restrict__ d_state)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t r = t >> 4;
    const int cc = 4 * (int)(t & 15);
    if (r >= n || cc >= Ds) return;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    const int32_t e1 = sip[r + 1];
    for (int32_t e = sip[r]; e < e1; e += 8) {
        float w[8];
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int32_t ee = e + u < e1 ? e + u : e;                   // clamp: a real entry, result unused
            w[u] = sw[ee];
            x[u] = *reinterpret_cast<const float4 *>(dsg + (int64_t)sdst[ee] * 2 * Ds + Ds + cc);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (e + u < e1) {
                acc[0] = __builtin_fmaf(w[u], x[u].x, acc[0]); acc[1] = __builtin_fmaf(w[u], x[u].y, acc[1]);
                acc[2] = __builtin_fmaf(w[u], x[u].z, acc[2]); acc[3] = __builtin_fmaf(w[u], x[u].w, acc[3]);
            }
    }
    const float4 own = *reinterpret_cast<const float4 *>(dsg + r * 2 * Ds + cc);
    *reinterpret_cast<float4 *>(d_state + r * Ds + cc) = float4{own.x + acc[0], own.y + acc[1], own.z + acc[2], own.w + acc[3]};
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
        if (gnn_dev_malloc((void **)&s.p, s.size) != hipSuccess) return nullptr;
        slabs.push_back(s);
        cur = slabs.size() - 1;
        off = bytes;
        return s.p;
    }
    // optimizer armed for the next gnn_loop_train_step (gnn_loop_arm_optimizer): applied behind the backward pass, before the
    // step's only wait for the device
    struct { bool armed = false; int kind = 0; float h[4] = {0, 0, 0, 0}; bool mean = false; float mom_s = 0.99f, mom_o = 0.99f; } opt;
    // pinned host words for the results the host waits for (iteration gates, loss partials)
    void *pinned = nullptr;
    size_t pinned_bytes = 0;
    void *host(size_t bytes)
    {
        if (bytes > pinned_bytes) {
            if (pinned) (void)hipHostFree(pinned);
            pinned = nullptr; pinned_bytes = 0;
            if (hipHostMalloc(&pinned, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
            pinned_bytes = bytes;
        }
        return pinned;
    }
    // pinned staging for the small per-step uploads (targets, sample weights, NodeGraph CSR): packed by the host, one transfer
    void *staging = nullptr;
    size_t staging_bytes = 0;
    void *stage(size_t bytes)
    {
        if (bytes > staging_bytes) {
            if (staging) (void)hipHostFree(staging);
            staging = nullptr; staging_bytes = 0;
            if (hipHostMalloc(&staging, bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
            staging_bytes = bytes;
        }
        return staging;
    }
    ~TrainArena()
    {
        for (Slab &s : slabs) (void)hipFree(s.p);
        if (pinned) (void)hipHostFree(pinned);
        if (staging) (void)hipHostFree(staging);
    }
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
    float *grads = nullptr;       // flat: dW1, db1, ..., dgamma, dbeta
    std::vector<size_t> g_off;
    size_t g_total = 0;
    float *part = nullptr;        // [chunks of the rows][g_total]: the partial gradients of ONE net_backward call
    int64_t part_rows = -1;
    float *stats_all = nullptr;   // [max forward calls][2 F]: batch mean | biased batch variance of every BatchNormalization call, in call order
    int calls = 0, max_calls = 0;
};

inline unsigned elementwise_grid(int64_t total) { return (unsigned)std::min<int64_t>(std::max<int64_t>(1, (total + 255) / 256), 2048); }

// floats of zero-initialised memory a Net needs: the gradient vector and the BatchNormalization statistics of every call
inline size_t net_zero_floats(const gnn_mlp *m, int max_calls)
{
    size_t t = 0;
    for (int l = 0; l < m->n_layers; ++l) t += (size_t)m->dims[l] * m->dims[l + 1] + (size_t)m->dims[l + 1];
    if (m->has_bn) t += (size_t)2 * m->dims.back() + (size_t)std::max(1, max_calls) * 2 * m->dims.back();
    return (t + 63) & ~(size_t)63;
}

// zero_mem: net_zero_floats() floats the caller has zeroed (one memset for everything a step needs zeroed)
int net_setup(hipStream_t st, Buf &buf, Net &net, const gnn_mlp *m, const float *rates, const float *bn_gamma_beta_host, int max_calls, float *zero_mem)
{
    net.m = m;
    net.max_calls = max_calls;
    const int L = m->n_layers;
    net.rate.assign(rates, rates + L + 1);
    net.WT.assign(L, nullptr);
    size_t off = 0;
    int rc;
    for (int l = 0; l < L; ++l) {
        const int ni = m->dims[l], no = m->dims[l + 1];
        if ((rc = buf.get(&net.WT[l], (size_t)ni * no))) return rc;
        hipLaunchKernelGGL(k_transpose, cdiv((int64_t)ni * no, 256), 256, 0, st, ni, no, m->W[l], net.WT[l]);
        HIPCHK(hipGetLastError());
        net.g_off.push_back(off); off += (size_t)ni * no;
        net.g_off.push_back(off); off += (size_t)no;
    }
    if (m->has_bn) {
        const int F = m->dims.back();
        // gamma | beta: the caller's arrays, or (NULL) the MLP's own device copy (the one the device-side optimizer updates)
        if (bn_gamma_beta_host) {
            if ((rc = buf.get(&net.gamma, (size_t)2 * F))) return rc;
            net.beta = net.gamma + F;
            HIPCHK(hipMemcpyAsync(net.gamma, bn_gamma_beta_host, sizeof(float) * 2 * F, hipMemcpyHostToDevice, st));
        } else { net.gamma = m->bn_raw; net.beta = m->bn_raw + F; }
        net.g_off.push_back(off); off += F;
        net.g_off.push_back(off); off += F;
        net.stats_all = zero_mem + off;
    }
    net.g_total = off;
    net.grads = zero_mem;
    return GNN_OK;
}

// training-mode forward of one Sequential on n rows (x: [n, dims[0]]); *y_out: [n, dims.back()].  keep0 != NULL: the Dropout
// in front of the first Dense layer has been applied by the producer of x (k_train_input), its mask is keep0.
// few rows, no Dropout between the layers, softmax at most as the last activation: all Dense layers in one launch (k_mlp_fwd)?
static bool mlp_small_fused_ok(const Net &net, int64_t n, size_t *lds_out = nullptr, int *maxpad_out = nullptr)
{
    const gnn_mlp *m = net.m;
    const int L = m->n_layers;
    if (!(L >= 2 && L <= GNN_FUSED_MAXL + 1 && n > 0 && !tg_many_rows(n))) return false;
    int maxpad = (m->dims[0] + 3) & ~3;
    for (int q = 1; q <= L; ++q) {
        if (q < L && (net.rate[q] != 0.0f || m->acts[q - 1] == GNN_ACT_SOFTMAX)) return false;
        maxpad = std::max(maxpad, (m->dims[q] + 3) & ~3);
    }
    const size_t lds = sizeof(float) * ((size_t)2 * 8 * maxpad + (size_t)256 * 8);
#ifdef GNN_DIAG
    static const bool off = getenv("GNN_TRAIN_MLP_FUSED") && atoi(getenv("GNN_TRAIN_MLP_FUSED")) == 0;
    if (off) return false;
#endif
    if (lds > 64 * 1024) return false;
    if (lds_out) *lds_out = lds;
    if (maxpad_out) *maxpad_out = maxpad;
    return true;
}

// comm != NULL (sharded forward, one process per rank): the BatchNormalization statistics are those of the rows of ALL ranks.
// build != NULL (only where mlp_small_fused_ok and no Dropout in front of the first layer): x has NOT been filled - k_mlp_fwd builds the
// concat rows itself (and writes them to x for the backward pass); the build fields of *build say from what.
int net_forward(hipStream_t st, Buf &buf, Net &net, int64_t n, float *x, uint8_t *keep0, const uint8_t *masks, uint64_t seed, NetCache &c,
                float **y_out, gnn_comm *comm = nullptr, const MlpFwd *build = nullptr)
{
    const gnn_mlp *m = net.m;
    const int L = m->n_layers;
    c.n = n;
    c.hin.assign(L, nullptr); c.a.assign(L, nullptr); c.keep.assign(L + 1, nullptr);
    float *h = x;
    size_t mask_off = 0;
    int rc;
    int l_start = 0;
    // a 3-layer net without Dropout behind its first layer, on many rows: the three Dense layers in one pass (k_fwd3_split)
    bool fuse3 = n > 0 && tg_many_rows(n) && fwd3_covers(m) && net.rate[1] == 0.0f && net.rate[2] == 0.0f && (net.rate[0] == 0.0f || keep0);
#ifdef GNN_DIAG
    static const bool fuse_off = getenv("GNN_TRAIN_FWD3") && atoi(getenv("GNN_TRAIN_FWD3")) == 0;
    if (fuse_off) fuse3 = false;
#endif
    if (fuse3) {
        if (net.rate[0] != 0.0f) { c.keep[0] = keep0; mask_off += (size_t)n * m->dims[0]; }
        for (int l = 0; l < 3; ++l)
            if ((rc = buf.get(&c.a[l], (size_t)n * m->dims[l + 1]))) return rc;
        c.hin[0] = x; c.hin[1] = c.a[0]; c.hin[2] = c.a[1];
        if ((rc = launch_fwd3(st, buf, m, n, x, c.a[0], c.a[1], c.a[2]))) return rc;
        h = c.a[2];
        l_start = L;
    }
    for (int l = l_start; l <= L; ++l) {
        const int width = m->dims[l];
        if (net.rate[l] != 0.0f) {
            if (l == 0 && keep0) c.keep[0] = keep0;
            else {
                float *hd = nullptr;
                if ((rc = buf.get(&hd, (size_t)n * width)) || (rc = buf.get(&c.keep[l], (size_t)n * width))) return rc;
                if (n > 0) {
                    hipLaunchKernelGGL(k_dropout_fwd, cdiv(n * width, 256), 256, 0, st, n * width, h, masks ? masks + mask_off : nullptr, net.rate[l],
                                       seed + 0x9E37ull * (uint64_t)(l + 1), c.keep[l], hd);
                    HIPCHK(hipGetLastError());
                }
                h = hd;
            }
            mask_off += (size_t)n * width;
        }
        if (l == L) break;
        // few rows, no Dropout between the layers: every Dense layer of the net in one launch (k_mlp_fwd)
        size_t lds_small = 0;
        int maxpad = 0;
        if (l == 0 && mlp_small_fused_ok(net, n, &lds_small, &maxpad)) {
            constexpr int R = 8;
            MlpFwd p{};
            if (build) { p = *build; p.build = 1; p.X_out = h; }
            p.n = n; p.L = L; p.maxpad = maxpad; p.X = h;
            for (int q = 0; q <= L; ++q) { p.dims[q] = m->dims[q]; p.pad[q] = (m->dims[q] + 3) & ~3; }
            for (int q = 0; q < L; ++q) {
                if ((rc = buf.get(&c.a[q], (size_t)n * m->dims[q + 1]))) return rc;
                p.cshift[q] = dense_cshift(m->dims[q + 1]); p.act[q] = m->acts[q]; p.W[q] = m->W[q]; p.b[q] = m->b[q]; p.Y[q] = c.a[q];
                c.hin[q] = q == 0 ? h : c.a[q - 1];
            }
            hipLaunchKernelGGL((k_mlp_fwd<R>), cdiv(n, R), 256, lds_small, st, p);
            HIPCHK(hipGetLastError());
            if (m->acts[L - 1] == GNN_ACT_SOFTMAX) {
                hipLaunchKernelGGL(k_act_fwd, cdiv(n, 256), 256, 0, st, n, m->dims[L], c.a[L - 1], m->acts[L - 1], c.a[L - 1]);
                HIPCHK(hipGetLastError());
            }
            h = c.a[L - 1];
            l = L - 1;               // (the loop goes on with index L: the Dropout in front of BatchNormalization, if any)
            continue;
        }
        if (build) return gnn_fail(GNN_ERR_STATE, "internal: the input rows were left to a fused forward that did not run");
        const int no = m->dims[l + 1];
        c.hin[l] = h;
        if ((rc = buf.get(&c.a[l], (size_t)n * no))) return rc;
        const bool sm = m->acts[l] == GNN_ACT_SOFTMAX;          // softmax needs the whole row: separate pass, in place
        if (n > 0 && tg_wide(width, no) && !sm && tg_many_rows(n)) {      // wide layer on many rows: matrix cores
            if ((rc = launch_gemm_f32(st, buf, n, width, no, h, m->W[l], m->b[l], m->acts[l], 0, nullptr, 0.0f, nullptr, c.a[l]))) return rc;
        } else if (n > 0) {
            constexpr int R = 8;
            const int ni_pad = (width + 3) & ~3;
            const size_t lds = dense_lds_bytes(R, ni_pad);
            if (lds > 64 * 1024) return gnn_fail(GNN_ERR_UNSUPPORTED, "layer input width %d too large", width);
            hipLaunchKernelGGL((k_dense_fwd<R>), cdiv(n, R), 256, lds, st, n, width, ni_pad, no, dense_cshift(no), h, m->W[l], m->b[l],
                               sm ? GNN_ACT_LINEAR : m->acts[l], c.a[l]);
            HIPCHK(hipGetLastError());
        }
        if (n > 0 && sm) {
            hipLaunchKernelGGL(k_act_fwd, cdiv(n, 256), 256, 0, st, n, no, c.a[l], m->acts[l], c.a[l]);
            HIPCHK(hipGetLastError());
        }
        h = c.a[l];
    }
    if (m->has_bn) {
        const int F = m->dims.back();
        float *y = nullptr;
        if ((rc = buf.get(&c.xhat, (size_t)n * F)) || (rc = buf.get(&y, (size_t)n * F))) return rc;
        if (net.calls >= std::max(1, net.max_calls)) return gnn_fail(GNN_ERR_STATE, "more BatchNormalization calls than announced");
        c.stats = net.stats_all + (size_t)net.calls++ * 2 * F;
        if (comm) {
            // every rank takes part in the exchange, also one without rows (count 0)
            float *tri = nullptr, *tri_all = nullptr;
            if ((rc = buf.get(&tri, (size_t)3 * F)) || (rc = buf.get(&tri_all, (size_t)3 * F * comm->world))) return rc;
            const int64_t rpb = n > 0 ? rows_per_block(n) : 1;
            const int parts = n > 0 ? (int)cdiv(n, rpb) : 0;
            float *part = nullptr;
            if ((rc = buf.get(&part, (size_t)std::max(parts, 1) * 2 * F))) return rc;
            if (n > 0) {
                const int cs = column_shift(F);
                hipLaunchKernelGGL(k_bn_stats, dim3(cdiv(F, 1 << cs), parts), 256, 0, st, n, F, cs, h, part, rpb);
            }
            hipLaunchKernelGGL(k_bn_local, cdiv(F, 256), 256, 0, st, n, F, part, parts, rpb, tri);
            HIPCHK(hipGetLastError());
            if ((rc = gnn_comm_allgather32(comm, tri, tri_all, (size_t)3 * F, st))) return rc;
            hipLaunchKernelGGL(k_bn_apply_ext, n > 0 ? elementwise_grid(n * F) : 1, 256, sizeof(float) * 2 * F, st, n, F, h, tri_all, comm->world, m->eps, net.gamma,
                               net.beta, c.xhat, y, c.stats);
            HIPCHK(hipGetLastError());
        } else if (n > 0) {
            const int64_t rpb = rows_per_block(n);
            const int parts = (int)cdiv(n, rpb);
            float *part = nullptr;
            if ((rc = buf.get(&part, (size_t)parts * 2 * F))) return rc;
            const int cs = column_shift(F);
            hipLaunchKernelGGL(k_bn_stats, dim3(cdiv(F, 1 << cs), parts), 256, 0, st, n, F, cs, h, part, rpb);
            hipLaunchKernelGGL(k_bn_apply, elementwise_grid(n * F), 256, sizeof(float) * 2 * F, st, n, F, cs, h, part, parts, rpb, m->eps, net.gamma,
                               net.beta, c.xhat, y, c.stats);
            HIPCHK(hipGetLastError());
        }
        h = y;
    }
    *y_out = h;
    return GNN_OK;
}

// what follows the last layer of net_state's backward pass in the same launch as the sum of the chunk partials (k_state_grad_sum)
struct StateGradJob {
    int64_t N;
    int Ds, in_s, c_aggs;
    const int32_t *sip, *sdst;
    const float *sw;
    float *d_state;               // out: d loss / d state of the body's input
};

// back-propagation through one Sequential: d is d loss / d y on entry ([n, dims.back()], overwritten); on return *dx_out is
// d loss / d x ([n, dims[0]]); weight gradients are ADDED into net.grads (one sum over the call's chunk partials)
// comm != NULL (sharded backward, one process per rank): the sums of BatchNormalization's backward pass are those of the rows of ALL
// ranks (n_global of them); the weight gradients stay this rank's share (train_backward adds the shares up at the end).
int net_backward(hipStream_t st, Buf &buf, Net &net, const NetCache &c, float *d, float **dx_out, const StateGradJob *job = nullptr,
                 gnn_comm *comm = nullptr, int64_t n_global = 0)
{
    const gnn_mlp *m = net.m;
    const int L = m->n_layers;
    const int64_t n = c.n;
    int rc;
    if (n <= 0) {                              // no rows: no gradient; d x is empty
        if (comm && m->has_bn) {               // ... but the other ranks wait for this one's (zero) share of the sums
            const int F = m->dims.back();
            float *loc = nullptr, *all = nullptr;
            if ((rc = buf.get(&loc, (size_t)2 * F)) || (rc = buf.get(&all, (size_t)2 * F * comm->world))) return rc;
            HIPCHK(hipMemsetAsync(loc, 0, sizeof(float) * 2 * F, st));
            if ((rc = gnn_comm_allgather32(comm, loc, all, (size_t)2 * F, st))) return rc;
        }
        *dx_out = d;
        return GNN_OK;
    }
    const int64_t rpb = rows_per_block(n);
    const int parts = (int)cdiv(n, rpb);
    if (net.part_rows != n) {
        if ((rc = buf.get(&net.part, (size_t)parts * net.g_total))) return rc;
        net.part_rows = n;
    }
    const int64_t ps = (int64_t)net.g_total;
    const int act_last = m->acts[L - 1];
    // the derivative of the last activation rides on the BatchNormalization pass when nothing sits between them
    bool last_act_done = false;
    if (m->has_bn) {
        const int F = m->dims.back(), cs = column_shift(F);
        float *p_dyx = net.part + net.g_off[2 * L], *p_dy = net.part + net.g_off[2 * L + 1];
        const bool fuse = net.rate[L] == 0.0f && act_last != GNN_ACT_SOFTMAX;
        hipLaunchKernelGGL(k_colreduce2, dim3(cdiv(F, 1 << cs), parts), 256, 0, st, n, F, cs, d, c.xhat, p_dyx, p_dy, ps, rpb);
        if (comm) {
            // this rank's sums [sum d y xhat | sum d y] (adjacent in the gradient vector: they ARE the gamma / beta gradients), those of
            // all ranks all-gathered, added in rank order by every block of the apply kernel
            float *loc = nullptr, *all = nullptr;
            if ((rc = buf.get(&loc, (size_t)2 * F)) || (rc = buf.get(&all, (size_t)2 * F * comm->world))) return rc;
            hipLaunchKernelGGL(k_sum_strided, cdiv(2 * F, 256), 256, 0, st, parts, ps, p_dyx, 2 * F, loc);
            HIPCHK(hipGetLastError());
            if ((rc = gnn_comm_allgather32(comm, loc, all, (size_t)2 * F, st))) return rc;
            hipLaunchKernelGGL(k_bn_bwd_apply, elementwise_grid(n * F), 256, sizeof(float) * 2 * F, st, n, F, cs, d, c.xhat, net.gamma, c.stats, m->eps,
                               all, all + F, (int64_t)2 * F, comm->world, c.a[L - 1], fuse ? act_last : -1, n_global);
        } else
            hipLaunchKernelGGL(k_bn_bwd_apply, elementwise_grid(n * F), 256, sizeof(float) * 2 * F, st, n, F, cs, d, c.xhat, net.gamma, c.stats, m->eps,
                               p_dyx, p_dy, ps, parts, c.a[L - 1], fuse ? act_last : -1, (int64_t)0);
        HIPCHK(hipGetLastError());
        last_act_done = fuse;
    }
    if (net.rate[L] != 0.0f) {
        const int F = m->dims.back();
        hipLaunchKernelGGL(k_dropout_bwd, cdiv(n * F, 256), 256, 0, st, n * F, c.keep[L], net.rate[L], d);
        HIPCHK(hipGetLastError());
    }
    if (!last_act_done) {
        const int no = m->dims[L];
        const bool sm = act_last == GNN_ACT_SOFTMAX;
        if (sm || act_last != GNN_ACT_LINEAR) {
            hipLaunchKernelGGL(k_act_bwd, cdiv(sm ? n : n * no, 256), 256, 0, st, n, no, d, c.a[L - 1], act_last);
            HIPCHK(hipGetLastError());
        }
    }
    // d is d loss / d z of layer l at the top of every pass
    // a 3-layer net without Dropout on many rows: the whole chain d z2 -> d z1 -> d z0 -> d inp in one pass (k_bwd3_split), then the three weight gradients
    float *dsg = nullptr;
    bool chain3 = n > 0 && L == 3 && tg_many_rows(n) && bwd3_covers(m) && net.rate[0] == 0.0f && net.rate[1] == 0.0f && net.rate[2] == 0.0f;
    for (int l = 0; l < 3 && chain3; ++l) chain3 = tg_wide(m->dims[l + 1], m->dims[l]) && tg_wgrad_covers(m->dims[l], m->dims[l + 1]);
#ifdef GNN_DIAG
    static const bool chain_off = getenv("GNN_TRAIN_BWD3") && atoi(getenv("GNN_TRAIN_BWD3")) == 0;
    if (chain_off) chain3 = false;
#endif
    if (chain3) {
        float *dz1 = nullptr, *dz0 = nullptr, *dinp = nullptr;
        if ((rc = buf.get(&dz1, (size_t)n * m->dims[2])) || (rc = buf.get(&dz0, (size_t)n * m->dims[1])) || (rc = buf.get(&dinp, (size_t)n * m->dims[0]))) return rc;
        // the state gradient of this body reads two column blocks of d inp: the chain leaves them once more as aligned rows
        if (job && job->N == n && (job->Ds & 3) == 0 && job->Ds <= 64 && job->in_s == m->dims[0] && (rc = buf.get(&dsg, (size_t)n * 2 * job->Ds))) return rc;
        if ((rc = launch_bwd3(st, buf, m, net.WT.data(), n, d, c.a[1], c.a[0], dz1, dz0, dinp, dsg, job ? job->Ds : 0, job ? job->c_aggs : 0))) return rc;
        const float *dzs[3] = {dz0, dz1, d};
        for (int l = 2; l >= 0; --l)
            if ((rc = launch_wgrad_f32(st, n, rpb, parts, ps, m->dims[l], m->dims[l + 1], c.hin[l], dzs[l], net.part + net.g_off[2 * l]))) return rc;
        d = dinp;
    }
    for (int l = chain3 ? -1 : L - 1; l >= 0; --l) {
        const int ni = m->dims[l], no = m->dims[l + 1];
        float *dprev = nullptr;
        if ((rc = buf.get(&dprev, (size_t)n * ni))) return rc;
        // weight + bias gradient tiles and d h_in = d z . W^T (back through Dropout l and, l > 0, the activation of layer l - 1) in one launch
        const int act_prev = l > 0 ? m->acts[l - 1] : -1;
        const bool prev_sm = act_prev == GNN_ACT_SOFTMAX;
        if (tg_wide(no, ni) && tg_wgrad_covers(ni, no) && !prev_sm && tg_many_rows(n)) {      // both products of a wide layer on the matrix cores
            if ((rc = launch_wgrad_f32(st, n, rpb, parts, ps, ni, no, c.hin[l], d, net.part + net.g_off[2 * l]))) return rc;
            if ((rc = launch_gemm_f32(st, buf, n, no, ni, d, net.WT[l], nullptr, act_prev, 1, net.rate[l] != 0.0f ? c.keep[l] : nullptr, net.rate[l],
                                      l > 0 ? c.a[l - 1] : nullptr, dprev))) return rc;
            d = dprev;
            continue;
        }
        constexpr int R = 8;
        LayerBwd p;
        p.n = n; p.rows_per_block = rpb; p.pstride = ps;
        p.n_in = ni; p.n_out = no; p.n_out_pad = (no + 3) & ~3; p.act = prev_sm ? -1 : act_prev;
        p.wg_bx = (int)cdiv(ni + 1, GNN_WG_TILE); p.wg_by = (int)cdiv(no, GNN_WG_TILE); p.wg_blocks = p.wg_bx * p.wg_by * parts;
        p.rate = net.rate[l];
        p.H = c.hin[l]; p.DZ = d; p.WT = net.WT[l]; p.a_prev = l > 0 ? c.a[l - 1] : nullptr;
        p.keep = net.rate[l] != 0.0f ? c.keep[l] : nullptr;
        p.part = net.part + net.g_off[2 * l]; p.dprev = dprev;
        p.cshift = dense_cshift(ni);
        const size_t lds = std::max(dense_lds_bytes(R, p.n_out_pad), sizeof(float) * 2 * 64 * GNN_WG_LD);
        if (lds > 64 * 1024) return gnn_fail(GNN_ERR_UNSUPPORTED, "layer width %d too large", no);
        hipLaunchKernelGGL((k_layer_bwd<R>), (unsigned)(p.wg_blocks + cdiv(n, R)), 256, lds, st, p);
        if (prev_sm) hipLaunchKernelGGL(k_act_bwd, cdiv(n, 256), 256, 0, st, n, ni, dprev, c.a[l - 1], act_prev);
        HIPCHK(hipGetLastError());
        d = dprev;
    }
    const unsigned sum_blocks = cdiv((int64_t)net.g_total, 64);
    if (job && dsg) {
        hipLaunchKernelGGL(k_state_grad_rows_al, cdiv(job->N * 16, 256), 256, 0, st, job->N, job->Ds, dsg, job->sip, job->sdst, job->sw, job->d_state);
        hipLaunchKernelGGL(k_sum_parts, sum_blocks, 256, 0, st, parts, (int64_t)net.g_total, net.part, net.grads);
    } else if (job && tg_many_rows(job->N) && (job->Ds & 3) == 0 && job->Ds <= 64) {
        hipLaunchKernelGGL(k_state_grad_rows, cdiv(job->N * 16, 256), 256, 0, st, job->N, job->Ds, job->in_s, job->c_aggs, d, job->sip, job->sdst, job->sw, job->d_state);
        hipLaunchKernelGGL(k_sum_parts, sum_blocks, 256, 0, st, parts, (int64_t)net.g_total, net.part, net.grads);
    } else if (job && job->N > 0) {
        const int sg = (int)cdiv(job->N * job->Ds, 256);
        hipLaunchKernelGGL(k_state_grad_sum, sg + sum_blocks, 256, 0, st, sg, job->N, job->Ds, job->in_s, job->c_aggs, d, job->sip, job->sdst, job->sw,
                           job->d_state, parts, (int64_t)net.g_total, net.part, net.grads);
    } else
        hipLaunchKernelGGL(k_sum_parts, sum_blocks, 256, 0, st, parts, (int64_t)net.g_total, net.part, net.grads);
    HIPCHK(hipGetLastError());
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

// The same on the device, one thread per target row: d_o [n, T] and, per block of 256 rows, the sum of w_i L_i (fixed tree, double)
__global__ void __launch_bounds__(256) k_loss_rows(int kind, int64_t n, int T, const float *t, const float *o, const float *w, float *d_o, double *loss_part)
{
    __shared__ double sl[256];
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double lw = 0.0;
    if (i < n) {
        const float *ti = t + i * T, *oi = o + i * T;
        const double wi = w[i];
        if (kind == 0) {
            double s = 0.0;
            for (int j = 0; j < T; ++j) s += oi[j];
            double li = 0.0, gp = 0.0;
            for (int j = 0; j < T; ++j) {
                const double pj = oi[j] / s;
                const bool in = pj >= 1e-7 && pj <= 1.0 - 1e-7;
                const double pc = fmin(fmax(pj, 1e-7), 1.0 - 1e-7);
                li -= ti[j] * log(pc);
                gp += (in ? -ti[j] / pc : 0.0) * pj;
            }
            for (int j = 0; j < T; ++j) {
                const double pj = oi[j] / s;
                const bool in = pj >= 1e-7 && pj <= 1.0 - 1e-7;
                const double pc = fmin(fmax(pj, 1e-7), 1.0 - 1e-7);
                d_o[i * T + j] = (float)(wi * ((in ? -ti[j] / pc : 0.0) - gp) / s);
            }
            lw = wi * li;
        } else if (kind == 2) {
            double mx = oi[0], s = 0.0, st = 0.0, li = 0.0;
            for (int j = 1; j < T; ++j) mx = fmax(mx, (double)oi[j]);
            for (int j = 0; j < T; ++j) { s += exp(oi[j] - mx); st += ti[j]; }
            for (int j = 0; j < T; ++j) {
                const double logp = (oi[j] - mx) - log(s);
                li -= ti[j] * logp;
                d_o[i * T + j] = (float)(wi * (exp(logp) * st - ti[j]));
            }
            lw = wi * li;
        } else {
            double li = 0.0;
            for (int j = 0; j < T; ++j) { const double e = (double)oi[j] - ti[j]; li += e * e; d_o[i * T + j] = (float)(wi * 2.0 * e / T); }
            lw = wi * li / T;
        }
    }
    sl[threadIdx.x] = lw;
    __syncthreads();
    for (int h = 128; h > 0; h >>= 1) {
        if ((int)threadIdx.x < h) sl[threadIdx.x] += sl[threadIdx.x + h];
        __syncthreads();
    }
    if (threadIdx.x == 0) loss_part[blockIdx.x] = sl[0];
}

// GNNgraphBased readout inside the step: og = NodeGraph^T . out_nodes (GNN.py:331-332) over the CSR by graph, one wave per graph
// (lanes take every 64th entry, then a fixed butterfly adds the lanes), and its transpose
__global__ void __launch_bounds__(256) k_graph_out(int n_graphs, int T, const int32_t *__restrict__ ng_indptr, const int32_t *__restrict__ ng_node,
                                                   const float *__restrict__ ng_w, const float *__restrict__ out_nodes, float *og)
{
    const int gi = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (gi >= n_graphs) return;
    const int e0 = ng_indptr[gi], e1 = ng_indptr[gi + 1];
    for (int c = 0; c < T; ++c) {
        float acc = 0.0f;
        for (int e = e0 + lane; e < e1; e += 64) acc += ng_w[e] * out_nodes[(int64_t)ng_node[e] * T + c];
        for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
        if (lane == 0) og[gi * T + c] = acc;
    }
}

// d out_nodes[node, c] = w * d og[graph of the entry, c]: one thread per entry and column, the entry's graph by bisection
__global__ void k_graph_out_bwd(int n_graphs, int T, const int32_t *__restrict__ ng_indptr, const int32_t *__restrict__ ng_node,
                                const float *__restrict__ ng_w, const float *__restrict__ d_og, float *d_nodes)
{
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int ne = ng_indptr[n_graphs];
    if (t >= ne * T) return;
    const int e = t / T, c = t - e * T;
    int lo = 0, hi = n_graphs;                 // largest lo with ng_indptr[lo] <= e
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (ng_indptr[mid] <= e) lo = mid; else hi = mid; }
    // a node belongs to one graph: the adds of different threads do not meet (atomic only so that a malformed NodeGraph stays defined)
    atomicAdd(&d_nodes[(int64_t)ng_node[e] * T + c], ng_w[e] * d_og[lo * T + c]);
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

// GNNedgeBased backward: row q of d_feats = d [F[dst(e)] | F[src(e)] | arc label], e = rows[q]; F = [state | labels?].  Both endpoints
// receive their half.  Several arcs share a node: every (node, column) is one thread that adds the halves of the node's masked arcs in
// ascending arc order (gnn_loop_set_edge_readout builds the incidence lists) on top of what d_state / d_nodes hold - no float atomics,
// so edge-based steps are run-to-run identical like the others (round 3; the scatter with atomicAdd was not).
__global__ void k_gather_edge_grad(int64_t n, const int32_t *__restrict__ inc_ptr, const int32_t *__restrict__ inc, const float *__restrict__ d_feats, int we,
                                   int wn, int Ds, int NL, float *d_state, float *d_nodes)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= n * wn) return;
    const int64_t node = t / wn;
    const int c = (int)(t - node * wn);
    float *dst = c < Ds ? d_state + node * Ds + c : (d_nodes ? d_nodes + node * NL + (c - Ds) : nullptr);
    if (!dst) return;
    float v = *dst;
    for (int32_t i = inc_ptr[node]; i < inc_ptr[node + 1]; ++i) {
        const int32_t x = inc[i];
        v = v + d_feats[(int64_t)(x >> 1) * we + (x & 1) * wn + c];
    }
    *dst = v;
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
    int64_t N = 0, M = 0;
    int64_t N_global = 0, M_global = 0;   // sharded forward: the rows / masked rows of all ranks
    bool backward_done = false;   // the gradients are complete (and the activations spent)
    bool applied = false;         // gnn_loop_optimizer_step has consumed them
};

constexpr int TRAIN_CHUNK = 5;     // bodies enqueued between two looks at the iteration gates (see train_forward)

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


// Training-mode Loop.  final_sync: wait for the published state / outputs (and out_nodes_host) before returning.
// owned rows [n_rows, Ds] -> a fresh replica [N_pad, Ds] with the rows of all ranks (all-gather in place); sharded training only
static int train_replicate(gnn_loop *l, Buf &buf, hipStream_t st, const float *own_rows, float **replica)
{
    const int Ds = l->Ds;
    const size_t replica_floats = (size_t)l->N_pad * Ds, shard_floats = (size_t)l->shard_rows * Ds;
    int rc;
    if ((rc = buf.get(replica, replica_floats))) return rc;
    HIPCHK(hipMemsetAsync(*replica, 0, sizeof(float) * replica_floats, st));        // rows past the last shard's end are never read, but stay finite
    if (l->g->n_rows) HIPCHK(hipMemcpyAsync(*replica + (size_t)l->own_off * Ds, own_rows, sizeof(float) * (size_t)l->g->n_rows * Ds, hipMemcpyDeviceToDevice, st));
    return gnn_comm_allgather32(l->comm, *replica + (size_t)l->rank * shard_floats, *replica, shard_floats, st);
}

static int train_forward(gnn_loop *l, const int32_t *src_indptr, const int32_t *src_dst, const float *src_w, const float *dropout_state,
                         const float *dropout_output, const uint8_t *masks_state, const uint8_t *masks_output, uint64_t seed,
                         const float *bn_state, const float *bn_output, float *k_out, float *out_nodes_host, bool final_sync)
{
    ARGCHK(l && dropout_state && dropout_output && k_out, "bad arguments");
    // Sharded FORWARD (round 3): node-range shards with full-replica numbering, one process per rank - the state rows are all-gathered
    // after every body, the BatchNormalization statistics and the iteration gates are those of all ranks.  gnn_loop_train_backward
    // continues on the shards when the by-source adjacency of the owned rows was given here.
    const bool sharded = l->world > 1;
    gnn_comm *comm = sharded ? l->comm : nullptr;
    if (sharded) {
        ARGCHK(l->comm && !l->comm->grp, "training forward on shards: one process per rank (an RCCL communicator), not a loopback group");
        ARGCHK(!l->g->halo_world && !l->slice_mode && !l->edge_mode, "training forward on shards: node-range shards with full-replica numbering, node- or graph-based");
    }
    ARGCHK(l->edge_mode == l->edge_expected, "edge-based net_output: call gnn_loop_set_edge_readout first");
    if (!l->have_state0 && l->D) return gnn_fail(GNN_ERR_STATE, "state_vect_dim > 0: call gnn_loop_set_state0 first");
    gnn_graph *g = l->g;
    const int64_t N = g->n_rows, M = l->edge_mode ? l->n_edge_masked : g->n_masked, E = g->E;
    const int Ds = l->Ds, NLc = l->NLc, in_s = l->in_s, T = l->T, wf = l->ou->dims[0];
    int rc0 = 0;
    HIPCHK(hipSetDevice(l->device));
    hipStream_t st = l->stream;
    if (!l->graph_ready_seen) {      // creation-time fills of a derived graph's labels come before their first read (gnn_graph_wait_ready)
        if ((rc0 = gnn_graph_wait_ready(g, st))) return rc0;
        l->graph_ready_seen = true;
    }
    gnn_train_ctx_free(l);
    if (!l->train_arena) l->train_arena = new TrainArena();
    TrainArena *arena = static_cast<TrainArena *>(l->train_arena);
    arena->reset();
    TrainCtx *cx = new TrainCtx();
    l->train_ctx = cx;
    cx->buf.arena = arena;
    cx->N = N; cx->M = M;
    Buf &buf = cx->buf;
    Net &ns = cx->ns, &no_ = cx->no_;
    int rc;
    // everything the step needs zeroed, in one block and one memset: gradients and BatchNormalization statistics of both nets, the
    // iteration gates, the template of the concat
    const int max_iter = l->max_iter;
    const size_t flag_words = (size_t)(max_iter + 1) * GNN_FLAG_WORDS;
    const size_t z_s = net_zero_floats(l->st, max_iter), z_o = net_zero_floats(l->ou, 1), z_f = (flag_words + 63) & ~(size_t)63;
    const size_t z_total = z_s + z_o + z_f + (size_t)N * l->in_s;
    float *zero_mem = nullptr;
    if ((rc = buf.get(&zero_mem, z_total))) return rc;
    HIPCHK(hipMemsetAsync(zero_mem, 0, sizeof(float) * std::max<size_t>(1, z_total), st));
    if ((rc = net_setup(st, buf, ns, l->st, dropout_state, bn_state, max_iter, zero_mem)) ||
        (rc = net_setup(st, buf, no_, l->ou, dropout_output, bn_output, 1, zero_mem + z_s))) return rc;
    int *flags = reinterpret_cast<int *>(zero_mem + z_s + z_o);
    float *tmpl = zero_mem + z_s + z_o + z_f;
    // Adjacency by source for the transposed aggregation of the backward pass: the caller's arrays, or (NULL) the graph's
    // own copy, built once from its CSR by destination (a stable counting sort by source keeps destinations ascending)
    if (sharded) {
        // the backward pass on shards needs the arcs that LEAVE the owned rows (by-source CSR over the owned rows, destinations as
        // replica rows): the caller's arrays, or none - gnn_loop_train_backward is then refused
        if (src_indptr) {
            const int64_t Es = src_indptr[N];
            ARGCHK(src_indptr[0] == 0 && Es >= 0 && (Es == 0 || (src_dst && src_w)), "bad by-source CSR");
            if ((rc = buf.get(&cx->d_sip, (size_t)N + 1)) || (rc = buf.get(&cx->d_sdst, (size_t)std::max<int64_t>(Es, 1))) || (rc = buf.get(&cx->d_sw, (size_t)std::max<int64_t>(Es, 1)))) return rc;
            HIPCHK(hipMemcpy(cx->d_sip, src_indptr, sizeof(int32_t) * (N + 1), hipMemcpyHostToDevice));
            if (Es) { HIPCHK(hipMemcpy(cx->d_sdst, src_dst, sizeof(int32_t) * Es, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(cx->d_sw, src_w, sizeof(float) * Es, hipMemcpyHostToDevice)); }
        }
        // rows and masked rows of all ranks (BatchNormalization's backward pass divides by them)
        int *cnt = nullptr, *cnt_all = nullptr;
        if ((rc = buf.get(&cnt, (size_t)4)) || (rc = buf.get(&cnt_all, (size_t)4 * l->world))) return rc;
        const int mine[4] = {(int)N, (int)M, 0, 0};
        HIPCHK(hipMemcpy(cnt, mine, sizeof(mine), hipMemcpyHostToDevice));
        if ((rc = gnn_comm_allgather32(comm, cnt, cnt_all, 4, st))) return rc;
        std::vector<int> all((size_t)4 * l->world);
        HIPCHK(hipStreamSynchronize(st));
        HIPCHK(hipMemcpy(all.data(), cnt_all, sizeof(int) * all.size(), hipMemcpyDeviceToHost));
        for (int p = 0; p < l->world; ++p) { cx->N_global += all[(size_t)4 * p]; cx->M_global += all[(size_t)4 * p + 1]; }
    } else if (src_indptr) {
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
            if (gnn_dev_malloc((void **)&sh->src_indptr, sizeof(int32_t) * (N + 1)) != hipSuccess || gnn_dev_malloc((void **)&sh->src_dst, sizeof(int32_t) * std::max<int64_t>(E, 1)) != hipSuccess ||
                gnn_dev_malloc((void **)&sh->src_w, sizeof(float) * std::max<int64_t>(E, 1)) != hipSuccess)
                return gnn_fail(GNN_ERR_HIP, "hipMalloc of the by-source adjacency failed");
            HIPCHK(hipMemcpy(sh->src_indptr, sip.data(), sizeof(int32_t) * (N + 1), hipMemcpyHostToDevice));
            if (E) { HIPCHK(hipMemcpy(sh->src_dst, sdst.data(), sizeof(int32_t) * E, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(sh->src_w, sw.data(), sizeof(float) * E, hipMemcpyHostToDevice)); }
        }
        cx->d_sip = sh->src_indptr; cx->d_sdst = sh->src_dst; cx->d_sw = sh->src_w;
    }

    // template of the concat with the loop-invariant columns filled in (GNN.py:259, :263)
    const int c_nodes = Ds, c_aggs = Ds + NLc, c_aggn = c_aggs + Ds, c_agga = c_aggn + NLc;
    if ((rc = gnn_launch_spmm(st, N, g->sh->indptr, nullptr, g->sh->arc_w, gnn_graph_arc_labels(g), g->AL, g->AL, tmpl + c_agga, in_s, nullptr, 1))) return rc;
    if (l->D) {
        if ((rc = gnn_launch_spmm(st, N, g->sh->indptr, g->sh->adj_src, g->sh->adj_w, g->nodes, g->NL, g->NL, tmpl + c_aggn, in_s, nullptr, 1))) return rc;
        if ((rc = gnn_launch_copy_cols(st, N, g->NL, g->nodes + (size_t)g->own_off * g->NL, g->NL, tmpl + c_nodes, in_s, nullptr, 1))) return rc;
    }
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
    // Gate i = condition(state_i, state_{i-1}) decides whether body i runs; it is evaluated by the body's own first kernel.  The
    // bodies are enqueued TRAIN_CHUNK at a time without waiting for their gates; the host then reads the gates of the chunk in one
    // synchronisation, and the bodies enqueued from a closed gate on (at most TRAIN_CHUNK of them) are dropped: their results are
    // never read.
    int *hflags = static_cast<int *>(arena->host(std::max<size_t>(sizeof(int) * flag_words, 4096)));
    if (!hflags) return gnn_fail(GNN_ERR_HIP, "hipHostMalloc failed");
    std::vector<float *> states;                       // states[i]: the state body i reads (row 0 of a replica); states[0] is read in place
    const size_t own_off = sharded ? (size_t)l->own_off : 0;      // replica row of the first owned row
    const size_t replica_floats = (size_t)l->N_pad * Ds;
    auto replicate = [&](const float *own_rows, float **replica) -> int { return train_replicate(l, buf, st, own_rows, replica); };
    if (sharded && l->D) {
        float *rep0 = nullptr;
        if ((rc = replicate(l->state_init, &rep0))) return rc;
        states.push_back(rep0);
    } else
        states.push_back(const_cast<float *>(l->D ? l->state_init : g->nodes));      // (D == 0: the node labels, a replica already)
    int enq = 0, k = -1;
    if ((!sharded && N == 0) || max_iter == 0) k = 0;  // no node can raise a gate / no body allowed (a rank without rows still follows the others)
    int *flags_all = nullptr;
    if (sharded && (rc = buf.get(&flags_all, flag_words * (size_t)l->world))) return rc;
    while (k < 0) {
        // the first look at the gates comes behind as many bodies as the loop's last training forward ran, plus one (a batch's iteration
        // count moves slowly from epoch to epoch: k = 11 is then one synchronisation and one dropped body instead of three and four)
        int chunk = enq == 0 && l->train_k_hint >= TRAIN_CHUNK ? l->train_k_hint + 1 : TRAIN_CHUNK;
#ifdef GNN_DIAG
        static const bool hint_off = getenv("GNN_TRAIN_K_HINT") && atoi(getenv("GNN_TRAIN_K_HINT")) == 0;
        if (hint_off) chunk = TRAIN_CHUNK;
#endif
        const int target = std::min(max_iter, enq + chunk);
        for (; enq < target; ++enq) {
            float *inp = nullptr, *y = nullptr;
            uint8_t *keep0 = nullptr;
            const float r0 = dropout_state[0];
            if ((rc = buf.get(&inp, (size_t)N * in_s))) return rc;
            if (r0 != 0.0f && (rc = buf.get(&keep0, (size_t)N * in_s))) return rc;
            const uint8_t *mk = d_masks_s ? d_masks_s + mask_iter_bytes * (size_t)enq : nullptr;
            const uint64_t sd = seed + 7919ull * (uint64_t)(enq + 1);
            const float *own_cur = states[enq] + own_off * Ds, *own_prev = enq ? states[enq - 1] + own_off * Ds : (const float *)nullptr;
            MlpFwd build{};
            const bool fused_input = N > 0 && r0 == 0.0f && mlp_small_fused_ok(ns, N);
            if (fused_input) {
                // few rows: k_mlp_fwd builds the concat rows itself (one launch for input + all Dense layers) and evaluates the gate
                build.Ds = Ds; build.c_aggs = c_aggs; build.tmpl = tmpl; build.state = states[enq]; build.own = own_cur; build.own_prev = own_prev;
                build.indptr = g->sh->indptr; build.adj_src = g->sh->adj_src; build.adj_w = g->sh->adj_w; build.thr = l->thr;
                build.flag = flags + (size_t)enq * GNN_FLAG_WORDS;
            } else if (N == 0) {
                // (a rank without rows: nothing to compute, it only takes part in the exchanges below)
            } else if (r0 == 0.0f && (Ds & 3) == 0 && Ds <= 64 && tg_many_rows(N)) {
                // many rows: the concat 16 lanes per row, gate i = condition(state_i, state_{i-1}) by k_check beside it
                hipLaunchKernelGGL(k_train_input_rows, cdiv(N * 16, 256), 256, 0, st, N, in_s, Ds, c_aggs, tmpl, states[enq], own_cur, g->sh->indptr, g->sh->adj_src,
                                   g->sh->adj_w, inp);
                HIPCHK(hipGetLastError());
                if ((rc = gnn_launch_check(st, N, Ds, own_cur, own_prev, l->thr, flags + (size_t)enq * GNN_FLAG_WORDS))) return rc;
            } else {
                // the input kernel of body i also evaluates gate i = condition(state_i, state_{i-1})
                hipLaunchKernelGGL(k_train_input, cdiv(N * in_s, 256), 256, 0, st, N, in_s, Ds, c_aggs, tmpl, states[enq], own_cur, g->sh->indptr, g->sh->adj_src,
                                   g->sh->adj_w, r0, mk, sd + 0x9E37ull, keep0, inp, own_prev, l->thr,
                                   flags + (size_t)enq * GNN_FLAG_WORDS);
                HIPCHK(hipGetLastError());
            }
            cx->caches.emplace_back();
            if ((rc = net_forward(st, buf, ns, N, inp, keep0, mk, sd, cx->caches.back(), &y, comm, fused_input ? &build : nullptr))) return rc;
            if (sharded) {                             // the new rows of all ranks: what the next body gathers from
                float *rep = nullptr;
                if ((rc = replicate(y, &rep))) return rc;
                states.push_back(rep);
            } else
                states.push_back(y);
        }
        if (sharded) {                                 // the gates of all ranks (GNN.py:218: reduce_any over ALL nodes)
            if ((rc = gnn_comm_allgather32(comm, flags, flags_all, flag_words, st))) return rc;
            HIPCHK(hipMemcpyAsync(hflags, flags_all, sizeof(int) * flag_words, hipMemcpyDeviceToHost, st));     // rank 0's block first; the others below
        } else
            HIPCHK(hipMemcpyAsync(hflags, flags, sizeof(int) * (size_t)enq * GNN_FLAG_WORDS, hipMemcpyDeviceToHost, st));
        HIPCHK(hipStreamSynchronize(st));
        std::vector<int> others;
        if (sharded && l->world > 1) {
            others.resize(flag_words * (size_t)(l->world - 1));
            HIPCHK(hipMemcpy(others.data(), flags_all + flag_words, sizeof(int) * others.size(), hipMemcpyDeviceToHost));
        }
        for (int i = 0; i < enq && k < 0; ++i) {       // gates 0 .. enq - 1 are known; gate enq belongs to the next chunk's first body
            int any = 0;
            for (int w = 0; w < GNN_FLAG_WORDS; w += GNN_FLAG_STRIDE) any |= hflags[(size_t)i * GNN_FLAG_WORDS + w];
            for (int p = 1; sharded && p < l->world; ++p)
                for (int w = 0; w < GNN_FLAG_WORDS; w += GNN_FLAG_STRIDE) any |= others[(size_t)(p - 1) * flag_words + (size_t)i * GNN_FLAG_WORDS + w];
            if (!any) k = i;
        }
        if (k < 0 && enq == max_iter) k = max_iter;
    }
    cx->caches.resize((size_t)k);
    float *state = states[(size_t)k];
    // ---- net_output on the masked rows --------------------------------------------------------------------------------------
    float *feats = nullptr;
    if ((rc = buf.get(&feats, (size_t)M * wf))) return rc;
    if (l->edge_mode) {
        if ((rc = gnn_launch_feats_edge(st, l, state, feats))) return rc;
    } else if (M) {
        hipLaunchKernelGGL(k_gather_feats, cdiv(M * wf, 256), 256, 0, st, M, g->sh->masked_rows, state + own_off * Ds, Ds, g->nodes + (size_t)g->own_off * g->NL, g->NL, NLc, feats);
        HIPCHK(hipGetLastError());
    }
    if ((rc = net_forward(st, buf, no_, M, feats, nullptr, d_masks_o, seed + 104729ull, cx->co, &cx->out_nodes, comm))) return rc;
    cx->state = state;
    cx->k = k;
    l->train_k_hint = k;
    // publish the training-mode state / outputs as the loop's result: gnn_loop_get_state / get_output / readout and
    // gnn_graph_update_labels (LGNN stacking) read them exactly like an inference run's
    if (sharded) {                                     // the whole replica, as after an inference Loop (k == 0 with D == 0: the label rows there are)
        const size_t have = state == g->nodes ? (size_t)g->nodes_rows * Ds : replica_floats;
        HIPCHK(hipMemcpyAsync(l->state[0], state, sizeof(float) * std::min(have, replica_floats), hipMemcpyDeviceToDevice, st));
    }
    else if (N) HIPCHK(hipMemcpyAsync(l->state[0], state, sizeof(float) * (size_t)N * Ds, hipMemcpyDeviceToDevice, st));
    // (l->out changes here without loop_prepare: a graph readout that an earlier inference run folded into its persistent launch -
    // ng_host - must not be handed out for these outputs)
    l->ng_inlaunch = false;
    ++l->out_runs;
    if (M) HIPCHK(hipMemcpyAsync(l->out, cx->out_nodes, sizeof(float) * (size_t)M * T, hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemsetAsync(l->kfinal_dev, 0, sizeof(int), st));
    if (out_nodes_host && M) HIPCHK(hipMemcpyAsync(out_nodes_host, cx->out_nodes, sizeof(float) * (size_t)M * T, hipMemcpyDeviceToHost, st));
    if (final_sync) HIPCHK(hipStreamSynchronize(st));
    l->kfinal = 0;
    *l->kfinal_host = 0;
    l->ran = true;
    *k_out = (float)k;
    return GNN_OK;
}

extern "C" int gnn_loop_train_forward(gnn_loop *l, const int32_t *src_indptr, const int32_t *src_dst, const float *src_w,
                                      const float *dropout_state, const float *dropout_output, const uint8_t *masks_state,
                                      const uint8_t *masks_output, uint64_t seed, const float *bn_state, const float *bn_output,
                                      float *k_out, float *out_nodes_host)
{
    return train_forward(l, src_indptr, src_dst, src_w, dropout_state, dropout_output, masks_state, masks_output, seed, bn_state, bn_output, k_out,
                         out_nodes_host, true);
}

// d_out_dev: d loss / d out_nodes already on the device (gnn_loop_train_step), else d_out_host is uploaded
static int train_backward(gnn_loop *l, float *d_out_dev, const float *d_out_host, const float *d_state_extra, float *grads_state,
                          float *grads_output, float *bn_batch_state, float *bn_batch_output, float *d_nodes_host, float *d_arcs_host, bool sync)
{
    ARGCHK(l && grads_state && grads_output, "bad arguments");
    TrainCtx *cx = static_cast<TrainCtx *>(l->train_ctx);
    if (!cx || cx->backward_done) return gnn_fail(GNN_ERR_STATE, "gnn_loop_train_forward has not been called (one backward per forward)");
    // Sharded backward (round 3; after a sharded gnn_loop_train_forward that was given the by-source adjacency of the owned rows): per
    // body the gradient of the aggregated-state columns is all-gathered like the state in the forward pass and every rank adds up, for
    // its own rows, what its out-arcs carry back; BatchNormalization's sums are those of all ranks; at the end the ranks' shares of the
    // weight gradients are all-gathered and added in rank order, so every rank returns the same, complete gradients.
    const bool sharded = l->world > 1;
    gnn_comm *comm = sharded ? l->comm : nullptr;
    if (sharded) {
        ARGCHK(cx->d_sip, "backward on shards: gnn_loop_train_forward needs the by-source adjacency of the owned rows (src_indptr / src_dst / src_w)");
        ARGCHK(!d_state_extra && !d_nodes_host && !d_arcs_host && !l->edge_mode, "backward on shards: no extra state gradient, label or arc-label gradients, node- or graph-based only");
    }
    gnn_graph *g = l->g;
    const int64_t N = g->n_rows, M = l->edge_mode ? l->n_edge_masked : g->n_masked;
    const int Ds = l->Ds, NLc = l->NLc, in_s = l->in_s, T = l->T, wf = l->ou->dims[0], NL = g->NL, k = cx->k;
    ARGCHK(M == 0 || d_out_dev || d_out_host, "d_out_nodes is NULL");
    HIPCHK(hipSetDevice(l->device));
    hipStream_t st = l->stream;
    Buf &buf = cx->buf;
    Net &ns = cx->ns, &no_ = cx->no_;
    const int c_nodes = Ds, c_aggs = Ds + NLc, c_aggn = c_aggs + Ds;
    int rc;
    float *d_out = d_out_dev, *d_feats = nullptr, *d_state = nullptr, *tmp = nullptr, *d_nodes = nullptr, *via = nullptr;
    if ((!d_out && (rc = buf.get(&d_out, (size_t)M * T))) || (rc = buf.get(&d_state, (size_t)N * Ds))) return rc;
    if (!d_out_dev && M) HIPCHK(hipMemcpyAsync(d_out, d_out_host, sizeof(float) * (size_t)M * T, hipMemcpyHostToDevice, st));
    if ((rc = net_backward(st, buf, no_, cx->co, d_out, &d_feats, nullptr, comm, cx->M_global))) return rc;
    if (d_state_extra) { if (N) HIPCHK(hipMemcpyAsync(d_state, d_state_extra, sizeof(float) * (size_t)N * Ds, hipMemcpyHostToDevice, st)); }
    else HIPCHK(hipMemsetAsync(d_state, 0, sizeof(float) * std::max<size_t>(1, (size_t)N * Ds), st));
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
        HIPCHK(hipMemsetAsync(d_nodes, 0, sizeof(float) * std::max<size_t>(1, (size_t)N * NL), st));
    }
    if (l->edge_mode) {
        if (M) {
            const int wn = Ds + NLc;
            hipLaunchKernelGGL(k_gather_edge_grad, cdiv(N * wn, 256), 256, 0, st, N, l->edge_inc_ptr, l->edge_inc, d_feats, wf, wn, Ds, NL, d_state, d_nodes);
            HIPCHK(hipGetLastError());
        }
    } else if (M) {
        if (d_state_extra) {        // extra + scattered rows: scatter into a zero buffer, then add
            if ((rc = buf.get(&tmp, (size_t)N * Ds))) return rc;
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
        // net_backward consumes d_state (d loss / d state_{it+1}) in place; its last launch also writes d loss / d state_it into a new buffer
        float *d_inp = nullptr, *d_prev = nullptr;
        if ((rc = buf.get(&d_prev, (size_t)N * Ds))) return rc;
        if (sharded) {
            if ((rc = net_backward(st, buf, ns, cx->caches[it], d_state, &d_inp, nullptr, comm, cx->N_global))) return rc;
            // d state_it[r] = d inp[r, :Ds] + sum over the arcs r -> dst of w * d inp[dst, c_aggs:]: the aggregate columns of all ranks first
            float *dagg = nullptr, *rep = nullptr;
            if ((rc = buf.get(&dagg, (size_t)std::max<int64_t>(N, 1) * Ds))) return rc;
            if (N && (rc = gnn_launch_copy_cols(st, N, Ds, d_inp + c_aggs, in_s, dagg, Ds, nullptr, 1))) return rc;
            if ((rc = train_replicate(l, buf, st, dagg, &rep))) return rc;
            if (N) {
                if ((rc = gnn_launch_spmm(st, N, cx->d_sip, cx->d_sdst, cx->d_sw, rep, Ds, Ds, d_prev, Ds, nullptr, 1))) return rc;
                hipLaunchKernelGGL(k_add_cols, cdiv(N * Ds, 256), 256, 0, st, N, Ds, d_inp, in_s, 0, d_prev);
                HIPCHK(hipGetLastError());
            }
        } else {
            const StateGradJob job{N, Ds, in_s, c_aggs, cx->d_sip, cx->d_sdst, cx->d_sw, d_prev};
            if ((rc = net_backward(st, buf, ns, cx->caches[it], d_state, &d_inp, &job))) return rc;
        }
        d_state = d_prev;
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
    if (sharded) {          // the ranks' shares of the weight gradients -> their sum in rank order, on every rank (and in place: gnn_loop_optimizer_step reads it)
        for (Net *net : {&ns, &no_}) {
            float *all = nullptr;
            if ((rc = buf.get(&all, net->g_total * (size_t)l->world))) return rc;
            if ((rc = gnn_comm_allgather32(comm, net->grads, all, net->g_total, st))) return rc;
            HIPCHK(hipMemsetAsync(net->grads, 0, sizeof(float) * net->g_total, st));
            hipLaunchKernelGGL(k_sum_parts, cdiv((int64_t)net->g_total, 64), 256, 0, st, l->world, (int64_t)net->g_total, all, net->grads);
            HIPCHK(hipGetLastError());
        }
    }
    HIPCHK(hipMemcpyAsync(grads_state, ns.grads, sizeof(float) * ns.g_total, hipMemcpyDeviceToHost, st));
    HIPCHK(hipMemcpyAsync(grads_output, no_.grads, sizeof(float) * no_.g_total, hipMemcpyDeviceToHost, st));
    if (bn_batch_state && l->st->has_bn && k > 0)      // the statistics of the calls are adjacent, in call order
        HIPCHK(hipMemcpyAsync(bn_batch_state, ns.stats_all, sizeof(float) * (size_t)k * 2 * Ds, hipMemcpyDeviceToHost, st));
    if (bn_batch_output && l->ou->has_bn && M) HIPCHK(hipMemcpyAsync(bn_batch_output, cx->co.stats, sizeof(float) * 2 * T, hipMemcpyDeviceToHost, st));
    if (want_arcs && g->E) {
        if (M) hipLaunchKernelGGL(k_arc_grad_readout, cdiv(M * AL, 256), 256, 0, st, M, AL, l->edge_rows, d_feats, wf, 2 * (Ds + NLc), d_arcs);
        hipLaunchKernelGGL(k_arc_grad_agg, cdiv(g->E * AL, 256), 256, 0, st, g->E, AL, l->edge_dst, g->sh->arc_id, g->sh->arc_w, d_aa, d_arcs);
        HIPCHK(hipGetLastError());
        HIPCHK(hipMemcpyAsync(d_arcs_host, d_arcs, sizeof(float) * (size_t)g->E * AL, hipMemcpyDeviceToHost, st));
    }
    if (want_nodes && N)        // D == 0: state_0 = nodes (GNN.py:265), so the gradient of the initial state IS the label gradient
        HIPCHK(hipMemcpyAsync(d_nodes_host, l->D ? d_nodes : d_state, sizeof(float) * (size_t)N * NL, hipMemcpyDeviceToHost, st));
    if (sync) HIPCHK(hipStreamSynchronize(st));
    cx->backward_done = true;
    return GNN_OK;
}

extern "C" int gnn_loop_train_backward(gnn_loop *l, const float *d_out_nodes, const float *d_state_extra, float *grads_state,
                                       float *grads_output, float *bn_batch_state, float *bn_batch_output, float *d_nodes_host,
                                       float *d_arcs_host)
{
    // the context stays (its gradients feed gnn_loop_optimizer_step) until the next forward; a second backward is refused
    return train_backward(l, nullptr, d_out_nodes, d_state_extra, grads_state, grads_output, bn_batch_state, bn_batch_output, d_nodes_host, d_arcs_host, true);
}

// ---------------------------------------------------------------------------------------------------------------------
// Optimizer step on the device (reference GNN_BaseClass.py:243-247: optimizer.apply_gradients on the trainable variables of
// both nets; Keras BatchNormalization moving statistics): the weights, the optimizer slots and the gradients never leave HBM.
// ---------------------------------------------------------------------------------------------------------------------
namespace {
struct ParamMap {                 // gradient vector index -> parameter array
    int n = 0;
    int goff[36];                 // [n + 1]
    float *p[35];
};

// kind 0, SGD: h = {learning rate, momentum}: v <- momentum v - lr g, p <- p + v
// kind 1, Adam (Keras): h = {lr_t = lr sqrt(1 - b2^t) / (1 - b1^t), b1, b2, epsilon}: m, v updated, p <- p - lr_t m / (sqrt(v) + epsilon)
__global__ void k_optimizer(ParamMap mp, const float *g, float gscale, float *sa, float *sb, int kind, float h0, float h1, float h2, float h3)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= mp.goff[mp.n]) return;
    int a = 0;
    while (j >= mp.goff[a + 1]) ++a;
    float *p = mp.p[a] + (j - mp.goff[a]);
    const float gr = g[j] * gscale;
    if (kind == 1) {
        const float m = h1 * sa[j] + (1.0f - h1) * gr;
        const float v = h2 * sb[j] + (1.0f - h2) * gr * gr;
        sa[j] = m; sb[j] = v;
        *p = *p - h0 * m / (sqrtf(v) + h3);
    } else {
        const float v = h1 * sa[j] - h0 * gr;
        sa[j] = v;
        *p = *p + v;
    }
}

// moving <- moving * momentum + batch * (1 - momentum), once per BatchNormalization call, in call order
__global__ void k_bn_moving(int F, int calls, const float *stats_all, float momentum, float *raw)
{
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= F) return;
    float mean = raw[2 * F + j], var = raw[3 * F + j];
    for (int c = 0; c < calls; ++c) {
        mean = mean * momentum + stats_all[(size_t)c * 2 * F + j] * (1.0f - momentum);
        var = var * momentum + stats_all[(size_t)c * 2 * F + F + j] * (1.0f - momentum);
    }
    raw[2 * F + j] = mean; raw[3 * F + j] = var;
}

int optimizer_apply(hipStream_t st, gnn_mlp *m, const Net &net, int calls, int kind, const float *h, float gscale, float bn_momentum)
{
    const size_t total = net.g_total;
    if (!m->opt_a) {
        if (gnn_dev_malloc((void **)&m->opt_a, sizeof(float) * total) != hipSuccess || gnn_dev_malloc((void **)&m->opt_b, sizeof(float) * total) != hipSuccess)
            return gnn_fail(GNN_ERR_HIP, "hipMalloc of the optimizer slots failed");
        HIPCHK(hipMemsetAsync(m->opt_a, 0, sizeof(float) * total, st));
        HIPCHK(hipMemsetAsync(m->opt_b, 0, sizeof(float) * total, st));
    }
    ParamMap mp;
    const int L = m->n_layers;
    for (int l = 0; l < L; ++l) {
        mp.goff[2 * l] = (int)net.g_off[2 * l]; mp.p[2 * l] = m->W[l];
        mp.goff[2 * l + 1] = (int)net.g_off[2 * l + 1]; mp.p[2 * l + 1] = m->b[l];
    }
    mp.n = 2 * L;
    if (m->has_bn) {
        const int F = m->dims.back();
        mp.goff[mp.n] = (int)net.g_off[2 * L]; mp.p[mp.n] = m->bn_raw; ++mp.n;
        mp.goff[mp.n] = (int)net.g_off[2 * L + 1]; mp.p[mp.n] = m->bn_raw + F; ++mp.n;
    }
    mp.goff[mp.n] = (int)total;
    hipLaunchKernelGGL(k_optimizer, cdiv((int64_t)total, 256), 256, 0, st, mp, net.grads, gscale, m->opt_a, m->opt_b, kind, h[0], h[1], h[2], h[3]);
    HIPCHK(hipGetLastError());
    if (m->has_bn) {
        const int F = m->dims.back();
        if (calls > 0) hipLaunchKernelGGL(k_bn_moving, cdiv(F, 64), 64, 0, st, F, calls, net.stats_all, bn_momentum, m->bn_raw);
        HIPCHK(hipGetLastError());
        int rc = gnn_mlp_refresh_bn(m, st);
        if (rc) return rc;
    }
    m->version++;
    m->pack_dirty = true;
    return GNN_OK;
}
}   // namespace

extern "C" int gnn_loop_optimizer_step(gnn_loop *l, int kind, const float *hyper, float state_grad_scale, float bn_momentum_state,
                                       float bn_momentum_output)
{
    ARGCHK(l && hyper && (kind == 0 || kind == 1), "bad arguments (kind: 0 SGD, 1 Adam)");
    TrainCtx *cx = static_cast<TrainCtx *>(l->train_ctx);
    if (!cx || !cx->backward_done || cx->applied) return gnn_fail(GNN_ERR_STATE, "no fresh gradients: run gnn_loop_train_step (or forward + backward) first");
    ARGCHK(l->st->n_layers <= 16 && l->ou->n_layers <= 16, "too many layers");
    HIPCHK(hipSetDevice(l->device));
    hipStream_t st = l->stream;
    int rc = optimizer_apply(st, l->st, cx->ns, cx->k, kind, hyper, state_grad_scale, bn_momentum_state);
    if (!rc) rc = optimizer_apply(st, l->ou, cx->no_, cx->M > 0 ? 1 : 0, kind, hyper, 1.0f, bn_momentum_output);
    cx->applied = true;
    if (!rc) HIPCHK(hipStreamSynchronize(st));   // other loops (other streams) may use these weights next
    return rc;
}

extern "C" int gnn_loop_update_moving_statistics(gnn_loop *l, float bn_momentum_state, float bn_momentum_output)
{
    ARGCHK(l, "loop is NULL");
    TrainCtx *cx = static_cast<TrainCtx *>(l->train_ctx);
    if (!cx || cx->applied) return gnn_fail(GNN_ERR_STATE, "no training-mode forward pass to take the batch statistics from");
    HIPCHK(hipSetDevice(l->device));
    hipStream_t st = l->stream;
    struct { gnn_mlp *m; Net *net; int calls; float mom; } nets[2] = {{l->st, &cx->ns, cx->k, bn_momentum_state}, {l->ou, &cx->no_, cx->M > 0 ? 1 : 0, bn_momentum_output}};
    for (auto &n : nets) {
        if (!n.m->has_bn || n.calls <= 0) continue;
        const int F = n.m->dims.back();
        hipLaunchKernelGGL(k_bn_moving, cdiv(F, 64), 64, 0, st, F, n.calls, n.net->stats_all, n.mom, n.m->bn_raw);
        HIPCHK(hipGetLastError());
        int rc = gnn_mlp_refresh_bn(n.m, st);
        if (rc) return rc;
        n.m->version++;
        n.m->pack_dirty = true;
    }
    cx->applied = true;                           // once per forward pass
    HIPCHK(hipStreamSynchronize(st));
    return GNN_OK;
}

extern "C" int gnn_loop_arm_optimizer(gnn_loop *l, int kind, const float *hyper, int mean, float bn_momentum_state, float bn_momentum_output)
{
    ARGCHK(l && hyper && (kind == 0 || kind == 1), "bad arguments (kind: 0 SGD, 1 Adam)");
    ARGCHK(l->st->n_layers <= 16 && l->ou->n_layers <= 16, "too many layers");
    if (!l->train_arena) l->train_arena = new TrainArena();
    TrainArena *arena = static_cast<TrainArena *>(l->train_arena);
    arena->opt.armed = true; arena->opt.kind = kind; arena->opt.mean = mean != 0;
    for (int i = 0; i < 4; ++i) arena->opt.h[i] = hyper[i];
    arena->opt.mom_s = bn_momentum_state; arena->opt.mom_o = bn_momentum_output;
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
    ARGCHK(l->world == 1, "a training step is single-GPU (only the training-mode forward runs on shards: gnn_loop_train_forward)");
    ARGCHK(loss_kind >= 0 && loss_kind <= 2, "loss_kind: 0 categorical_crossentropy, 1 mean_squared_error, 2 categorical_crossentropy(from_logits=True)");
    const int64_t M = l->edge_mode ? l->n_edge_masked : l->g->n_masked;
    const int T = l->T;
    ARGCHK(!(l->edge_mode && n_graphs > 0), "an edge-based loop has no graph readout");
    ARGCHK(n_targets == (n_graphs > 0 ? n_graphs : M), "%lld target rows but %lld outputs", (long long)n_targets, (long long)(n_graphs > 0 ? n_graphs : M));
    ARGCHK(n_graphs <= 0 || (ng_indptr && ng_node && ng_w), "NodeGraph^T CSR required for a graph-based step");
    int rc = train_forward(l, src_indptr, src_dst, src_w, dropout_state, dropout_output, masks_state, masks_output, seed, bn_state, bn_output, k_out,
                           nullptr, false);
    if (rc) return rc;
    // loss and d loss / d out_nodes on the device, enqueued behind the forward pass: the step waits for the device once more, at its end
    TrainCtx *cx = static_cast<TrainCtx *>(l->train_ctx);
    TrainArena *arena = static_cast<TrainArena *>(l->train_arena);
    Buf &buf = cx->buf;
    hipStream_t st = l->stream;
    float *d_t = nullptr, *d_w = nullptr, *d_o = nullptr, *d_dnodes = nullptr;
    double *d_lp = nullptr;
    const int64_t nt = n_targets;
    const unsigned lblocks = nt ? cdiv(nt, 256) : 0;
    double loss = 0.0;
    double *h_lp = static_cast<double *>(arena->host(std::max<size_t>(sizeof(double) * lblocks, 4096)));
    if (!h_lp) return gnn_fail(GNN_ERR_HIP, "hipHostMalloc failed");
    // the step's small inputs, packed into pinned memory and uploaded in one transfer: targets | sample weights | NodeGraph^T CSR
    const int64_t ne = n_graphs > 0 ? ng_indptr[n_graphs] : 0;
    auto pad4 = [](size_t words) { return (words + 63) & ~(size_t)63; };
    const size_t o_t = 0, o_w = o_t + pad4((size_t)nt * T), o_ip = o_w + pad4((size_t)nt), o_nd = o_ip + pad4(n_graphs > 0 ? (size_t)n_graphs + 1 : 0),
                 o_nw = o_nd + pad4((size_t)ne), up_words = o_nw + pad4((size_t)ne);
    float *up = nullptr;
    if (up_words) {
        float *hs = static_cast<float *>(arena->stage(sizeof(float) * up_words));
        if (!hs) return gnn_fail(GNN_ERR_HIP, "hipHostMalloc failed");
        if ((rc = buf.get(&up, up_words))) return rc;
        if (nt) { memcpy(hs + o_t, targets, sizeof(float) * (size_t)nt * T); memcpy(hs + o_w, sample_weights, sizeof(float) * (size_t)nt); }
        if (n_graphs > 0) {
            memcpy(hs + o_ip, ng_indptr, sizeof(int32_t) * ((size_t)n_graphs + 1));
            if (ne) { memcpy(hs + o_nd, ng_node, sizeof(int32_t) * (size_t)ne); memcpy(hs + o_nw, ng_w, sizeof(float) * (size_t)ne); }
        }
        HIPCHK(hipMemcpyAsync(up, hs, sizeof(float) * up_words, hipMemcpyHostToDevice, st));
        d_t = up + o_t; d_w = up + o_w;
    }
    if (nt && ((rc = buf.get(&d_o, (size_t)nt * T)) || (rc = buf.get(&d_lp, (size_t)lblocks)))) return rc;
    if (n_graphs > 0) {                            // GNNgraphBased: out = NodeGraph^T . out_nodes (GNN.py:331-332)
        const int32_t *d_ip = reinterpret_cast<const int32_t *>(up + o_ip), *d_nd = reinterpret_cast<const int32_t *>(up + o_nd);
        const float *d_nw = up + o_nw;
        float *og = nullptr;
        if ((rc = buf.get(&og, (size_t)n_graphs * T)) || (rc = buf.get(&d_dnodes, (size_t)M * T))) return rc;
        HIPCHK(hipMemsetAsync(d_dnodes, 0, sizeof(float) * std::max<size_t>(1, (size_t)M * T), st));
        hipLaunchKernelGGL(k_graph_out, cdiv(n_graphs, 4), 256, 0, st, n_graphs, T, d_ip, d_nd, d_nw, cx->out_nodes, og);
        hipLaunchKernelGGL(k_loss_rows, lblocks, 256, 0, st, loss_kind, nt, T, d_t, og, d_w, d_o, d_lp);
        if (ne) hipLaunchKernelGGL(k_graph_out_bwd, cdiv(ne * T, 256), 256, 0, st, n_graphs, T, d_ip, d_nd, d_nw, d_o, d_dnodes);
        HIPCHK(hipGetLastError());
    } else if (nt) {
        hipLaunchKernelGGL(k_loss_rows, lblocks, 256, 0, st, loss_kind, nt, T, d_t, cx->out_nodes, d_w, d_o, d_lp);
        HIPCHK(hipGetLastError());
        d_dnodes = d_o;
    }
    if (lblocks) HIPCHK(hipMemcpyAsync(h_lp, d_lp, sizeof(double) * lblocks, hipMemcpyDeviceToHost, st));
    rc = train_backward(l, d_dnodes, nullptr, nullptr, grads_state, grads_output, bn_batch_state, bn_batch_output, nullptr, nullptr, false);
    if (rc) return rc;
    if (arena->opt.armed) {                        // gnn_loop_arm_optimizer: the update rides on this step's stream work
        arena->opt.armed = false;
        const float gscale = (arena->opt.mean && cx->k > 0) ? 1.0f / (float)cx->k : 1.0f;
        if ((rc = optimizer_apply(st, l->st, cx->ns, cx->k, arena->opt.kind, arena->opt.h, gscale, arena->opt.mom_s))) return rc;
        if ((rc = optimizer_apply(st, l->ou, cx->no_, cx->M > 0 ? 1 : 0, arena->opt.kind, arena->opt.h, 1.0f, arena->opt.mom_o))) return rc;
        cx->applied = true;
    }
    HIPCHK(hipStreamSynchronize(st));
    for (unsigned b = 0; b < lblocks; ++b) loss += h_lp[b];
    *loss_out = (float)loss;
    return GNN_OK;
}

