// Device-side graph build: arc list (COO) -> the transposed, row-major reordered sparse operands of the loop.
//
// Replaces, for graphs where minutes of Python would be spent on it, the host chain
//   GraphObject.buildArcNode / buildAdiacency           reference GNN/graph_class.py:90-121
//   GraphTensor.COO2SparseTransposedTensor              reference GNN/graph_class.py:365-372 (transpose + tf.sparse.reorder)
// with two stable radix sorts (hipCUB) and a histogram:
//   Adjacency^T  entries of destination d ordered by source   = arcs sorted by the key (dst, src)
//   ArcNode^T    entries of destination d ordered by arc id   = arc ids stably sorted by dst
//   indptr       exclusive scan of the in-degree histogram (shared by both: same entries per destination)
//   value of arc a: 1 ('sum'), float(1 / n_arcs) ('normalized', the reference divides by the number of ARCS),
//                   float(1 / indegree(dst(a))) ('average'); the division is done in double like NumPy's.
#include <hipcub/hipcub.hpp>
#include <string.h>

#include <algorithm>
#include <vector>

#include "gnn_common.h"

namespace {

inline unsigned cdiv(int64_t a, int64_t b) { return (unsigned)((a + b - 1) / b); }

__global__ void k_keys(int64_t e, const int32_t *src, const int32_t *dst, unsigned long long *key, int32_t *id, int32_t *indeg)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= e) return;
    key[a] = ((unsigned long long)(uint32_t)dst[a] << 32) | (uint32_t)src[a];
    id[a] = (int32_t)a;
    atomicAdd(indeg + dst[a], 1);
}

__global__ void k_values(int64_t e, const int32_t *dst, const int32_t *indeg, int mode, float *w)
{
    const int64_t a = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (a >= e) return;
    float v = 1.0f;
    if (mode == 1) v = (float)(1.0 / (double)e);
    else if (mode == 2) v = (float)(1.0 / (double)indeg[dst[a]]);
    w[a] = v;
}

// entry q of the sorted order is arc perm[q]
__global__ void k_permute_adj(int64_t e, const int32_t *perm, const int32_t *src, const float *w, int32_t *adj_src, float *adj_w)
{
    const int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= e) return;
    const int32_t a = perm[q];
    adj_src[q] = src[a];
    adj_w[q] = w[a];
}

__global__ void k_permute_arc(int64_t e, int AL, const int32_t *perm, const float *w, const float *labels, float *arc_w, float *arc_labels)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int width = AL > 0 ? AL : 1;
    if (t >= e * width) return;
    const int64_t q = t / width;
    const int c = (int)(t - q * width);
    const int32_t a = perm[q];
    if (c == 0) arc_w[q] = w[a];
    if (AL > 0) arc_labels[q * AL + c] = labels[(int64_t)a * AL + c];
}

struct Scratch {
    std::vector<void *> p;
    template <typename T>
    int get(T **out, size_t count)
    {
        *out = nullptr;
        if (gnn_dev_malloc((void **)out, std::max<size_t>(1, count) * sizeof(T)) != hipSuccess) return gnn_fail(GNN_ERR_HIP, "hipMalloc of %zu bytes failed", count * sizeof(T));
        p.push_back(*out);
        return GNN_OK;
    }
    ~Scratch() { for (void *q : p) (void)hipFree(q); }
};

template <typename T>
int keep(T **dst, size_t count)
{
    *dst = nullptr;
    if (gnn_dev_malloc((void **)dst, std::max<size_t>(1, count) * sizeof(T)) != hipSuccess) return gnn_fail(GNN_ERR_HIP, "hipMalloc of %zu bytes failed", count * sizeof(T));
    return GNN_OK;
}

}   // namespace

extern "C" int gnn_graph_create_from_arcs(int64_t n_nodes, int64_t n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                                          const float *arc_labels, int dim_arc_label, int aggregation_mode, const float *nodes,
                                          int dim_node_label, const uint8_t *mask, int device, gnn_graph **out,
                                          int32_t *indptr_out, int32_t *adj_src_out, float *adj_w_out, int32_t *arc_id_out,
                                          float *arc_w_out)
{
    ARGCHK(out, "out is NULL");
    *out = nullptr;
    ARGCHK(n_nodes > 0 && n_nodes < (int64_t)1 << 31, "n_nodes=%lld out of range", (long long)n_nodes);
    ARGCHK(n_arcs >= 0 && n_arcs < (int64_t)1 << 31, "n_arcs=%lld out of range", (long long)n_arcs);
    ARGCHK(dim_node_label > 0 && dim_arc_label >= 0, "label dims must be NL>0, AL>=0");
    ARGCHK(aggregation_mode >= 0 && aggregation_mode <= 2, "aggregation_mode: 0 sum, 1 normalized, 2 average");   // graph_class.py:86
    ARGCHK(nodes && mask, "nodes/mask are required");
    ARGCHK(n_arcs == 0 || (arc_src && arc_dst && (arc_labels || dim_arc_label == 0)), "arc arrays are required");
    for (int64_t a = 0; a < n_arcs; ++a)
        ARGCHK(arc_src[a] >= 0 && arc_src[a] < n_nodes && arc_dst[a] >= 0 && arc_dst[a] < n_nodes, "arc %lld = (%d, %d) outside [0,%lld)",
               (long long)a, arc_src[a], arc_dst[a], (long long)n_nodes);
    HIPCHK(hipSetDevice(device));
    const int64_t N = n_nodes, E = n_arcs;
    const int AL = dim_arc_label;
    Scratch tmp;
    int32_t *d_src = nullptr, *d_dst = nullptr, *d_id = nullptr, *d_perm1 = nullptr, *d_perm2 = nullptr, *d_indeg = nullptr, *d_dst_sorted = nullptr;
    unsigned long long *d_key = nullptr, *d_key_sorted = nullptr;
    float *d_lab = nullptr, *d_w = nullptr;
    int rc;
    if ((rc = tmp.get(&d_src, E)) || (rc = tmp.get(&d_dst, E)) || (rc = tmp.get(&d_id, E)) || (rc = tmp.get(&d_perm1, E)) ||
        (rc = tmp.get(&d_perm2, E)) || (rc = tmp.get(&d_indeg, N + 1)) || (rc = tmp.get(&d_dst_sorted, E)) || (rc = tmp.get(&d_key, E)) ||
        (rc = tmp.get(&d_key_sorted, E)) || (rc = tmp.get(&d_lab, (size_t)E * AL)) || (rc = tmp.get(&d_w, E)))
        return rc;
    if (E) {
        HIPCHK(hipMemcpy(d_src, arc_src, sizeof(int32_t) * E, hipMemcpyHostToDevice));
        HIPCHK(hipMemcpy(d_dst, arc_dst, sizeof(int32_t) * E, hipMemcpyHostToDevice));
        if (AL) HIPCHK(hipMemcpy(d_lab, arc_labels, sizeof(float) * E * AL, hipMemcpyHostToDevice));
    }
    HIPCHK(hipMemset(d_indeg, 0, sizeof(int32_t) * (N + 1)));

    gnn_graph *g = new gnn_graph();
    g->device = device; g->N = N; g->N_global = N; g->nodes_rows = N; g->row_begin = 0; g->own_off = 0; g->n_rows = N; g->E = E;
    g->NL = dim_node_label; g->AL = AL; g->base_NL = dim_node_label; g->base_AL = AL;
    g->sh = new gnn_graph_shared();
    gnn_graph_shared *sh = g->sh;
    auto fail = [&](int code) { gnn_graph_destroy(g); return code; };
    // from here on a failing HIP call releases the half-built handle (HIPCHK alone would leak it)
#define HIPCHK_G(expr)                                                                                         \
    do {                                                                                                       \
        hipError_t e_ = (expr);                                                                                \
        if (e_ != hipSuccess) return fail(gnn_fail(GNN_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)));    \
    } while (0)
    if ((rc = keep(&sh->indptr, N + 1)) || (rc = keep(&sh->adj_src, E)) || (rc = keep(&sh->adj_w, E)) || (rc = keep(&sh->arc_w, E)) ||
        (rc = keep(&sh->arc_labels, (size_t)E * AL)))
        return fail(rc);

    if (E) {
        hipLaunchKernelGGL(k_keys, cdiv(E, 256), 256, 0, 0, E, d_src, d_dst, d_key, d_id, d_indeg);
        hipLaunchKernelGGL(k_values, cdiv(E, 256), 256, 0, 0, E, d_dst, d_indeg, aggregation_mode, d_w);
    }
    // indptr = exclusive scan of the in-degrees (the extra last element makes indptr[N] = E)
    {
        size_t bytes = 0;
        if (hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, d_indeg, sh->indptr, (int)(N + 1)) != hipSuccess) return fail(gnn_fail(GNN_ERR_HIP, "hipcub scan sizing failed"));
        void *ws = nullptr;
        if ((rc = tmp.get((char **)&ws, bytes))) return fail(rc);
        if (hipcub::DeviceScan::ExclusiveSum(ws, bytes, d_indeg, sh->indptr, (int)(N + 1)) != hipSuccess) return fail(gnn_fail(GNN_ERR_HIP, "hipcub scan failed"));
    }
    if (E) {
        int node_bits = 1;
        while (((int64_t)1 << node_bits) < N) ++node_bits;
        size_t b1 = 0, b2 = 0;
        // Adjacency^T order: (dst, src) ascending; ArcNode^T order: dst ascending, arc id ascending (the sort is stable)
        if (hipcub::DeviceRadixSort::SortPairs(nullptr, b1, d_key, d_key_sorted, d_id, d_perm1, (int)E, 0, 32 + node_bits) != hipSuccess ||
            hipcub::DeviceRadixSort::SortPairs(nullptr, b2, d_dst, d_dst_sorted, d_id, d_perm2, (int)E, 0, node_bits) != hipSuccess)
            return fail(gnn_fail(GNN_ERR_HIP, "hipcub sort sizing failed"));
        void *ws = nullptr;
        if ((rc = tmp.get((char **)&ws, std::max(b1, b2)))) return fail(rc);
        if (hipcub::DeviceRadixSort::SortPairs(ws, b1, d_key, d_key_sorted, d_id, d_perm1, (int)E, 0, 32 + node_bits) != hipSuccess ||
            hipcub::DeviceRadixSort::SortPairs(ws, b2, d_dst, d_dst_sorted, d_id, d_perm2, (int)E, 0, node_bits) != hipSuccess)
            return fail(gnn_fail(GNN_ERR_HIP, "hipcub sort failed"));
        hipLaunchKernelGGL(k_permute_adj, cdiv(E, 256), 256, 0, 0, E, d_perm1, d_src, d_w, sh->adj_src, sh->adj_w);
        hipLaunchKernelGGL(k_permute_arc, cdiv(E * (AL > 0 ? AL : 1), 256), 256, 0, 0, E, AL, d_perm2, d_w, d_lab, sh->arc_w, sh->arc_labels);
        HIPCHK_G(hipGetLastError());
    }
    HIPCHK_G(hipStreamSynchronize(nullptr));        // everything above ran on the null stream, in order: no device-wide wait

    // what stays on the host side of the handle: the row list of the mask and the maximum in-degree
    std::vector<int32_t> indptr((size_t)N + 1);
    HIPCHK_G(hipMemcpy(indptr.data(), sh->indptr, sizeof(int32_t) * (N + 1), hipMemcpyDeviceToHost));
    if (indptr[N] != E) return fail(gnn_fail(GNN_ERR_HIP, "device build: indptr[N]=%d but %lld arcs", indptr[N], (long long)E));
    int maxdeg = 0;
    for (int64_t r = 0; r < N; ++r) maxdeg = std::max(maxdeg, indptr[r + 1] - indptr[r]);
    sh->max_degree = maxdeg;
    std::vector<int32_t> rows, pos((size_t)N);
    for (int64_t r = 0; r < N; ++r) { pos[r] = (int32_t)rows.size(); if (mask[r]) rows.push_back((int32_t)r); }
    g->n_masked = (int64_t)rows.size();
    std::vector<int32_t> both(rows);
    both.insert(both.end(), pos.begin(), pos.end());
    if ((rc = keep(&sh->mask, N)) || (rc = keep(&sh->masked_rows, both.size())) || (rc = keep(&g->nodes, (size_t)N * dim_node_label))) return fail(rc);
    HIPCHK_G(hipMemcpy(sh->mask, mask, (size_t)N, hipMemcpyHostToDevice));
    HIPCHK_G(hipMemcpy(sh->masked_rows, both.data(), sizeof(int32_t) * both.size(), hipMemcpyHostToDevice));
    HIPCHK_G(hipMemcpy(g->nodes, nodes, sizeof(float) * (size_t)N * dim_node_label, hipMemcpyHostToDevice));

    // host mirrors for the caller (GraphTensor keeps them like the reference keeps its SparseTensors); any may be NULL
    if (indptr_out) memcpy(indptr_out, indptr.data(), sizeof(int32_t) * (N + 1));
    if (E) {
        if (adj_src_out) HIPCHK_G(hipMemcpy(adj_src_out, sh->adj_src, sizeof(int32_t) * E, hipMemcpyDeviceToHost));
        if (adj_w_out) HIPCHK_G(hipMemcpy(adj_w_out, sh->adj_w, sizeof(float) * E, hipMemcpyDeviceToHost));
        if (arc_id_out) HIPCHK_G(hipMemcpy(arc_id_out, d_perm2, sizeof(int32_t) * E, hipMemcpyDeviceToHost));
        if (arc_w_out) HIPCHK_G(hipMemcpy(arc_w_out, sh->arc_w, sizeof(float) * E, hipMemcpyDeviceToHost));
    }
    *out = g;
    return GNN_OK;
#undef HIPCHK_G
}
