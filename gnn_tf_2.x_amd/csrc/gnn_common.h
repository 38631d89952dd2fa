// Shared declarations of the gfx950 engine (host side + device helpers).  Not part of the public ABI (include/gnn_hip.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <vector>

#include "gnn_hip.h"

int gnn_fail(int code, const char *fmt, ...);

// Every device allocation of the library goes through here.  Shipped build: hipMalloc.  Diagnostic build (make DIAG=1) with GNN_POISON=1
// in the environment: the block is filled with 0xFF bytes (a quiet NaN as float, -1 as int) and the fill has COMPLETED before the call
// returns, so a kernel that reads a word nobody wrote produces a NaN / a wild index instead of whatever the previous owner of the
// memory left there (gnn_engine.hip).  The same switch makes k_fused / k_small_loop / k_small16 write NaN over their whole LDS
// allocation before their first tile.
hipError_t gnn_dev_malloc(void **p, size_t bytes);
bool gnn_poison_enabled();

#define HIPCHK(expr)                                                                             \
    do {                                                                                         \
        hipError_t e_ = (expr);                                                                  \
        if (e_ != hipSuccess) return gnn_fail(GNN_ERR_HIP, "%s -> %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define ARGCHK(cond, ...)                                     \
    do {                                                      \
        if (!(cond)) return gnn_fail(GNN_ERR_ARG, __VA_ARGS__); \
    } while (0)

// ---------------------------------------------------------------------------------------------------------------------
// device math with a pinned evaluation order (mirrors oracle/gnn_oracle.c; the file is compiled with
// -ffp-contract=off, so only the explicit fmaf() calls fuse)
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ float gnn_expf(float x)
{
    // branch-free form of oracle/gnn_oracle.c:orc_expf: same operations on the in-range path, selects for the rest
    const float xc = __builtin_fminf(__builtin_fmaxf(x, -87.33654022216797f), 88.72283935546875f);
    const float u = xc * 1.44269504088896341f;
    const float n = __builtin_rintf(u);
    const float f = u - n;
    float p = 0.0013218672247603536f;
    p = __builtin_fmaf(p, f, 0.009671698324382305f);
    p = __builtin_fmaf(p, f, 0.05550893023610115f);
    p = __builtin_fmaf(p, f, 0.24022237956523895f);
    p = __builtin_fmaf(p, f, 0.6931468844413757f);
    p = __builtin_fmaf(p, f, 1.0f);
    float e = __builtin_ldexpf(p, (int)n);
    e = x > 88.72283935546875f ? __builtin_inff() : e;
    e = x < -87.33654022216797f ? 0.0f : e;
    return x != x ? x : e;
}

__device__ __forceinline__ float gnn_act(float v, int act)
{
    switch (act) {
    case GNN_ACT_RELU: return v > 0.0f ? v : 0.0f;
    case GNN_ACT_SELU: {
        float neg = 1.6732632423543772f * (gnn_expf(v) - 1.0f);
        return 1.0507009873554805f * (v > 0.0f ? v : neg);
    }
    case GNN_ACT_ELU: return v > 0.0f ? v : (gnn_expf(v) - 1.0f);
    case GNN_ACT_TANH: {
        float a = __builtin_fabsf(v);
        float t = gnn_expf(-2.0f * a);
        float q = __fdiv_rn(1.0f - t, 1.0f + t);
        return v < 0.0f ? -q : q;
    }
    case GNN_ACT_SIGMOID: return __fdiv_rn(1.0f, 1.0f + gnn_expf(-v));
    default: return v;
    }
}

// Iteration gate.  flags[k][rank][slot * GNN_FLAG_STRIDE] != 0 means "some node owned by `rank` had not converged when
// body k was about to run"; body k runs iff any slot of any rank is set.  Writers spread their atomicOr over
// GNN_FLAG_SLOTS words on separate 128-byte lines (slot = blockIdx & 15) so that ~10^4 workgroups do not queue on one
// address.
#define GNN_FLAG_SLOTS 16
#define GNN_FLAG_STRIDE 32
#define GNN_FLAG_WORDS (GNN_FLAG_SLOTS * GNN_FLAG_STRIDE)   // ints per (iteration, rank)
#define GNN_BODY_CHUNK 16

__device__ __forceinline__ bool gnn_gate_open(const int *gate, int world)
{
    if (!gate) return true;
    int any = 0;
    for (int p = 0; p < world * GNN_FLAG_SLOTS; ++p) any |= gate[p * GNN_FLAG_STRIDE];
    return any != 0;
}

// The same for a whole workgroup: the gate words are read once per block (one word per thread, OR through LDS) instead of once per
// thread - the elementwise kernels of the per-op path and of the exchanges have millions of threads.  Every thread of the block
// must call it (it contains barriers).
__device__ __forceinline__ bool gnn_gate_open_block(const int *gate, int world)
{
    if (!gate) return true;
    __shared__ int gate_any;
    if (threadIdx.x == 0) gate_any = 0;
    __syncthreads();
    int any = 0;
    for (int p = threadIdx.x; p < world * GNN_FLAG_SLOTS; p += blockDim.x) any |= gate[p * GNN_FLAG_STRIDE];
    if (any) gate_any = 1;
    __syncthreads();
    return gate_any != 0;
}

__device__ __forceinline__ void gnn_flag_raise(int *flag_rank_base)
{
    int *w = flag_rank_base + (blockIdx.x & (GNN_FLAG_SLOTS - 1)) * GNN_FLAG_STRIDE;
    if (__hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) atomicOr(w, 1);
}

// Certified gate of the split-arithmetic path (impl 2; reference GNN/GNN.py:206-220: `distance > threshold * norm`, reduce_any).
// The states of impl 2 differ from the bit-exact chain (impl 1) in the last bits, so a node whose test sits ON the threshold could
// be decided differently by the two - and with it k.  Besides the gate word (word 0 of its slot's line) the split path therefore
// raises, per body,
//     word 1  "some node moves ROBUSTLY":  distance >  threshold * norm + band
//     word 2  "some node is BORDERLINE":  |distance - threshold * norm| <= band,      band = ABS * norm + REL * threshold * norm.
// A gate is certified when a robust mover exists (open under either arithmetic) or no node is borderline (every node decided by a
// margin); k_finalize reports a gate that is neither, and the host then repeats that Loop on impl 1 and returns ITS k / state / output
// (run_loops).  The band is far wider than the measured state difference of the two paths near convergence (<= 2e-6 relative:
// tests/test_gpu_parity.py), and it only matters for the body at which a loop stops - where, by definition, nothing moves robustly.
#define GNN_BAND_ABS 1e-5f
#define GNN_BAND_REL 1e-3f
// a node is borderline when its test sits within the band of the threshold; a node whose old state AND movement are exactly zero (an isolated or
// dead node: norm 0, distance 0, band 0) is decided identically by every arithmetic and is NOT borderline
__device__ __forceinline__ bool gnn_gate_borderline(float root, float rhs, float band) { return band > 0.0f && __builtin_fabsf(root - rhs) <= band; }
// The same in two steps, for the hot path: peek() requests the three words early (the flags only ever go from 0 to 1, so a value read
// early is at worst a reason for a redundant OR), raise() decides at the end of the tile without waiting for memory.
struct GnnFlagPeek { int c0, c1, c2; };
__device__ __forceinline__ GnnFlagPeek gnn_flag_peek(int *flag_rank_base)
{
    int *w = flag_rank_base + (blockIdx.x & (GNN_FLAG_SLOTS - 1)) * GNN_FLAG_STRIDE;
    GnnFlagPeek p;
    p.c0 = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.c1 = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    p.c2 = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return p;
}
__device__ __forceinline__ void gnn_flag_raise_peeked(int *flag_rank_base, const GnnFlagPeek &p, bool moved, bool robust, bool border)
{
    int *w = flag_rank_base + (blockIdx.x & (GNN_FLAG_SLOTS - 1)) * GNN_FLAG_STRIDE;
    if (moved && p.c0 == 0) atomicOr(w, 1);
    if (robust && p.c1 == 0) atomicOr(w + 1, 1);
    if (border && p.c2 == 0) atomicOr(w + 2, 1);
}

__device__ __forceinline__ void gnn_flag_raise_certified(int *flag_rank_base, bool moved, bool robust, bool border)
{
    int *w = flag_rank_base + (blockIdx.x & (GNN_FLAG_SLOTS - 1)) * GNN_FLAG_STRIDE;
    if (!(moved || border)) return;                  // (robust implies moved)
    // the three words of the line are requested together (L1-bypassing loads, as gnn_flag_raise): one wait, then only the missing ORs
    const int c0 = __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c1 = __hip_atomic_load(w + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int c2 = __hip_atomic_load(w + 2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (moved && c0 == 0) atomicOr(w, 1);
    if (robust && c1 == 0) atomicOr(w + 1, 1);
    if (border && c2 == 0) atomicOr(w + 2, 1);
}

// ---------------------------------------------------------------------------------------------------------------------
// handles
// ---------------------------------------------------------------------------------------------------------------------
struct gnn_graph_shared {
    int refs = 1;
    int32_t *indptr = nullptr, *adj_src = nullptr, *masked_rows = nullptr;
    float *adj_w = nullptr, *arc_w = nullptr, *arc_labels = nullptr;
    uint8_t *mask = nullptr;
    int max_degree = 0;
    // Adjacency as CSR over SOURCE nodes (destinations ascending): operand of the transposed aggregation of the backward
    // pass; built on first use by gnn_train.hip when the caller passes no by-source arrays
    int32_t *src_indptr = nullptr, *src_dst = nullptr;
    float *src_w = nullptr;
    // arc-side LGNN relabelling (LGNN.py:253-254): arc id of every ArcNode^T entry and the arc labels in ORIGINAL arc order
    // (gnn_graph_set_arc_order)
    int32_t *arc_id = nullptr;
    float *arc_labels_orig = nullptr;
    // Adjacency of the WHOLE graph (CSR by destination over all N_global rows, global source ids): operand of the feature-sliced
    // exchange (gnn_loop_set_slice_exchange), where a rank aggregates its columns of the state for every node.  Shared with the
    // graphs derived from this one (LGNN layers > 0 on sliced shards use the same adjacency).
    int32_t *full_indptr = nullptr, *full_src = nullptr;
    float *full_w = nullptr;
    int64_t full_rows = 0;
};

struct gnn_graph {
    int device = 0;
    // N: size of the index space adj_src refers to = rows of the state replica / of `nodes` (all nodes, or own rows + halo
    // slots of a boundary-exchange shard); N_global: nodes of the whole graph; row_begin: GLOBAL id of the first owned row;
    // own_off: index of the first owned row inside that index space (== row_begin for full replicas, 0 for halo shards)
    int64_t N = 0, N_global = 0, row_begin = 0, own_off = 0, n_rows = 0, E = 0, n_masked = 0;
    int NL = 0, AL = 0;
    // boundary ("halo") exchange plan of a shard created by gnn_graph_create_halo, else halo_world == 0
    int halo_world = 0, halo_rank = 0;
    int64_t halo_block = 0, halo_count = 0;      // rows per rank block in the replica, boundary rows this rank sends
    int32_t *halo_send = nullptr;                // device [halo_count]: owned-row indices of the boundary rows, ascending (derived graphs: the base's array, see halo_send_owned)
    bool halo_send_owned = true;
    int64_t nodes_rows = 0;                      // rows allocated for `nodes` (>= N; derived graphs are padded for the all-gather)
    gnn_graph_shared *sh = nullptr;
    float *nodes = nullptr;   // [N, NL]
    int base_NL = 0;          // derived graphs: label width of the base graph
    // derived graphs of an edge-based LGNN own widened arc labels: ArcNode^T order (aggregation) and original order (readout)
    float *arc_labels_own = nullptr, *arc_labels_orig_own = nullptr;
    int base_AL = 0;
    uint64_t label_version = 1;   // bumped whenever node / arc labels are rewritten (gnn_graph_update_labels)
    // recorded behind the creation-time zero fills of a derived graph's label arrays (engine fill stream); streams that touch the
    // labels wait for it on the device (gnn_graph_wait_ready), host readers synchronise on it.  nullptr: nothing to wait for.
    hipEvent_t ready = nullptr;
};

inline const float *gnn_graph_arc_labels(const gnn_graph *g) { return g->arc_labels_own ? g->arc_labels_own : g->sh->arc_labels; }

struct gnn_mlp {
    int device = 0;
    int n_layers = 0;
    std::vector<int> dims, acts;
    std::vector<float *> W, b;          // device, Keras layout; slices of `slab` (256-byte aligned)
    float *slab = nullptr;              // all kernels and biases, in the order W1, b1, W2, b2, ...
    size_t slab_floats = 0;
    float *bn_scale = nullptr, *bn_shift = nullptr;   // device, [dims.back()]: the inference form of BatchNormalization
    float *bn_raw = nullptr;            // device, [4 * dims.back()]: gamma | beta | moving mean | moving variance
    // slots of the device-side optimizer (gnn_loop_optimizer_step), laid out like the gradient vector
    // (dW1, db1, ..., dgamma, dbeta); allocated on first use, zero at that point
    float *opt_a = nullptr, *opt_b = nullptr;
    bool has_bn = false;
    float eps = 1e-3f;
    // fused-kernel weight image (see gnn_fused.hip), rebuilt by set_weights
    float *packed = nullptr;
    size_t packed_floats = 0;
    int *packed_split = nullptr;        // bf16-piece weight image of the split-arithmetic fused kernel (impl 2)
    size_t packed_split_dwords = 0;
    int pack_nlc = -1;                  // node-label columns of the concat the split image was laid out for (alignment hole)
    bool pack_dirty = true;             // the images are rebuilt on the next fused use (training rewrites the weights every step)
    uint64_t version = 0;
};

// In-process loopback group (gnn_comm_create_loopback): `world` communicators on ONE device that share one stream; the
// exchange steps become device-to-device copies between the members' buffers.  It exists so that the sharded code path
// (row_begin > 0, padded replicas, per-rank flag slots, the exchange call sites) runs and is checked on a single GPU.
struct gnn_loop;
struct gnn_comm_group {
    int world = 1, refs = 0;
    hipStream_t stream = nullptr;
    std::vector<gnn_loop *> member;     // loop registered by each rank (gnn_loop_create), nullptr when none
};

struct gnn_comm {
    int rank = 0, world = 1, device = 0;
    void *nccl = nullptr;       // ncclComm_t (RCCL communicators)
    gnn_comm_group *grp = nullptr;   // loopback communicators
    hipStream_t stream = nullptr;
    hipStream_t xstream = nullptr;   // second stream: transfers that run beside the kernels of `stream` (sliced exchange, return all-to-all); created on first use
    double *scratch = nullptr;  // device, for allreduce_max
    // loops created on this communicator use its stream (and, loopback, its group): the communicator outlives them.  gnn_comm_destroy
    // with loops still alive only marks it closed; the last gnn_loop_destroy then releases it (either destruction order is safe).
    int loops = 0;
    bool closed = false;
};

struct gnn_loop {
    gnn_graph *g = nullptr;
    gnn_mlp *st = nullptr, *ou = nullptr;
    gnn_comm *comm = nullptr;
    int device = 0, rank = 0, world = 1;
    int D = 0, Ds = 0, NLc = 0, in_s = 0, wf = 0, T = 0, max_iter = 0;
    float thr = 0.f;
    int64_t shard_rows = 0, N_pad = 0, own_off = 0;   // rows per rank, rows of the state replica, replica row of the first owned row
    hipStream_t stream = nullptr;
    float *state[2] = {nullptr, nullptr};   // [N_pad, Ds] full replicas, ping-pong
    float *state_init = nullptr;            // [n_rows, Ds] initial state of the owned rows (D > 0)
    float *inp = nullptr;                   // unfused: materialised concat [n_rows, in_s]
    float *inv = nullptr;                   // fused: loop-invariant label block [n_rows, inv_w]
    uint64_t inv_version = 0;               // label_version of the graph the block was built from
    float *tmp[2] = {nullptr, nullptr};     // unfused: layer activations
    float *feats = nullptr, *out = nullptr, *otmp[2] = {nullptr, nullptr};
    int *flags = nullptr;                   // [(max_iter+2), world, GNN_FLAG_WORDS]
    int *tile_ctr = nullptr;                // fused path: one tile counter per iteration [max_iter + 1]
    int *kfinal_dev = nullptr, *kfinal_host = nullptr;   // device [4]: k, status word of the persistent loop, "a gate of this run is not certified" (impl 2), pad; host mirror [4]
    int certified_reruns = 0;               // Loops of this handle that were repeated on impl 1 because a gate of the impl-2 run was not certified
    bool last_run_rerun = false;
    bool small_words_clean = false;         // the double-buffered gate words of the persistent loop are zero / in their run-parity state
    unsigned small_runs = 0;
    float *small_xs = nullptr;              // the persistent loop's padded exchange rows (gnn_small.hip), allocated with its first run
    size_t small_xs_floats = 0;
    bool small_disabled = false;            // the persistent small-graph loop gave up once on this loop: keep to per-body launches
    int kfinal = -1;
    bool have_state0 = false, ran = false;
    bool graph_ready_seen = false;          // this loop's stream has waited for the graph's creation-time fills
    int impl_req = 1, impl_used = 0;
    int tile_form = 0;                      // gnn_loop_set_tile_form: 0 library's choice, 1 one wave per tile, 2 wave pair per tile
    int32_t *ng_ip = nullptr, *ng_node = nullptr;   // cached NodeGraph^T (graph readout)
    float *ng_w = nullptr, *ng_out = nullptr, *ng_part = nullptr;   // ng_part [world, G, T]: per-rank partial readouts
    std::vector<int32_t> ng_key;
    std::vector<float> ng_w_host;
    // graph readout folded into the persistent small-graph launch (gnn_small.hip): result [G, T] in pinned host memory, valid for the last run
    float *ng_host = nullptr;
    int ng_G = 0, ng_host_floats = 0;
    bool ng_inlaunch = false;
    // l->out is rewritten by every run - inference (loop_prepare) AND training (train_forward): the folded readout in ng_host is only valid
    // for the run whose number it carries (gnn_loop_ng_folded)
    uint64_t out_runs = 0, ng_inlaunch_run = 0;
    // edge-based readout (GNNedgeBased.apply_filters): entry -> CSR row, arc labels in original order, masked arc list
    bool edge_mode = false, edge_expected = false;
    int32_t *edge_dst = nullptr, *edge_rows = nullptr;
    float *edge_labels = nullptr;
    int64_t n_edge_masked = 0;
    // incidence of the masked arcs by node (training backward): node -> (masked arc q << 1 | side), ascending q; side 0: the arc's
    // destination half of the readout features, 1: its source half.  Lets the backward pass GATHER per node in a fixed order (no atomics).
    int32_t *edge_inc_ptr = nullptr, *edge_inc = nullptr;
    int *gate_host = nullptr;               // pinned copy of one gate (early-exit check every GNN_BODY_CHUNK bodies)
    bool profiling = false;
    std::vector<hipEvent_t> ev;
    hipEvent_t ev_total[2] = {nullptr, nullptr};
    float total_ms = 0.f, avg_iter_ms = 0.f;
    float avg_gap_ms = 0.f;                 // profiling: mean time on the stream BETWEEN the end of body k's kernel(s) and the start of body k + 1's: the exchange (+ pack / aggregate / unpack of the sliced layout)
    int n_iter_timed = 0;
    // feature-sliced exchange (gnn_loop_set_slice_exchange): rank q aggregates columns [q Cs, (q + 1) Cs) of the state for ALL nodes
    bool slice_mode = false;
    int Cs = 0;                             // Ds / world
    float *sl_send = nullptr;               // [world][shard_rows, Cs]: column slices of the owned state rows, one block per destination
    float *sl_state = nullptr;              // [N_pad, Cs]: this rank's column slice of every node's state
    float *sl_agg = nullptr;                // [N_pad, Cs]: its aggregate
    float *sl_recv = nullptr;               // [world][shard_rows, Cs]: the aggregate of the owned rows, one block per source rank
    bool sl_pipeline = true;                // the return all-to-all runs block by block on the communicator's second stream, under the aggregation of the next block
    std::vector<hipEvent_t> sl_ev;          // [world]: block t of the aggregation is complete (recorded on `stream`)
    hipEvent_t sl_done = nullptr;           // this rank's transfers of the body are complete (recorded on the second stream)
    float *agg_own = nullptr;               // [shard_rows, Ds]: aggregated states of the owned rows (GNN.py:234), input of the body
    void *train_ctx = nullptr;              // gnn_train.hip: what train_forward leaves for train_backward
    void *train_arena = nullptr;            // gnn_train.hip: device scratch slabs kept from step to step
    int train_k_hint = 0;                   // bodies the last training forward of this loop ran (the next one enqueues that many + 1 before it looks at the gates)
};

// the graph readout of the loop's LAST run was computed inside that run's persistent launch (result in ng_host)
inline bool gnn_loop_ng_folded(const gnn_loop *l) { return l->ng_inlaunch && l->ng_inlaunch_run == l->out_runs; }

// gnn_engine.hip
int gnn_graph_wait_ready(const gnn_graph *g, hipStream_t st);
int gnn_launch_spmm(hipStream_t st, int64_t n_rows, const int32_t *indptr, const int32_t *idx, const float *w, const float *X,
                    int width, int64_t ldx, float *out, int64_t ldo, const int *gate, int world);

int gnn_launch_dense(hipStream_t st, int64_t n, int n_in, int n_out, const float *X, int64_t ldx, const float *W, const float *b,
                     int act, float *Y, int64_t ldy);
int gnn_launch_check(hipStream_t st, int64_t n_rows, int d, const float *s, const float *so, float thr, int *flag_rank_base);
// all-gather of `count` 4-byte elements per rank on an RCCL communicator (one process per rank; recv: [world][count], in place when
// send == recv + rank * count), queued on st - for the translation units that do not see the RCCL table (gnn_train.hip)
int gnn_comm_allgather32(gnn_comm *c, const void *send, void *recv, size_t count, hipStream_t st);
// GNNedgeBased.apply_filters on `state` (training path): feats [n_edge_masked, 2 (Ds + NLc) + AL]
int gnn_launch_feats_edge(hipStream_t st, const gnn_loop *l, const float *state, float *feats);

int gnn_launch_copy_cols(hipStream_t st, int64_t n_rows, int w, const float *src, int64_t lds_, float *dst, int64_t ldd, const int *gate, int world);

int gnn_mlp_refresh_bn(gnn_mlp *m, hipStream_t st);
// gnn_train.hip
void gnn_train_ctx_free(gnn_loop *l);
void gnn_train_arena_free(gnn_loop *l);

// gnn_fused.hip
bool gnn_fused_supported(const gnn_loop *l);
bool gnn_fused_pair_selected(const gnn_loop *l);   // the default path's bodies run as k_fused_pair (wave pair per tile) rather than k_fused
int gnn_fused_prepare(gnn_loop *l);
int gnn_fused_pack(gnn_mlp *m, int nlc);
int gnn_fused_iteration(gnn_loop *l, int k);
bool gnn_small_supported(const gnn_loop *l);   // persistent small-graph loop (gnn_small.hip): all bodies in one launch
int gnn_small_run(gnn_loop *l, bool *output_done);
void gnn_fused_release(gnn_mlp *m);
