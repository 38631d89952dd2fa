/*
 * gnn_hip.h - C ABI of the MI355X (gfx950) engine for the GNN fixed-point state-propagation loop.
 *
 * The reference (sailab-code/GNN_tf_2.x) has no FFI / plugin boundary: the path is Python calling TensorFlow ops
 * (SURVEY.md 8b).  The entry points below are what a binding for that path needs; each one names the reference
 * interface it replaces (paths relative to the reference root).  The reference-side ctypes stub is in INTEGRATION.md;
 * the Python mirror of the reference classes lives in gnn_tf_2.x_amd/GNN/.
 *
 * Conventions
 *   - every function returns 0 on success, a negative gnn_status otherwise; gnn_last_error() gives the text
 *     (thread-local, valid until the next failing call on the thread).  Nothing throws or aborts across the ABI.
 *   - host buffers are caller-owned, C-contiguous, float32 / int32 / uint8, and only read during the call;
 *     outputs are written into caller-allocated buffers.
 *   - device memory is owned by the handles (create .. destroy).  Handles are not thread-safe; different handles may
 *     be used from different host threads.  All work of a loop runs on that loop's HIP stream.
 *   - sparse operands are the TRANSPOSED, row-major reordered matrices of GraphTensor (GNN/graph_class.py:355-372)
 *     in CSR form "by destination node": one indptr shared by Adjacency^T (inner order: ascending source id) and
 *     ArcNode^T (inner order: ascending arc id).
 */
#ifndef GNN_HIP_H
#define GNN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    GNN_OK = 0,
    GNN_ERR_ARG = -1,      /* bad argument / inconsistent dimensions (reference: ValueError / TypeError) */
    GNN_ERR_HIP = -2,      /* HIP runtime error (no device, out of memory, launch failure) */
    GNN_ERR_STATE = -3,    /* call order (e.g. fetching results before gnn_loop_run) */
    GNN_ERR_COMM = -4,     /* RCCL error */
    GNN_ERR_UNSUPPORTED = -5
} gnn_status;

/* activation codes of Dense layers built by GNN/MLP.py:11-64 (Keras names) */
typedef enum {
    GNN_ACT_LINEAR = 0, GNN_ACT_RELU = 1, GNN_ACT_SELU = 2, GNN_ACT_ELU = 3,
    GNN_ACT_TANH = 4, GNN_ACT_SIGMOID = 5, GNN_ACT_SOFTMAX = 6
} gnn_activation;

typedef struct gnn_graph gnn_graph;   /* device-resident GraphTensor (GNN/graph_class.py:330-372) */
typedef struct gnn_mlp gnn_mlp;       /* device-resident Sequential built by GNN/MLP.py:11-64 */
typedef struct gnn_loop gnn_loop;     /* one configured GNN.Loop (GNN/GNN.py:251-280) with its workspaces */
typedef struct gnn_comm gnn_comm;     /* RCCL communicator for node-range sharding (no reference counterpart) */

const char *gnn_last_error(void);
int gnn_version(void);                               /* ABI version, currently 1 */
int gnn_device_count(int *count);                    /* number of visible HIP devices (0 is not an error) */
int gnn_device_synchronize(int device);

/* ---- graph ------------------------------------------------------------------------------------------------------
 * Replaces GraphTensor.__init__ / fromGraphObject (GNN/graph_class.py:331-363).
 *   n_nodes            N, global node count
 *   row_begin, n_rows  destination rows owned by this handle ([0, N) unless sharded across GPUs)
 *   indptr[n_rows+1]   CSR row pointers of the owned rows, indptr[0] == 0
 *   adj_src, adj_w     Adjacency^T entries of the owned rows: global source id and weight, ascending source id
 *   arc_w, arc_labels  ArcNode^T entries of the owned rows in ascending arc id order: weight and the arc's label
 *                      row arcs[arc, 2:] (already permuted into this order), [n_arcs, dim_arc_label]
 *   nodes              node labels of ALL N nodes [N, dim_node_label] (neighbour labels are aggregated, GNN.py:263)
 *   mask               set_mask & output_mask of the owned rows (GNN.py:275), uint8 [n_rows]
 */
int gnn_graph_create(int64_t n_nodes, int64_t row_begin, int64_t n_rows, int64_t n_arcs, const int32_t *indptr,
                     const int32_t *adj_src, const float *adj_w, const float *arc_w, const float *arc_labels,
                     int dim_arc_label, const float *nodes, int dim_node_label, const uint8_t *mask, int device,
                     gnn_graph **out);
/* Builds the same graph from its arc list on the device (replaces GraphObject.buildArcNode / buildAdiacency and
 * GraphTensor.COO2SparseTransposedTensor, reference GNN/graph_class.py:90-121, :365-372, for graphs where the host build
 * would take minutes): arc a goes from arc_src[a] to arc_dst[a] with labels arc_labels[a, :] (ORIGINAL arc order);
 * aggregation_mode 0 = 'sum', 1 = 'normalized' (1 / number of arcs), 2 = 'average' (1 / in-degree of the destination).
 * Two stable radix sorts give Adjacency^T (entries of a destination by ascending source) and ArcNode^T (by ascending arc
 * id).  Single GPU: the handle owns all rows.  The optional outputs (any may be NULL) return the host mirrors:
 * indptr [n_nodes + 1], adj_src / adj_w [n_arcs] in Adjacency^T order, arc_id / arc_w [n_arcs] in ArcNode^T order. */
int gnn_graph_create_from_arcs(int64_t n_nodes, int64_t n_arcs, const int32_t *arc_src, const int32_t *arc_dst,
                               const float *arc_labels, int dim_arc_label, int aggregation_mode, const float *nodes,
                               int dim_node_label, const uint8_t *mask, int device, gnn_graph **out,
                               int32_t *indptr_out, int32_t *adj_src_out, float *adj_w_out, int32_t *arc_id_out,
                               float *arc_w_out);
/* LGNN.update_graph (GNN/LGNN.py:227-260) for node/graph-based layers, on the device:
 * dst.nodes <- [base.nodes | state of `from` (if get_state) | scatter(mask, output of `from`) (if get_output)].
 * `dst` must have been created by gnn_graph_derive(base, extra) with extra = get_state*Ds + get_output*T. */
int gnn_graph_derive(const gnn_graph *base, int extra_node_label_dims, gnn_graph **out);
/* Sharded graphs: every rank relabels its own rows, then the new label rows are exchanged - whole shards (full-replica shards) or
 * the rank's boundary rows into its block (gnn_graph_create_halo shards) - RCCL communicator: call on every rank; loopback group:
 * gnn_graph_update_labels_group below.  A derived graph's labels are zero until then; that fill is ordered on the device before
 * whatever touches the labels first (no call waits for it on the host). */
int gnn_graph_update_labels(gnn_graph *dst, const gnn_graph *base, const gnn_loop *from, int get_state, int get_output);
/* Edge-based LGNN (reference GNN/LGNN.py:253-254: the output of an edge-based layer widens the ARC labels, its state the
 * node labels).  gnn_graph_set_arc_order gives the original graph what the arc side needs: arc_id [n_arcs] = arc of every
 * ArcNode^T entry (the permutation COO2SparseTransposedTensor applied) and the arc labels in ORIGINAL arc order.
 * gnn_graph_derive_edge is gnn_graph_derive with extra arc-label columns; gnn_graph_update_labels then fills node labels
 * [base | state?] and arc labels [base | scatter(out) over the arc mask?] when `from` is an edge-based loop, and loops
 * created on the derived graph read their per-arc readout labels from it (gnn_loop_set_edge_readout: arc_labels = NULL). */
int gnn_graph_set_arc_order(gnn_graph *g, const int32_t *arc_id, const float *arc_labels_orig);
int gnn_graph_derive_edge(const gnn_graph *base, int extra_nodes, int extra_arcs, gnn_graph **out);
int gnn_graph_get_nodes(const gnn_graph *g, float *nodes_out /* [N, dim_node_label] */);
int gnn_graph_dims(const gnn_graph *g, int64_t *n_nodes, int64_t *n_rows, int64_t *n_arcs, int *dim_node_label,
                   int *dim_arc_label, int64_t *n_masked);
int gnn_graph_destroy(gnn_graph *g);

/* ---- MLP --------------------------------------------------------------------------------------------------------
 * Replaces the Keras Sequential of GNN/MLP.py:62-64 at inference: Dense(act) x n_layers [+ BatchNormalization].
 * Weights in Keras get_weights() layout (GNN/GNN.py:163-165): W[l] row-major [dims[l], dims[l+1]], b[l] [dims[l+1]];
 * bn = [gamma | beta | moving_mean | moving_variance], each dims[n_layers] long, or NULL without BatchNormalization. */
int gnn_mlp_create(int n_layers, const int32_t *dims, const int32_t *acts, const float *const *W, const float *const *b,
                   const float *bn, float bn_eps, int device, gnn_mlp **out);
int gnn_mlp_set_weights(gnn_mlp *m, const float *const *W, const float *const *b, const float *bn);   /* GNN.py:168-172 */
int gnn_mlp_forward(gnn_mlp *m, int64_t n_rows, const float *x, float *y);
/* Sequential.get_weights(): W[l] [dims[l] x dims[l+1]], b[l] [dims[l+1]], bn [4 x width] = gamma | beta | moving_mean |
 * moving_variance (NULL without BatchNormalization); waits for the device first (a device-side optimizer step may be running) */
int gnn_mlp_get_weights(gnn_mlp *m, float *const *W, float *const *b, float *bn);
/* forget the slots (Adam moments / SGD velocity) of the device-side optimizer: a NEW optimizer object starts from zero, as a
 * tf.keras optimizer creates its slot variables on first use (reference starter.py:81) */
int gnn_mlp_reset_optimizer(gnn_mlp *m);   /* Sequential.__call__(x, training=False) */
int gnn_mlp_destroy(gnn_mlp *m);

/* ---- loop -------------------------------------------------------------------------------------------------------
 * Replaces GNNnodeBased.Loop (GNN/GNN.py:251-280): condition :202-220, convergence :223-242, apply_filters :245-248.
 *   state_dim      state_vect_dim (0: state is initialised with the node labels, GNN.py:265)
 *   max_iter, thr  max_iteration, state_threshold (GNN.py:61-62)
 * gnn_loop_set_state0: injected initial state [n_rows owned, state_dim] (the reference draws tf.random.normal(stddev=0.1),
 * GNN.py:262, whose stream cannot be reproduced); NULL draws N(0, 0.1^2) from the engine's own counter RNG with `seed`.
 * gnn_loop_run: runs the whole loop on the device.  Bodies are enqueued 16 at a time without waiting for them (a body whose gate is
 * closed returns at once); after each 16 the host reads the next body's gate so that a converged loop stops enqueuing, and it
 * synchronises once at the end.  Small graphs (all tiles resident, nets <= 32 wide) take ONE persistent launch for the whole loop,
 * the output stage and - once a NodeGraph is cached with the loop by an earlier gnn_loop_readout - the graph readout.  *k_out = number of executed iterations as float (GNN.py:267).  This is the
 * inference Loop (training=False); training != 0 is GNN_ERR_UNSUPPORTED here: the training-mode Loop and its backward pass
 * are gnn_loop_train_forward / gnn_loop_train_backward / gnn_loop_train_step below.
 */
int gnn_loop_create(gnn_graph *g, gnn_mlp *net_state, gnn_mlp *net_output, int state_dim, int max_iter, float threshold,
                    gnn_comm *comm /* NULL: single GPU */, gnn_loop **out);
int gnn_loop_set_state0(gnn_loop *l, const float *state0, uint64_t seed);
int gnn_loop_run(gnn_loop *l, int training, float *k_out);
/* n independent loops (the batches of a dataset, which GNN_BaseClass.evaluate - reference GNN_BaseClass.py:165-189 - runs one after the
 * other) in one call: the persistent launches of small graphs are all queued, each on its loop's stream, before the first is waited
 * for, and run side by side on the GPU; larger graphs run one after the other.  k_out [n]; results as of n gnn_loop_run calls. */
int gnn_loop_run_many(gnn_loop **loops, int n, float *k_out);
int gnn_loop_get_state(const gnn_loop *l, float *state_out /* [n_rows, Ds] */);
int gnn_loop_get_output(const gnn_loop *l, float *out /* [n_masked, T] */, int64_t *n_masked);
/* GNNgraphBased.Loop readout (GNN/GNN.py:331-332, LGNN.py:278): out_graph = NodeGraph^T . out_nodes.
 * NodeGraph^T is passed in CSR form over graphs: ng_indptr[G+1], ng_node (ascending), ng_w.  The arrays are kept with the loop (and
 * compared on every call): when the persistent small-graph launch of the last run has already computed the readout for the same
 * NodeGraph, the call only copies the result. */
int gnn_loop_readout(const gnn_loop *l, int n_graphs, const int32_t *ng_indptr, const int32_t *ng_node,
                     const float *ng_w, float *out_graph /* [G, T] */);
/* GNNedgeBased.apply_filters (GNN/GNN.py:289-302): switches the output stage of this loop to the per-arc readout.  Row e of
 * the readout is [F[i0(e)] | F[i1(e)] | arc_labels[e]] with F = [state | nodes (iff state_dim > 0)], (i0, i1) the e-th index
 * pair of the transposed, reordered Adjacency (i0 = entry_dst[e], the CSR row of entry e; i1 = adj_src[e]) and arc_labels in
 * ORIGINAL arc order: the reference pairs the two by position (consistent for symmetric, lexicographically sorted arc
 * lists; SURVEY.md 8a quirk 6, reproduced as is).  arc_mask = set_mask & output_mask over arcs.  net_output must take
 * 2 (NL [D>0] + Ds) + AL inputs (GNN/MLP.py:109).  Single GPU only. */
int gnn_loop_set_edge_readout(gnn_loop *l, const int32_t *entry_dst, const float *arc_labels, const uint8_t *arc_mask);
/* One training step without the optimizer (reference GNN/GNN_BaseClass.py:231-247: GradientTape around
 * evaluate_single_graph(training=True), GNN/GNN.py:180-199): training-mode forward through the unrolled loop (Dropout masks,
 * BatchNormalization batch statistics), loss = sum_i w_i L(t_i, out_i), back-propagation through every executed body.
 *   src_*            Adjacency in CSR form BY SOURCE (rows = source node, inner = destination ascending): the transposed
 *                    aggregation of the backward pass; all NULL = derived from the graph's own CSR on first use and kept
 *   targets, sample_weights, n_targets   rows = masked nodes (node-based) or graphs (graph-based); loss_kind 0 =
 *                    categorical_crossentropy(from_logits=False), 1 = mean_squared_error, 2 = categorical_crossentropy(from_logits=True)
 *   n_graphs, ng_*   NodeGraph^T in CSR form (as gnn_loop_readout) for GNNgraphBased, n_graphs = 0 otherwise
 *   dropout_state / dropout_output   [n_layers + 1] Dropout rate in front of Dense l (0 = none; last entry: in front of
 *                    BatchNormalization), i.e. GNN/MLP.py:54-55 after its position shift; a NEGATIVE value -r is an AlphaDropout
 *                    of rate r (MLP(..., alphadropout=True), GNN/MLP.py:59-61)
 *   masks_*          injected keep-masks (uint8, 1 = keep): for net_state max_iter blocks, each the concatenation over the
 *                    dropout positions of [N, width]; for net_output one such block over the masked rows; NULL = engine RNG(seed)
 *   bn_state / bn_output   [gamma | beta] of the trailing BatchNormalization (NULL without one)
 * Outputs: *loss_out, *k_out (executed bodies), grads_* flat in get_weights() order of the TRAINABLE arrays
 * [dW1, db1, ..., dgamma, dbeta] (raw sums over the iterations: the division by k of :241 is the caller's), bn_batch_state
 * [k][2][Ds] and bn_batch_output [2][T] = batch mean / biased batch variance of every BatchNormalization call (the caller
 * applies the moving-average updates).  Single GPU, node/graph-based. */
int gnn_loop_train_step(gnn_loop *l, const int32_t *src_indptr, const int32_t *src_dst, const float *src_w,
                        const float *targets, const float *sample_weights, int64_t n_targets, int loss_kind,
                        int n_graphs, const int32_t *ng_indptr, const int32_t *ng_node, const float *ng_w,
                        const float *dropout_state, const float *dropout_output, const uint8_t *masks_state,
                        const uint8_t *masks_output, uint64_t seed, const float *bn_state, const float *bn_output,
                        float *loss_out, float *k_out, float *grads_state, float *grads_output,
                        float *bn_batch_state, float *bn_batch_output);
/* The two halves of gnn_loop_train_step, for models whose loss spans several loops (LGNN 'parallel' / 'residual'
 * training, reference GNN/LGNN.py:201-224 inside GNN_BaseClass.py:231-247, where layer i + 1's labels contain layer
 * i's state / output, LGNN.py:227-260).
 *   gnn_loop_train_forward   training-mode Loop; out_nodes [n_masked, T] (may be NULL) are the node-level outputs.  The
 *                    training-mode state / outputs become the loop's result (gnn_loop_get_state / get_output / readout,
 *                    gnn_graph_update_labels), and the context of the backward pass stays with the loop.
 *                    On node-range shards with full-replica numbering (one process per rank, an RCCL communicator; round 3) this
 *                    half also runs sharded: the state rows are all-gathered after every body, and the BatchNormalization batch
 *                    statistics (reference GNN/MLP.py:62-63, training=True) and the gate of reduce_any (GNN.py:218) are those of
 *                    the rows of ALL ranks - 3 F floats per rank and BatchNormalization call.  Give src_indptr / src_dst / src_w = the
 *                    arcs that LEAVE the owned rows (CSR over the owned rows, destinations as replica rows) for the backward half.
 *   gnn_loop_train_backward  d_out_nodes [n_masked, T] = d loss / d out_nodes; d_state_extra [N, Ds] (or NULL) = an extra
 *                    gradient on the final state; d_nodes [N, NL] (or NULL) receives d loss / d node labels; d_arc_labels
 *                    [n_arcs, AL] (or NULL; edge-based loops after gnn_graph_set_arc_order) d loss / d arc labels in
 *                    ORIGINAL arc order (LGNN.py:253-254).  One backward per forward.  On shards (after a sharded forward that was
 *                    given the by-source adjacency; node- / graph-based, no d_state_extra / d_nodes / d_arc_labels): per body the
 *                    gradient of the aggregated-state columns is all-gathered and every rank sums what its out-arcs carry back, the
 *                    sums of BatchNormalization's backward pass are those of all ranks, and the ranks' shares of the weight
 *                    gradients are added in rank order - every rank returns the same, complete gradients.  gnn_loop_train_step (one
 *                    call with the loss inside) stays single-GPU.
 *   gnn_loss_grad    host helper: *loss = sum_i w_i L(t_i, out_i) and d_out = d loss / d out (may be NULL). */
int gnn_loop_train_forward(gnn_loop *l, const int32_t *src_indptr, const int32_t *src_dst, const float *src_w,
                           const float *dropout_state, const float *dropout_output, const uint8_t *masks_state,
                           const uint8_t *masks_output, uint64_t seed, const float *bn_state, const float *bn_output,
                           float *k_out, float *out_nodes);
int gnn_loop_train_backward(gnn_loop *l, const float *d_out_nodes, const float *d_state_extra, float *grads_state,
                            float *grads_output, float *bn_batch_state, float *bn_batch_output, float *d_nodes,
                            float *d_arc_labels);
/* Optimizer step on the device (reference GNN_BaseClass.py:243-247, optimizer.apply_gradients on the trainable variables of
 * both nets, and the moving statistics Keras BatchNormalization updates in training mode): weights, optimizer slots (kept with
 * the gnn_mlp) and gradients stay in HBM.  kind 0 = SGD, hyper = {learning_rate, momentum, -, -}; kind 1 = Adam, hyper =
 * {lr_t = lr sqrt(1 - beta_2^t) / (1 - beta_1^t), beta_1, beta_2, epsilon} (the caller counts t).
 *   gnn_loop_arm_optimizer   one-shot: the NEXT gnn_loop_train_step applies the update behind its backward pass, before the
 *                            step's single wait for the device; mean != 0 divides the net_state gradients by the iteration
 *                            count (GNN_BaseClass.py:241).  The gradients returned by that step are the raw ones.
 *   gnn_loop_optimizer_step  the same as a call of its own, after gnn_loop_train_step or forward + backward (once per backward);
 *                            state_grad_scale multiplies the net_state gradients.
 * With bn_state / bn_output NULL in the train calls, gamma / beta are the gnn_mlp's own device copy (the one updated here).
 * gnn_mlp_get_weights (above) reads the current arrays back. */
/* Keras BatchNormalization in training mode also updates its moving statistics when no gradient is taken (a Loop(training=True)
 * call, reference GNN/GNN.py:251-280 with training=True): moving <- moving * momentum + batch * (1 - momentum) per call, from the
 * batch statistics of the last gnn_loop_train_forward, on the device.  Once per forward pass; not after an optimizer step. */
int gnn_loop_update_moving_statistics(gnn_loop *l, float bn_momentum_state, float bn_momentum_output);
int gnn_loop_arm_optimizer(gnn_loop *l, int kind, const float *hyper, int mean, float bn_momentum_state, float bn_momentum_output);
int gnn_loop_optimizer_step(gnn_loop *l, int kind, const float *hyper, float state_grad_scale, float bn_momentum_state,
                            float bn_momentum_output);
int gnn_loss_grad(int loss_kind, int64_t n_rows, int n_out, const float *targets, const float *out,
                  const float *sample_weights, double *loss, float *d_out);
/* selects the implementation:
 *   0 = unfused reference kernels (one kernel per TF op), bit-identical to oracle/gnn_oracle.c;
 *   1 = fused gather + MLP kernel with the dense layers on the f32 MFMA: the same k-ordered fmaf chains, bit-identical
 *       to the oracle;
 *   2 = (default) fused kernel with the dense layers on the bf16 MFMA: every fp32 operand is cut into three exact bf16
 *       pieces and the six leading piece products are accumulated in fp32 (error per product <= 3 * 2^-24, i.e. fp32
 *       rounding level), activations through v_exp_f32 / v_rcp_f32; results within fp32 rounding noise of 0 / 1
 *       (tests: 1e-5 against the float64 oracle, the tolerance of BASELINE.json), not bit for bit.
 * 1 and 2 fall back to 0 when the shapes are not covered.  *used (may be NULL) reports the choice.
 * k contract of impl 2 (reference GNN/GNN.py:202-220: `distance > threshold * norm`, reduce_any, k < max_iteration): its iteration count
 * is the bit-exact chain's.  Every body also records whether some node moved by a margin ("robust") and whether some node's test lay
 * within a guard band of the threshold ("borderline"; band = 1e-5 norm + 1e-3 threshold norm); a gate with a robust mover, or without
 * a borderline node, is decided identically by both arithmetics.  When a gate of a run is neither, gnn_loop_run repeats that Loop on
 * impl 1 before it returns, and k, state and output are THAT run's (bit-identical to the oracle).  threshold 0 with moving states never
 * triggers it. */
int gnn_loop_set_impl(gnn_loop *l, int impl, int *used);
/* Outcome of the certified gate: *last_run_repeated = 1 when the last gnn_loop_run / _run_group / _run_many of this loop was repeated on
 * impl 1 (its results are the exact path's), *repeats_total = how often that has happened on this handle.  Either pointer may be NULL. */
int gnn_loop_gate_info(const gnn_loop *l, int *last_run_repeated, int *repeats_total);
/* Small graphs (every 32-node tile resident at once: <= 8,192 owned nodes, single GPU) with a net_state no wider than 32 run
 * the whole tf.while_loop of GNN/GNN.py:271 - initial state, first condition, every body with a grid barrier in between - in
 * ONE persistent launch when impl is 1 or 2 (exact f32-MFMA arithmetic in both cases, bit-identical to the oracle).  enable = 0
 * keeps such a loop to one launch per body; *used (may be NULL) tells whether the persistent launch will be taken. */
int gnn_loop_set_persistent(gnn_loop *l, int enable, int *used);
/* Which form of the fused iteration kernel runs the bodies of GNN/GNN.py:223-242 on the default path (impl 2) when both cover the
 * net (state width 64, two or three Dense layers, 128-wide hidden layers, concat width 129 .. 144, no feature-sliced exchange):
 *   form 1  one wave owns a 32-node tile from gather to store (k_fused);
 *   form 2  a wave PAIR shares the tile, each wave gathering 16 of its nodes and producing half of every layer's output features
 *           (k_fused_pair);
 *   form 0  the library's choice (default).
 * The two forms evaluate the same arithmetic per node: states, outputs and k are identical bit for bit.  *used (may be NULL) = the form
 * the next run will take (1 or 2; 0 when the fused path does not cover the loop at all). */
int gnn_loop_set_tile_form(gnn_loop *l, int form, int *used);
/* per-kernel HIP-event timing of the last gnn_loop_run when profiling was enabled:
 * avg_iter_ms = mean duration of the per-iteration kernel(s), total_ms = whole loop on the stream. */
int gnn_loop_set_profiling(gnn_loop *l, int enable);
/* The label aggregates ArcNode^T . arc labels and Adjacency^T . node labels (GNN/GNN.py:259, :263) do not depend on the state:
 * the fused path builds them on the first run and keeps them until the labels change (gnn_graph_update_labels).  This call
 * drops them, so that the next gnn_loop_run pays for them like every Loop() of the reference does (bench.py: cold figure). */
int gnn_loop_drop_cached_aggregates(gnn_loop *l);
int gnn_loop_get_timing(const gnn_loop *l, float *total_ms, float *avg_iter_ms, int *n_iter_timed);
/* With profiling on: mean time on the loop's stream between the end of one body's kernel(s) and the start of the next body's, over the last
 * run - on shards that is the per-iteration exchange (the reads of `state` at reference GNN/GNN.py:234 and the reduce_any of :218 that span
 * all nodes: all-gather of state rows / boundary rows + flag block, or the sliced layout's pack, all-to-all, aggregation, all-to-all, unpack);
 * a few microseconds of launch gap on one GPU.  0 unless gnn_loop_set_profiling was on. */
int gnn_loop_get_exchange_timing(const gnn_loop *l, float *avg_between_bodies_ms);
/* Work counters of one iteration of this loop on this rank and the timing of its last run (SURVEY.md 8b "gnn_counters_get"):
 * algorithmic bytes per iteration = E (4 Ds + 8) + 4 (n_rows + 1) + n_rows (8 Ds + 4 (2 NL + AL)) over the OWNED rows (SURVEY.md
 * 8d: fp32 values, int32 indices, no cache credit, fused iteration), FLOPs per iteration = n_rows 2 sum_l in_l out_l + 2 E Ds;
 * iterations / total_ms / avg_iteration_ms as gnn_loop_get_timing (the times are 0 unless gnn_loop_set_profiling was on).
 * Any output pointer may be NULL. */
int gnn_counters_get(const gnn_loop *l, double *bytes_per_iteration, double *flops_per_iteration, int *iterations, float *total_ms,
                     float *avg_iteration_ms);
/* LGNN.Loop (reference GNN/LGNN.py:263-290) in one call: loops[i] was created on graphs[i]; graphs[0] is the ORIGINAL graph `base`,
 * graphs[i > 0] derived from it (gnn_graph_derive / _derive_edge; the same derived graph may serve several layers).  Runs layer 0,
 * relabels graphs[1] from `base` with layer 0's state / output (gnn_graph_update_labels: LGNN.py:227-260, :287), runs layer 1, ...
 * k_out[n_layers] receives the iteration count of every layer (K of the reference); state / outputs of every layer stay readable
 * through gnn_loop_get_state / gnn_loop_get_output (or gnn_loop_readout) of its loop.  Single GPU or one rank of an RCCL job. */
int gnn_lgnn_run(gnn_loop *const *loops, gnn_graph *const *graphs, int n_layers, int get_state, int get_output, float *k_out);
/* A loop created on a communicator may be destroyed before or after it: gnn_comm_destroy with loops alive only marks the
 * communicator closed, the last gnn_loop_destroy releases it. */
int gnn_loop_destroy(gnn_loop *l);

/* ---- multi-GPU (one process per GPU, RCCL over xGMI) -------------------------------------------------------------
 * No reference counterpart: the reference is single-device.  What is sharded is the row-wise work of convergence()
 * (GNN/GNN.py:223-242: sparse_dense_matmul(adjacency, state) at :234 needs neighbour rows of other ranks) and the global
 * reduce_any of condition() (:218).  Nodes are sharded by contiguous ranges; every iteration ends with one grouped RCCL
 * all-gather of the owned state rows (or, for shards created by gnn_graph_create_halo, of the owned BOUNDARY rows only)
 * and the owned convergence flag.  The 128-byte id is produced on rank 0 and handed to the other ranks by the caller
 * (file, socket, torch.distributed store ...). */
int gnn_shard_range(int64_t n_nodes, int rank, int world, int64_t *row_begin, int64_t *n_rows);   /* owned rows of a rank */
int gnn_comm_unique_id(uint8_t id[128]);
int gnn_comm_create(const uint8_t id[128], int rank, int world, int device, gnn_comm **out);
int gnn_comm_allreduce_max(gnn_comm *c, double *value);   /* barrier + max over ranks (bench timing) */
int gnn_comm_destroy(gnn_comm *c);
/* Boundary ("halo") exchange.  gnn_halo_plan (host only, needs the CSR-by-destination of the WHOLE graph): slot[v] =
 * position of node v among the boundary rows of its owner (rows read by another rank), -1 for interior nodes; counts[q] =
 * boundary rows of rank q; *block = max count.  gnn_graph_create_halo builds rank r's shard in the compact index space
 *   [0, shard) owned rows | shard + q * block + slot: boundary rows of rank q
 * (adj_src_replica and the rows of nodes_replica are given in that space; send_rows = ascending owned-row indices of this
 * rank's boundary rows).  Loops on such a graph all-gather `block` rows per rank and iteration instead of whole shards.
 * LGNN relabelling and the edge-based readout are not available on halo shards. */
int gnn_halo_plan(int64_t n_nodes, int world, const int32_t *indptr, const int32_t *adj_src, int32_t *slot /* [n_nodes] */,
                  int64_t *counts /* [world] */, int64_t *block);
int gnn_graph_create_halo(int64_t n_nodes_global, int rank, int world, int64_t halo_block, int64_t n_send, const int32_t *send_rows,
                          int64_t n_arcs, const int32_t *indptr, const int32_t *adj_src_replica, const float *adj_w,
                          const float *arc_w, const float *arc_labels, int dim_arc_label, const float *nodes_replica,
                          int dim_node_label, const uint8_t *mask, int device, gnn_graph **out);
/* Feature-sliced exchange for node-range shards with full-replica numbering (an alternative to exchanging state ROWS).  The
 * aggregation tf.sparse.sparse_dense_matmul(adjacency, state) (reference GNN/GNN.py:234) is independent per state column, so rank q
 * aggregates columns [q Ds / P, (q + 1) Ds / P) for ALL nodes over the whole graph's adjacency, and two all-to-all steps per
 * iteration move column slices of the owned rows in and aggregated slices back: 2 (P - 1) / P^2 of N Ds floats received per rank
 * and iteration instead of (P - 1) / P (56 MB instead of 224 MB at N = 1 M, Ds = 64, P = 8).  Bit-identical results.
 *   gnn_graph_set_full_adjacency  the whole graph's CSR by destination (global ids) beside the shard's own rows
 *   gnn_loop_set_slice_exchange   on != 0: use it (Ds must be a multiple of the world size); every rank of the job alike.
 *                                 on == 1: the slice is aggregated in P row blocks (one per destination rank, in the order rank + 1, ...,
 *                                 rank) and block t is sent on a second stream while block t + 1 is aggregated - the return all-to-all is
 *                                 hidden behind the aggregation except for the rank's own block; on == 2: the whole slice, then one grouped
 *                                 all-to-all (kept for comparison) */
int gnn_graph_set_full_adjacency(gnn_graph *g, int64_t n_global, const int32_t *indptr, const int32_t *adj_src, const float *adj_w);
int gnn_loop_set_slice_exchange(gnn_loop *l, int on);
/* In-process LOOPBACK group: `world` communicators on ONE device sharing one stream; the exchange steps become
 * device-to-device copies between the members' buffers.  It runs the sharded code path (row offsets, padded replicas,
 * per-rank flag slots, every exchange call site) on a single GPU; used by the parity tests, never for speed.  One loop per
 * rank and group; the ranks of a group are driven together by the *_group entry points (k, readout: rank 0's values, after
 * checking that all ranks agree). */
int gnn_comm_create_loopback(int world, int device, gnn_comm **out /* [world] */);
int gnn_loop_run_group(gnn_loop **loops, int n, float *k_out);
int gnn_loop_readout_group(gnn_loop **loops, int n, int n_graphs, const int32_t *ng_indptr, const int32_t *ng_node,
                           const float *ng_w, float *out_graph);
int gnn_graph_update_labels_group(gnn_graph **dsts, gnn_graph *const *bases, gnn_loop *const *froms, int n, int get_state,
                                  int get_output);

#ifdef __cplusplus
}
#endif
#endif /* GNN_HIP_H */
