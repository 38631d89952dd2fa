"""CPU oracle (NumPy) for the GNN.Loop state-propagation hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` may be imported, linked or
executed by the product (``gnn_tf_2.x_amd/``).  Only ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` use it,
and there only as the checker.

What this file is
-----------------
A plain NumPy restatement of the reference algorithm, every function citing the
reference ``file:line`` it follows (paths relative to the reference root).  Every
routine takes ``dtype``: ``np.float32`` reproduces the reference's arithmetic type,
``np.float64`` is the shadow used as arbiter for rounding questions.

Pinning status (see DESIGN.md "Oracle")
---------------------------------------
* Graph half (ArcNode / Adjacency / NodeGraph / merge / transposition): PINNED.
  Checked against fixtures emitted by the reference's own ``GNN.graph_class`` /
  ``GNN.GNN_utils`` run in the build container (``tests/golden/make_fixtures.py``).
* TensorFlow half (sparse_dense_matmul, Dense, activations, BatchNormalization,
  while_loop, boolean_mask, scatter_nd): **parity unpinned**.  TensorFlow (unpinned
  in ``requirements.txt:1``) is not installable here and the reference ships no tests
  or golden outputs, so these follow the published Keras/TF semantics:
    - Dense: ``act(x @ W + b)``, W ``[in, out]``;
    - selu: ``scale * (x if x > 0 else alpha * (exp(x) - 1))`` with
      scale 1.0507009873554805, alpha 1.6732632423543772;
    - elu (alpha 1), relu, tanh, sigmoid, linear, softmax over the last axis;
    - BatchNormalization inference: ``x * inv + (beta - mean * inv)`` with
      ``inv = gamma / sqrt(var + eps)`` written as ``(1 / sqrt(var + eps)) * gamma``,
      defaults eps 1e-3, gamma 1, beta 0, mean 0, var 1;
    - Dropout / AlphaDropout: identity when ``training=False``.
"""
from __future__ import annotations

import numpy as np

SELU_SCALE = 1.0507009873554805
SELU_ALPHA = 1.6732632423543772
BN_EPS = 1e-3

ACT_CODES = {'linear': 0, None: 0, 'relu': 1, 'selu': 2, 'elu': 3, 'tanh': 4, 'sigmoid': 5, 'softmax': 6}


# ---------------------------------------------------------------------------------------------------------------------
# graph matrices
# ---------------------------------------------------------------------------------------------------------------------
def arcnode_values(arcs: np.ndarray, aggregation_mode: str) -> np.ndarray:
    """Per-arc aggregation weight (the data of ArcNode, also reused for Adjacency).

    Follows GNN/graph_class.py:98-121 (buildArcNode) and :90-95 (buildAdiacency re-uses ArcNode.data):
    'sum' -> 1; 'normalized' -> 1/len(arcs) (the code divides by the number of ARCS, :112-113);
    'average' -> 1/in-degree of the arc's destination (:116-118).  Returned as float32 (floatx, :40).
    """
    dst = np.asarray(arcs)[:, 1]
    w = np.ones(len(dst), dtype=np.float64)
    if aggregation_mode == 'normalized':
        w = w * float(1 / len(dst))
    elif aggregation_mode == 'average':
        _, inv, counts = np.unique(dst, return_inverse=True, return_counts=True)
        w = w / counts[inv]
    elif aggregation_mode != 'sum':
        raise ValueError('ERROR: Unknown aggregation mode')
    return w.astype(np.float32)


def transposed_csr(row: np.ndarray, col: np.ndarray, val: np.ndarray, n_rows_T: int):
    """CSR of the TRANSPOSE of a COO matrix, entries in row-major (row, col) order.

    Follows GNN/graph_class.py:365-372 (COO2SparseTransposedTensor): indices become (col, row) and
    tf.sparse.reorder sorts them row-major.  Returns (indptr int64[n_rows_T+1], inner int64[nnz], val[nnz]).
    A stable lexicographic sort keeps duplicates in input order.
    """
    r_t = np.asarray(col, dtype=np.int64)
    c_t = np.asarray(row, dtype=np.int64)
    order = np.lexsort((c_t, r_t))  # primary r_t, secondary c_t; lexsort is stable
    counts = np.bincount(r_t, minlength=n_rows_T)
    indptr = np.zeros(n_rows_T + 1, dtype=np.int64)
    np.cumsum(counts, out=indptr[1:])
    return indptr, c_t[order], np.asarray(val)[order]


def graph_matrices(arcs: np.ndarray, n_nodes: int, aggregation_mode: str):
    """(AdjT csr, ArcNodeT csr) exactly as GraphTensor.fromGraphObject hands them to Loop.

    Adjacency[src, dst] = w_a (graph_class.py:93-95) transposed -> rows = dst, inner = src ascending.
    ArcNode[a, dst] = w_a (graph_class.py:121) transposed -> rows = dst, inner = arc id ascending.
    """
    arcs = np.asarray(arcs)
    src = arcs[:, 0].astype(np.int64)
    dst = arcs[:, 1].astype(np.int64)
    w = arcnode_values(arcs, aggregation_mode)
    adjT = transposed_csr(src, dst, w, n_nodes)
    arcT = transposed_csr(np.arange(len(dst), dtype=np.int64), dst, w, n_nodes)
    return adjT, arcT


def nodegraph_single(n_nodes: int) -> np.ndarray:
    """GNN/graph_class.py:132-144 (buildNodeGraph, problem_based == 'g'): [n, 1] filled with 1/n, float32."""
    return np.ones((n_nodes, 1), dtype=np.float32) * 1 / n_nodes


def merge_graphs(graphs: list[dict], problem_based: str):
    """Batch several graphs into one block-diagonal graph.  Follows GNN/graph_class.py:285-319.

    ``graphs``: dicts with keys arcs, nodes, targets, set_mask, output_mask, sample_weights, NodeGraph.
    Node ids of graph i are shifted by the node count of graphs < i (:304); NodeGraph is block_diag (:314-315).
    """
    nodes_lens = [g['nodes'].shape[0] for g in graphs]
    arcs = []
    for i, g in enumerate(graphs):
        a = np.array(g['arcs'], copy=True)
        a[:, :2] += sum(nodes_lens[:i])
        arcs.append(a)
    out = dict(arcs=np.concatenate(arcs, axis=0),
               nodes=np.concatenate([g['nodes'] for g in graphs], axis=0),
               targets=np.concatenate([g['targets'] for g in graphs], axis=0),
               set_mask=np.concatenate([g['set_mask'] for g in graphs], axis=0),
               output_mask=np.concatenate([g['output_mask'] for g in graphs], axis=0),
               sample_weights=np.concatenate([g['sample_weights'] for g in graphs], axis=0),
               NodeGraph=None)
    if problem_based == 'g':
        tot_r = sum(g['NodeGraph'].shape[0] for g in graphs)
        tot_c = sum(g['NodeGraph'].shape[1] for g in graphs)
        ng = np.zeros((tot_r, tot_c), dtype=graphs[0]['NodeGraph'].dtype)
        r = c = 0
        for g in graphs:
            m = g['NodeGraph']
            ng[r:r + m.shape[0], c:c + m.shape[1]] = m
            r += m.shape[0]
            c += m.shape[1]
        out['NodeGraph'] = ng
    return out


# ---------------------------------------------------------------------------------------------------------------------
# MLP
# ---------------------------------------------------------------------------------------------------------------------
def get_inout_dims(net_name, dim_node_label, dim_arc_label, dim_target, problem_based, dim_state, hidden_units,
                   *, layer=0, get_state=False, get_output=False):
    """MLP widths.  Follows GNN/MLP.py:68-122 (incl. the LGNN layer>0 relabelling formulas :93-100)."""
    assert layer >= 0 and problem_based in ['a', 'n', 'g'] and dim_state >= 0
    ds, nl, al, t = dim_state, dim_node_label, dim_arc_label, dim_target
    arc_based = problem_based == 'a'
    if layer > 0:
        gs, go = int(get_state), int(get_output)
        if ds != 0:
            nl = nl + ds * gs + t * (not arc_based) * go
        else:
            nl = nl + layer * nl * gs + ((layer - 1) * gs + 1) * t * (not arc_based) * go
        al = al + t * arc_based * go
    if net_name == 'state':
        n_in, n_out = al + 2 * (nl + ds), (ds if ds else nl)
    elif net_name == 'output':
        n_in, n_out = arc_based * (nl + al + ds) + nl + dim_state, t
    else:
        raise ValueError(":param net_name: not in ['state', 'output']")
    if hidden_units is None or (type(hidden_units) == int and hidden_units <= 0):
        hidden_units = []
    layers = (hidden_units + [n_out]) if type(hidden_units) == list else [hidden_units, n_out]
    return n_in, layers


def activation(x: np.ndarray, name) -> np.ndarray:
    """Keras activation semantics (not in the reference repo; see module docstring)."""
    dt = x.dtype
    if name in ('linear', None):
        return x
    if name == 'relu':
        return np.maximum(x, dt.type(0))
    if name == 'selu':
        neg = dt.type(SELU_ALPHA) * (np.exp(np.minimum(x, dt.type(0))) - dt.type(1))
        return dt.type(SELU_SCALE) * np.where(x > 0, x, neg)
    if name == 'elu':
        return np.where(x > 0, x, np.exp(np.minimum(x, dt.type(0))) - dt.type(1))
    if name == 'tanh':
        return np.tanh(x)
    if name == 'sigmoid':
        return dt.type(1) / (dt.type(1) + np.exp(-x))
    if name == 'softmax':
        e = np.exp(x - x.max(axis=-1, keepdims=True))
        return e / e.sum(axis=-1, keepdims=True)
    raise ValueError(f'unknown activation {name!r}')


def mlp_forward(x: np.ndarray, weights: list[np.ndarray], activations: list, batch_normalization: bool,
                dtype=np.float32) -> np.ndarray:
    """Inference forward of the Sequential built by GNN/MLP.py:11-64.

    ``weights`` is the Keras ``get_weights()`` list ``[W1, b1, ..., Wn, bn, (gamma, beta, mean, var)]``
    (GNN/GNN.py:163-165).  Layer order: [Dropout -> identity]* Dense(act) ... then one trailing
    BatchNormalization when enabled (MLP.py:62-63, default True :13).
    """
    n_dense = len(activations)
    if x.shape[0] > 16384:      # rows are independent: evaluate in cache-sized row blocks (same values, a fraction of the time)
        return np.concatenate([mlp_forward(x[i:i + 8192], weights, activations, batch_normalization, dtype) for i in range(0, x.shape[0], 8192)])
    h = np.asarray(x, dtype=dtype)
    for l in range(n_dense):
        w = np.asarray(weights[2 * l], dtype=dtype)
        b = np.asarray(weights[2 * l + 1], dtype=dtype)
        h = activation(h @ w + b, activations[l])
    if batch_normalization:
        gamma, beta, mean, var = (np.asarray(a, dtype=dtype) for a in weights[2 * n_dense:2 * n_dense + 4])
        inv = (dtype(1) / np.sqrt(var + dtype(BN_EPS))) * gamma
        h = h * inv + (beta - mean * inv)
    return h


# ---------------------------------------------------------------------------------------------------------------------
# Loop
# ---------------------------------------------------------------------------------------------------------------------
def spmm_csr(csr, dense: np.ndarray, dtype=np.float32) -> np.ndarray:
    """tf.sparse.sparse_dense_matmul(sparse[n_rows x n_inner], dense) for a row-major sparse operand
    (GNN/GNN.py:234, :259, :263): out[r] = sum over the row's entries, in stored order, of val * dense[inner]."""
    indptr, inner, val = csr
    n_rows = len(indptr) - 1
    dense = np.asarray(dense, dtype=dtype)
    out = np.zeros((n_rows, dense.shape[1]), dtype=dtype)
    if len(inner) == 0 or dense.shape[1] == 0:
        return out
    val = np.asarray(val, dtype=dtype)
    if np.dtype(dtype) == np.float64 and len(inner) > 100_000:
        # float64 shadow on large graphs: SciPy's CSR product walks every row's entries in stored order as well; at float64 the
        # remaining freedom (fused or unfused multiply-add) is 1e-16, far below anything the shadow is compared at
        import scipy.sparse as sp
        return np.asarray(sp.csr_matrix((val, np.asarray(inner), np.asarray(indptr)), shape=(n_rows, dense.shape[0])) @ dense)
    deg = np.diff(indptr)
    # accumulate the j-th entry of every row in one vectorised step: same per-row order as a sequential CSR walk
    for j in range(int(deg.max())):
        rows = np.nonzero(deg > j)[0]
        e = indptr[rows] + j
        out[rows] = out[rows] + val[e, None] * dense[inner[e]]
    return out


def not_converged(state: np.ndarray, state_old: np.ndarray, threshold: float) -> np.ndarray:
    """Per-node boolean of GNN/GNN.py:206-215: sqrt(sum (s - s_old)^2) > threshold * sqrt(sum s_old^2).  Strict '>'."""
    dt = state.dtype
    dist = np.sqrt(np.sum(np.square(state - state_old), axis=1))
    norm = np.sqrt(np.sum(np.square(state_old), axis=1))
    return dist > dt.type(threshold) * norm


def loop_node(g: dict, net_state: dict, net_output: dict, state_vect_dim: int, max_iteration: int, threshold: float,
              state0: np.ndarray | None = None, dtype=np.float32, return_trace: bool = False):
    """GNNnodeBased.Loop, inference.  Follows GNN/GNN.py:251-280 with condition :202-220 and convergence :223-242.

    ``g``: dict with nodes [N,NL], arcs [E,2+AL], set_mask, output_mask (bool [N]), adjT, arcT (csr triples).
    ``net_*``: dict(weights=[...], activations=[...], batch_normalization=bool).
    ``state0``: injected initial state when state_vect_dim > 0 (the reference draws tf.random.normal, :262).
    Returns (k as float, state [N,Ds], out [M,T]) (+ list of per-iteration states when return_trace).
    """
    nodes = np.asarray(g['nodes'], dtype=dtype)
    arc_labels = np.asarray(g['arcs'], dtype=dtype)[:, 2:]
    n = nodes.shape[0]
    aggregated_arcs = spmm_csr(g['arcT'], arc_labels, dtype)                      # GNN.py:259
    aggregated_nodes = np.zeros((n, 0), dtype=dtype)                              # :260
    if state_vect_dim > 0:
        if state0 is None:
            raise ValueError('oracle needs an injected state0 when state_vect_dim > 0')
        state = np.asarray(state0, dtype=dtype)
        aggregated_nodes = spmm_csr(g['adjT'], nodes, dtype)                      # :263
    else:
        state = nodes.copy()                                                      # :265
    state_old = np.ones_like(state)                                               # :266
    k = 0
    trace = []
    # tf.while_loop(condition, convergence): condition is evaluated BEFORE each body (:271)
    while bool(np.any(not_converged(state, state_old, threshold))) and k < max_iteration:
        node_components = state if not state_vect_dim else np.concatenate([state, nodes], axis=1)   # :228-230
        aggregated_states = spmm_csr(g['adjT'], state, dtype)                     # :234
        inp_state = np.concatenate([node_components, aggregated_states, aggregated_nodes, aggregated_arcs], axis=1)  # :237
        state_new = mlp_forward(inp_state, net_state['weights'], net_state['activations'],
                                net_state['batch_normalization'], dtype)          # :240
        k, state, state_old = k + 1, state_new, state                             # :242
        if return_trace:
            trace.append(state)
    mask = np.logical_and(g['set_mask'], g['output_mask'])                        # :275
    feats = state if not state_vect_dim else np.concatenate([state, nodes], axis=1)   # :247
    out = mlp_forward(feats[mask], net_output['weights'], net_output['activations'],
                      net_output['batch_normalization'], dtype)                   # :248, :279
    if return_trace:
        return float(k), state, out, trace
    return float(k), state, out


def edge_features(g: dict, state: np.ndarray, state_vect_dim: int, dtype=np.float32) -> np.ndarray:
    """GNNedgeBased.apply_filters.  Follows GNN/GNN.py:289-302: rows of [state | nodes?] gathered by the index pairs of the
    transposed, reordered Adjacency (:294-295), concatenated with arcs[:, 2:] in arc order (:299), masked by the arc mask."""
    nodes = np.asarray(g['nodes'], dtype=dtype)
    feats = state if not state_vect_dim else np.concatenate([state, nodes], axis=1)      # :291
    indptr, src, _ = g['adjT']
    dst = np.repeat(np.arange(len(indptr) - 1), np.diff(indptr))
    pair = np.concatenate([feats[dst], feats[src]], axis=1)                             # :294-295 ([E, 2, F] -> [E, 2F])
    arc_state = np.concatenate([pair, np.asarray(g['arcs'], dtype=dtype)[:, 2:]], axis=1)   # :299
    return arc_state[np.logical_and(g['set_mask'], g['output_mask'])]                   # :302


def loop_graph(g: dict, net_state, net_output, state_vect_dim, max_iteration, threshold, state0=None, dtype=np.float32):
    """GNNgraphBased.Loop.  Follows GNN/GNN.py:318-333: node-based Loop, then NodeGraph^T @ out_nodes (:331-332)."""
    if g.get('NodeGraph') is None:
        raise ValueError('WRONG GNN. NodeGraph is None: GNN is graph-based, while problem is non graph-based.')
    k, state, out_nodes = loop_node(g, net_state, net_output, state_vect_dim, max_iteration, threshold, state0, dtype)
    nodegraph = np.asarray(g['NodeGraph'], dtype=dtype)
    return k, state, nodegraph.T @ out_nodes


def update_graph(g: dict, state: np.ndarray, output: np.ndarray, get_state: bool, get_output: bool, dtype=np.float32):
    """LGNN.update_graph for node/graph-based layers.  Follows GNN/LGNN.py:227-260.

    nodes <- [g.nodes | state (if get_state) | scatter_nd(where(mask), output) (if get_output)] (:241-259);
    always derived from the ORIGINAL graph g (LGNN.py:287)."""
    new = dict(g)
    nodes = np.asarray(g['nodes'], dtype=dtype)
    extra = []
    if get_state:
        extra.append(np.asarray(state, dtype=dtype))
    if get_output:
        mask = np.logical_and(g['set_mask'], g['output_mask'])
        scat = np.zeros((len(mask), output.shape[1]), dtype=dtype)
        scat[np.nonzero(mask)[0]] = output
        extra.append(scat)
    new['nodes'] = np.concatenate([nodes] + extra, axis=1)
    return new


def loop_edge(g: dict, net_state, net_output, state_vect_dim, max_iteration, threshold, state0=None, dtype=np.float32):
    """GNNedgeBased.Loop.  Follows GNN/GNN.py:286-302 on top of :251-280: the node-based state loop, then net_output on the
    per-arc rows of edge_features (set_mask / output_mask are over the ARCS)."""
    n = np.asarray(g['nodes']).shape[0]
    node_side = dict(g, set_mask=np.ones(n, bool), output_mask=np.ones(n, bool))
    ds = state_vect_dim if state_vect_dim else np.asarray(g['nodes']).shape[1]
    wn = ds + (np.asarray(g['nodes']).shape[1] if state_vect_dim else 0)
    dummy = dict(weights=[np.zeros((wn, 1), dtype), np.zeros(1, dtype)], activations=['linear'], batch_normalization=False)
    k, state, _ = loop_node(node_side, net_state, dummy, state_vect_dim, max_iteration, threshold, state0, dtype)
    feats = edge_features(g, state, state_vect_dim, dtype)
    out = mlp_forward(feats, net_output['weights'], net_output['activations'], net_output['batch_normalization'], dtype)
    return k, state, out


def update_graph_edge(g: dict, state: np.ndarray, output: np.ndarray, get_state: bool, get_output: bool, dtype=np.float32):
    """LGNN.update_graph for EDGE-based layers.  Follows GNN/LGNN.py:227-260 with :253-254: nodes <- [g.nodes | state?],
    arcs <- [g.arcs | scatter_nd(where(arc mask), output)?]; always from the ORIGINAL graph (:287).  The sparse operands
    are unchanged (same arcs), only the label columns grow."""
    new = dict(g)
    nodes = np.asarray(g['nodes'], dtype=dtype)
    if get_state:
        nodes = np.concatenate([nodes, np.asarray(state, dtype=dtype)], axis=1)
    new['nodes'] = nodes
    if get_output:
        mask = np.logical_and(g['set_mask'], g['output_mask'])
        scat = np.zeros((len(mask), output.shape[1]), dtype=dtype)
        scat[np.nonzero(mask)[0]] = output
        new['arcs'] = np.concatenate([np.asarray(g['arcs'], dtype=dtype), scat], axis=1)
    return new


def lgnn_loop(g: dict, gnns: list[dict], get_state: bool, get_output: bool, graph_based: bool, state0s=None,
              dtype=np.float32):
    """LGNN.Loop.  Follows GNN/LGNN.py:263-290.

    ``gnns``: list of dict(net_state, net_output, state_vect_dim, max_iteration, threshold).
    For graph-based layers the NODE-based Loop feeds update_graph and the readout is appended to outs (:276-278)."""
    gtmp = dict(g)
    ks, outs = [], []
    state0s = state0s or [None] * len(gnns)
    for gnn, s0 in zip(gnns[:-1], state0s[:-1]):
        k, state, out = loop_node(gtmp, gnn['net_state'], gnn['net_output'], gnn['state_vect_dim'],
                                  gnn['max_iteration'], gnn['threshold'], s0, dtype)
        outs.append(np.asarray(gtmp['NodeGraph'], dtype=dtype).T @ out if graph_based else out)
        ks.append(k)
        gtmp = update_graph(g, state, out, get_state, get_output, dtype)
    last = gnns[-1]
    fn = loop_graph if graph_based else loop_node
    k, state, out = fn(gtmp, last['net_state'], last['net_output'], last['state_vect_dim'], last['max_iteration'],
                       last['threshold'], state0s[-1], dtype)
    return ks + [k], state, outs + [out]


# ---------------------------------------------------------------------------------------------------------------------
# helpers for tests
# ---------------------------------------------------------------------------------------------------------------------
def make_graph_dict(arcs, nodes, aggregation_mode='average', set_mask=None, output_mask=None, NodeGraph=None,
                    targets=None, sample_weights=1):
    """Assemble the dict the oracle's Loop functions consume (GraphObject ctor defaults: graph_class.py:42-77)."""
    arcs = np.asarray(arcs, dtype=np.float32)
    nodes = np.asarray(nodes, dtype=np.float32)
    n = nodes.shape[0]
    adjT, arcT = graph_matrices(arcs, n, aggregation_mode)
    set_mask = np.ones(n, dtype=bool) if set_mask is None else np.asarray(set_mask, dtype=bool)
    output_mask = np.ones(n, dtype=bool) if output_mask is None else np.asarray(output_mask, dtype=bool)
    targets = np.zeros((n, 1), dtype=np.float32) if targets is None else np.asarray(targets, dtype=np.float32)
    return dict(arcs=arcs, nodes=nodes, set_mask=set_mask, output_mask=output_mask, adjT=adjT, arcT=arcT,
                NodeGraph=NodeGraph, targets=targets,
                sample_weights=sample_weights * np.ones(targets.shape[0]))
