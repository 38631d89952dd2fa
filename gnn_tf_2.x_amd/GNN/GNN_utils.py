"""Host-side dataset helpers with the reference's names (``GNN/GNN_utils.py``) plus the large synthetic generator used by
``bench.py``.  Nothing here is on the hot path."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

from GNN.graph_class import GraphObject, GraphTensor


def _cluster_targets(features: np.ndarray, n_classes: int) -> np.ndarray:
    from sklearn.cluster import AgglomerativeClustering
    labels = AgglomerativeClustering(n_clusters=n_classes).fit(features).labels_
    onehot = np.zeros((features.shape[0], n_classes))
    onehot[np.arange(features.shape[0]), labels] = 1
    return onehot


def _symmetric_random_arcs(rand, nodes_number: int, n_draws: int, dim_arc_label: int) -> np.ndarray:
    """Arc recipe of reference GNN_utils.py:36-57: draw (src, dst > src) pairs, de-duplicate, mirror, share the label
    between (i, j) and (j, i), sort lexicographically."""
    src = rand.choice(range(nodes_number)[:-1], n_draws)
    dst = src + np.ceil((np.ones_like(src) * nodes_number - src - 1) * rand.random(len(src)))
    ascending = np.unique(np.stack([src, dst], axis=1).astype(float), axis=0)
    labels = 2 * rand.random((ascending.shape[0], dim_arc_label)) - 1
    ids = np.concatenate((ascending, np.flip(ascending, axis=1)))
    return np.unique(np.concatenate((ids, np.concatenate((labels, labels))), axis=1), axis=0)


def randomGraph(nodes_number: int, dim_node_label: int, dim_arc_label: int, dim_target: int, density: float,
                *, normalize_features: bool = False, aggregation_mode: str = 'average', problem_based: str = 'n') -> GraphObject:
    """Random symmetric graph with uniform(-1, 1) labels and clustered one-hot targets (reference GNN_utils.py:16-84).
    Uses NumPy's global generator in the reference's call order, so ``np.random.seed(s)`` reproduces its graphs."""
    assert problem_based in ('n', 'a', 'g')
    nodes = 2 * np.random.random((nodes_number, dim_node_label)) - 1
    arcs_number = round(density * nodes_number * (nodes_number - 1) / 2)
    arcs = _symmetric_random_arcs(np.random, nodes_number, arcs_number // 2, dim_arc_label)
    if problem_based == 'g':
        targs = np.zeros((1, dim_target))
        targs[0, np.random.choice(range(dim_target))] = 1
    else:
        targs = _cluster_targets(arcs[:, 2:] if problem_based == 'a' else nodes, dim_target)
    output_mask = np.ones(arcs.shape[0] if problem_based == 'a' else nodes.shape[0], dtype=bool)
    if normalize_features:
        nodes = nodes / np.max(nodes, axis=0)
        arcs[:, 2:] = arcs[:, 2:] / np.max(arcs[:, 2:], axis=0)
    return GraphObject(arcs=arcs, nodes=nodes, targets=targs, problem_based=problem_based, output_mask=output_mask,
                       aggregation_mode=aggregation_mode)


def simple_graph(problem_based: str, aggregation_mode: str = 'average') -> GraphObject:
    """The reference's 4-node / 8-arc debugging graph (GNN_utils.py:88-105)."""
    nodes = np.array([[11, 21], [12, 22], [13, 23], [14, 24]])
    arcs = np.array([[0, 1, 10], [0, 2, 40], [1, 0, 10], [1, 2, 20], [2, 0, 40], [2, 1, 20], [2, 3, 30], [3, 2, 30]])
    if problem_based == 'g':
        targs = np.array([[0., 1.]])
    else:
        targs = _cluster_targets(arcs[:, 2:] if problem_based == 'a' else nodes, 2)
    return GraphObject(arcs=arcs, nodes=nodes, targets=targs, problem_based=problem_based, aggregation_mode=aggregation_mode)


def syntheticGraph(nodes_number: int, arcs_per_node: float = 10.0, dim_node_label: int = 3, dim_arc_label: int = 1,
                   dim_target: int = 2, seed: int = 20261003) -> dict:
    """Large synthetic workload: the ``randomGraph`` arc recipe with ``nodes_number * arcs_per_node / 2``
    undirected draws, WITHOUT the O(n^2) clustering (targets are random one-hot) and without building SciPy matrices.

    Returns plain arrays ready for the engine: arcs [E, 2+AL] float32 (lexicographically sorted, symmetric, duplicate and
    self-loop free), nodes [N, NL] float32, targets, and the CSR-by-destination triples of 'average' aggregation."""
    rng = np.random.default_rng(seed)
    n = int(nodes_number)
    draws = int(n * arcs_per_node / 2)
    src = rng.integers(0, n - 1, draws, dtype=np.int64)
    dst = src + np.ceil((n - 1 - src) * rng.random(draws)).astype(np.int64)
    key = np.unique(src * n + dst)                     # de-duplicate undirected pairs (src < dst)
    src, dst = key // n, key % n
    lab = (2 * rng.random((len(key), dim_arc_label)) - 1).astype(np.float32)
    s = np.concatenate([src, dst])
    d = np.concatenate([dst, src])
    lab = np.concatenate([lab, lab])
    order = np.argsort(s * n + d, kind='stable')       # lexicographic (src, dst): the reference's np.unique(arcs, axis=0)
    s, d, lab = s[order], d[order], lab[order]
    nodes = (2 * rng.random((n, dim_node_label)) - 1).astype(np.float32)
    targets = np.zeros((n, dim_target), dtype=np.float32)
    targets[np.arange(n), rng.integers(0, dim_target, n)] = 1
    # CSR by destination.  Arcs are (src, dst)-sorted, so for a fixed dst ascending arc id == ascending src: one permutation
    # serves both Adjacency^T and ArcNode^T.
    by_dst = np.argsort(d * n + s, kind='stable')
    indeg = np.bincount(d, minlength=n)
    indptr = np.zeros(n + 1, dtype=np.int32)
    np.cumsum(indeg, out=indptr[1:])
    w = (1.0 / indeg[d[by_dst]]).astype(np.float32)    # 'average': 1 / in-degree of the destination
    return dict(n_nodes=n, n_arcs=len(s), src=s.astype(np.int32), dst=d.astype(np.int32), arc_labels=lab, nodes=nodes,
                targets=targets, indptr=indptr, adj_src=s[by_dst].astype(np.int32), adj_w=w, arc_perm=by_dst.astype(np.int32),
                arc_w=w, arc_labels_csr=np.ascontiguousarray(lab[by_dst]), max_in_degree=int(indeg.max()))


def progressbar(percent: float, width: int = 30) -> None:
    done = round(width * percent / 100)
    print('\r[', '#' * done, ' ' * int(width - done), ']', f' {percent:.1f}%', sep='', end='', flush=True)


def getindices(len_dataset: int, perc_Train: float = 0.7, perc_Valid: float = 0.1, seed=None):
    """(train, test, validation) index lists (reference GNN_utils.py:117-150; note the return order)."""
    if perc_Train < 0 or perc_Valid < 0 or perc_Train + perc_Valid > 1:
        raise ValueError('Error - percentage must stay in [0-1] and their sum must be <= 1')
    idx = list(range(len_dataset))
    if seed: np.random.seed(seed)
    if seed is not False: np.random.shuffle(idx)
    n_test = round(len_dataset * (1 - perc_Train - perc_Valid))
    n_valid = round(len_dataset * perc_Valid)
    return idx[n_test + n_valid:], idx[:n_test], idx[n_test:n_test + n_valid]


def getSet(glist: list[str], set_indices: list[int], problem_based: str, aggregation_mode: str, verbose: bool = False) -> list[GraphObject]:
    if not (type(glist) == list and all(isinstance(x, str) for x in glist)):
        raise TypeError('type of param <glist> must be list of str \'path-like\' or GraphObjects')
    chosen = []
    for i, elem in enumerate(set_indices):
        chosen.append(glist[elem])
        if verbose: progressbar((i + 1) * 100 / len(set_indices))
    return [GraphObject.load(p, problem_based=problem_based, aggregation_mode=aggregation_mode) for p in chosen]


def getbatches(glist: list[GraphObject], problem_based: str, aggregation_mode: str, batch_size: int = 32, number_of_batches=None,
               one_graph_per_batch=True):
    """Split a list of graphs into batches; by default each batch is merged into one block-diagonal GraphObject
    (reference GNN_utils.py:177-195)."""
    if number_of_batches is None:
        batches = [glist[i:i + batch_size] for i in range(0, len(glist), batch_size)]
    else:
        batches = [list(i) for i in np.array_split(glist, number_of_batches)]
    if one_graph_per_batch:
        batches = [GraphObject.merge(b, problem_based=problem_based, aggregation_mode=aggregation_mode) for b in batches]
    return batches


def normalize_graphs(gTr, gVa, gTe, based_on: str = 'gTr', norm_rangeN: Optional[tuple] = None, norm_rangeA: Optional[tuple] = None) -> None:
    """Min-max scale node and arc matrices in place, fitted on gTr (or on everything).  As in the reference
    (GNN_utils.py:198-234) the arc scaler is fitted on ALL arc columns, the two id columns included: harmless only because
    ArcNode/Adjacency were built before and Loop reads arcs[:, 2:] alone."""
    def as_list(g, name):
        if g is None: return []
        if not (type(g) == GraphObject or (type(g) == list and all(isinstance(x, GraphObject) for x in g))):
            raise TypeError(f'type of param <{name}> must be GraphObject or list of Graphobjects')
        return g if type(g) == list else [g]

    gTr, gVa, gTe = as_list(gTr, 'gTr'), as_list(gVa, 'gVa'), as_list(gTe, 'gTe')
    if based_on not in ['gTr', 'all']: raise ValueError('param <based_on> must be \'gTr\' or \'all\'')
    fit_on = gTr if based_on == 'gTr' else gTr + gTe + gVa
    G = GraphObject.merge(fit_on, problem_based='n', aggregation_mode='sum')
    from sklearn.preprocessing import MinMaxScaler
    node_scaler = MinMaxScaler(feature_range=(0, 1) if norm_rangeN is None else norm_rangeN).fit(G.nodes)
    arcs_scaler = MinMaxScaler(feature_range=(0, 1) if norm_rangeA is None else norm_rangeA).fit(G.arcs)
    for g in gTr + gVa + gTe:
        g.nodes = node_scaler.transform(g.nodes)
        g.arcs = arcs_scaler.transform(g.arcs)
