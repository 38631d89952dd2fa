"""MUTAG ('Mutagenicity', TU format) -> list of graph-based GraphObjects, with the semantics of the reference's loader
(load_MUTAG.py:6-52), quirks included:

* `np.unique(edges, axis=0)` re-sorts the edge list, but the one-hot edge labels stay in FILE order and are then indexed
  by the per-graph mask over the SORTED list, so labels are not aligned with their edges (load_MUTAG.py:28, :41);
* node ids of a graph are renumbered from the set of ids that occur in its edges (:33-36).

Data source: `path` (a folder with the five Mutagenicity_*.txt files) or, by default, the packed copy of the reference's
own data files in tests/golden/mutag_raw.npz.  `graphs` is built lazily through `load()`; `from load_MUTAG import graphs`
works as in the reference (module attribute access triggers the load).
"""
import os

import numpy as np

from GNN.graph_class import GraphObject

_PACKED = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'mutag_raw.npz')


def _raw(path=None):
    if path is None:
        z = np.load(_PACKED)
        return (z['edges'].astype(int), z['edge_labels'].astype(int), z['node_labels'].astype(int),
                z['graph_indicator'].astype(int), z['graph_labels'].astype(int))
    if path[-1] != '/': path += '/'
    # the reference passes delimiter=', ' (two characters), which numpy >= 1.23 rejects; ',' parses the same files
    return (np.loadtxt(path + 'Mutagenicity_edges.txt', dtype=int, delimiter=','),
            np.loadtxt(path + 'Mutagenicity_edge_labels.txt', dtype=int), np.loadtxt(path + 'Mutagenicity_node_labels.txt', dtype=int),
            np.loadtxt(path + 'Mutagenicity_graph_indicator.txt', dtype=int), np.loadtxt(path + 'Mutagenicity_graph_labels.txt', dtype=int))


def _one_hot(labels):
    out = np.zeros((labels.shape[0], len(np.unique(labels))), dtype=int)
    out[np.arange(labels.shape[0]), labels] = 1
    return out


def load(path=None, limit=None):
    edges, edge_labels, node_labels, graph_of_node, graph_labels = _raw(path)
    # first node (0-based position) of every graph, plus the end sentinel (load_MUTAG.py:14-16)
    _, first = np.unique(graph_of_node, return_index=True)
    bounds = np.concatenate([first, [len(graph_of_node)]])
    node_onehot = _one_hot(node_labels)
    edges = np.unique(edges, axis=0)                       # re-sorted; ids are 1-based
    edge_onehot = _one_hot(edge_labels)                    # FILE order (quirk)
    targets = _one_hot(graph_labels)
    # graph g owns the 1-based ids in (bounds[g], bounds[g+1]]; an edge belongs to g when both ends do (load_MUTAG.py:30)
    g_src = np.searchsorted(bounds, edges[:, 0], side='left') - 1
    g_dst = np.searchsorted(bounds, edges[:, 1], side='left') - 1
    inside = g_src == g_dst
    order = np.argsort(g_src[inside], kind='stable')
    rows = np.nonzero(inside)[0][order]
    starts = np.searchsorted(g_src[rows], np.arange(len(bounds)))
    n_graphs = len(bounds) - 1 if limit is None else min(limit, len(bounds) - 1)
    graphs = []
    for g in range(n_graphs):
        sel = rows[starts[g]:starts[g + 1]]                # ascending = the boolean-mask order of the reference
        ids = edges[sel]
        present = np.unique(ids)                           # renumber from the ids that occur in edges (:33-36)
        ids = np.searchsorted(present, ids)
        arcs = np.concatenate([ids, edge_onehot[sel]], axis=1)
        graphs.append(GraphObject(arcs=arcs, nodes=node_onehot[bounds[g]:bounds[g + 1]], targets=targets[g][np.newaxis, ...],
                                  problem_based='g'))
    return graphs


def __getattr__(name):
    if name == 'graphs':
        globals()['graphs'] = load()
        return globals()['graphs']
    raise AttributeError(name)
