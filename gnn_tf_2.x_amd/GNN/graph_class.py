# coding=utf-8
"""Graph containers of the MI355X engine, API-compatible with the reference's ``GNN/graph_class.py``.

``GraphObject`` is the host-side (NumPy/SciPy) graph exactly as in the reference (same constructor, attributes, getters,
``merge`` / ``save`` / ``load``).  ``GraphTensor`` is what ``Loop`` consumes: where the reference holds TF constants and two
transposed ``tf.SparseTensor`` (graph_class.py:330-372), this one holds the same matrices as CSR-by-destination arrays and
a lazily created device-resident ``gnn_graph`` handle.
"""
import os
import shutil

import numpy as np
from scipy.sparse import coo_matrix

FLOATX = 'float32'   # the reference reads tf.keras.backend.floatx() (graph_class.py:40); the engine is float32 only
_AGGREGATIONS = ('average', 'normalized', 'sum')


def _transposed_csr(row, col, val, n_rows_t):
    """CSR of the transpose of COO (row, col, val), entries row-major ordered: graph_class.py:365-372 without TF."""
    r_t = np.asarray(col, dtype=np.int64)
    c_t = np.asarray(row, dtype=np.int64)
    order = np.lexsort((c_t, r_t))
    indptr = np.zeros(n_rows_t + 1, dtype=np.int64)
    np.cumsum(np.bincount(r_t, minlength=n_rows_t), out=indptr[1:])
    return indptr.astype(np.int32), c_t[order].astype(np.int32), np.asarray(val, dtype=np.float32)[order], order


class GraphObject:
    """Host graph.  Same constructor and attributes as reference graph_class.py:16-77."""

    def __init__(self, arcs, nodes, targets, problem_based: str = 'n', set_mask=None, output_mask=None,
                 sample_weights=1, NodeGraph=None, ArcNode=None, aggregation_mode: str = 'average'):
        self.dtype = FLOATX
        self.arcs = np.asarray(arcs).astype(self.dtype)
        self.nodes = np.asarray(nodes).astype(self.dtype)
        self.targets = np.asarray(targets).astype(self.dtype)
        self.sample_weights = sample_weights * np.ones(self.targets.shape[0])

        self.DIM_NODE_LABEL = self.nodes.shape[1]
        self.DIM_ARC_LABEL = self.arcs.shape[1] - 2
        self.DIM_TARGET = self.targets.shape[1]

        if problem_based not in ('n', 'a', 'g'):
            raise KeyError(problem_based)
        n_items = self.arcs.shape[0] if problem_based == 'a' else self.nodes.shape[0]
        self.set_mask = np.ones(n_items, dtype=bool) if set_mask is None else np.asarray(set_mask).astype(bool)
        self.output_mask = np.ones(len(self.set_mask), dtype=bool) if output_mask is None else np.asarray(output_mask).astype(bool)
        if len(self.set_mask) != len(self.output_mask):
            raise ValueError('Error - len(<set_mask>) != len(<output_mask>)')

        if aggregation_mode not in _AGGREGATIONS:
            raise ValueError('ERROR: Unknown aggregation mode')
        self.aggregation_mode = aggregation_mode
        self.ArcNode = self.buildArcNode() if ArcNode is None else ArcNode.astype(self.dtype)
        self.Adjacency = self.buildAdiacency()
        self.NodeGraph = self.buildNodeGraph(problem_based) if NodeGraph is None else np.asarray(NodeGraph).astype(self.dtype)

    # ---- matrices ------------------------------------------------------------------------------------------------
    def buildArcNode(self):
        """[n_arcs, n_nodes] with entry (a, dst(a)) = aggregation weight (reference graph_class.py:98-121)."""
        dst = self.arcs[:, 1]
        n_arcs = len(dst)
        weights = np.ones(n_arcs)
        if self.aggregation_mode == 'normalized':
            weights *= float(1 / n_arcs)          # divides by the number of ARCS, as the reference code does
        elif self.aggregation_mode == 'average':
            _, inverse, indegree = np.unique(dst, return_inverse=True, return_counts=True)
            weights /= indegree[inverse]
        return coo_matrix((weights, (np.arange(n_arcs), dst)), shape=(n_arcs, self.nodes.shape[0]), dtype=self.dtype)

    def buildAdiacency(self):
        """[n_nodes, n_nodes] with entry (src, dst) = the arc's ArcNode weight (reference graph_class.py:90-95)."""
        ends = self.arcs[:, :2].astype(int)
        n = self.nodes.shape[0]
        return coo_matrix((self.getArcNode().data, (ends[:, 0], ends[:, 1])), shape=(n, n), dtype=self.dtype)

    def buildNodeGraph(self, problem_based: str):
        """[n_nodes, 1] filled with 1/n_nodes for graph-based problems, else None (reference graph_class.py:132-144)."""
        if problem_based != 'g':
            return None
        n = self.nodes.shape[0]
        return np.ones((n, 1), dtype=np.float32) * 1 / n

    def setAggregation(self, aggregation_mode: str):
        if aggregation_mode not in _AGGREGATIONS:
            raise ValueError('ERROR: Unknown aggregation mode')
        self.aggregation_mode = aggregation_mode
        self.ArcNode = self.buildArcNode()
        self.Adjacency = self.buildAdiacency()

    # ---- copies / getters ------------------------------------------------------------------------------------------
    def copy(self):
        # like the reference (graph_class.py:80-87) the copy does not carry problem_based: it is node-based
        return GraphObject(arcs=self.getArcs(), nodes=self.getNodes(), targets=self.getTargets(), set_mask=self.getSetMask(),
                           output_mask=self.getOutputMask(), sample_weights=self.getSampleWeights(),
                           NodeGraph=self.getNodeGraph(), aggregation_mode=self.aggregation_mode)

    def getArcs(self): return self.arcs.copy()
    def getNodes(self): return self.nodes.copy()
    def getTargets(self): return self.targets.copy()
    def getSetMask(self): return self.set_mask.copy()
    def getOutputMask(self): return self.output_mask.copy()
    def getAdjacency(self): return self.Adjacency.copy()
    def getArcNode(self): return self.ArcNode.copy()
    def getNodeGraph(self): return None if self.NodeGraph is None else self.NodeGraph.copy()
    def getSampleWeights(self): return self.sample_weights.copy()

    # ---- disk format (one folder per graph; same file names as the reference, graph_class.py:192-281) ---------------
    _OPTIONAL = ('set_mask', 'output_mask', 'sample_weights', 'NodeGraph')

    def _files(self):
        items = {'arcs': self.arcs, 'nodes': self.nodes, 'targets': self.targets}
        if not all(self.set_mask): items['set_mask'] = self.set_mask
        if not all(self.output_mask): items['output_mask'] = self.output_mask
        if np.any(self.sample_weights != 1): items['sample_weights'] = self.sample_weights
        if self.NodeGraph is not None and self.targets.shape[0] > 1: items['NodeGraph'] = self.NodeGraph
        return items

    @staticmethod
    def _fresh_dir(path: str) -> str:
        if path[-1] != '/': path += '/'
        if os.path.exists(path): shutil.rmtree(path)
        os.makedirs(path)
        return path

    def save(self, graph_folder_path: str) -> None:
        GraphObject.save_graph(graph_folder_path, self)

    def savetxt(self, graph_folder_path: str, format: str = '%.10g') -> None:
        GraphObject.save_txt(graph_folder_path, self, format)

    @classmethod
    def save_graph(cls, graph_folder_path: str, g):
        path = cls._fresh_dir(graph_folder_path)
        for name, arr in g._files().items():
            np.save(f'{path}{name}.npy', arr)

    @classmethod
    def save_txt(cls, graph_folder_path: str, g, format: str = '%.10g'):
        path = cls._fresh_dir(graph_folder_path)
        for name, arr in g._files().items():
            np.savetxt(f'{path}{name}.txt', arr, fmt=format)

    @classmethod
    def _load(cls, graph_folder_path, problem_based, aggregation_mode, reader):
        if graph_folder_path[-1] != '/': graph_folder_path += '/'
        kwargs = {f.rsplit('.')[0]: reader(graph_folder_path + f) for f in os.listdir(graph_folder_path)}
        return cls(problem_based=problem_based, aggregation_mode=aggregation_mode, **kwargs)

    @classmethod
    def load(cls, graph_folder_path: str, problem_based: str, aggregation_mode: str):
        return cls._load(graph_folder_path, problem_based, aggregation_mode, np.load)

    @classmethod
    def load_txt(cls, graph_folder_path: str, problem_based: str, aggregation_mode: str):
        return cls._load(graph_folder_path, problem_based, aggregation_mode, lambda p: np.loadtxt(p, ndmin=2))

    # ---- batching --------------------------------------------------------------------------------------------------
    @classmethod
    def merge(cls, glist, problem_based: str, aggregation_mode: str):
        """One block-diagonal graph out of a list (reference graph_class.py:285-319): node ids of graph i are offset by
        the node count of the graphs before it, every matrix is rebuilt on the merged arcs, NodeGraph is block_diag."""
        if not (type(glist) == list and all(isinstance(x, (GraphObject, str)) for x in glist)):
            raise TypeError('type of param <glist> must be list of str \'path-like\' or GraphObjects')
        offset, arcs, nodegraphs = 0, [], []
        for g in glist:
            a = g.getArcs()
            a[:, :2] += offset
            arcs.append(a)
            offset += g.nodes.shape[0]
            nodegraphs.append(g.getNodeGraph())
        cat = lambda getter: np.concatenate([getter(g) for g in glist], axis=0)
        nodegraph = None
        if problem_based == 'g':
            from scipy.linalg import block_diag
            nodegraph = block_diag(*nodegraphs)
        return cls(arcs=np.concatenate(arcs, axis=0), nodes=cat(GraphObject.getNodes), targets=cat(GraphObject.getTargets),
                   problem_based=problem_based, set_mask=cat(GraphObject.getSetMask), output_mask=cat(GraphObject.getOutputMask),
                   sample_weights=cat(GraphObject.getSampleWeights), NodeGraph=nodegraph, aggregation_mode=aggregation_mode)

    @classmethod
    def fromGraphTensor(cls, g, problem_based: str):
        return cls(arcs=g.arcs, nodes=g.nodes, targets=g.targets, set_mask=g.set_mask, output_mask=g.output_mask,
                   sample_weights=g.sample_weights, NodeGraph=g.NodeGraph if problem_based == 'g' else None,
                   aggregation_mode=g.aggregation_mode, problem_based=problem_based)


class GraphTensor:
    """What Loop consumes.  Host mirrors of the reference's tensors plus the device handle.

    ``Adjacency`` / ``ArcNode`` are the ALREADY TRANSPOSED matrices (as in reference graph_class.py:342-345), stored as
    ``(indptr, inner, data)`` CSR triples whose entries are row-major ordered (what tf.sparse.reorder produces).
    """

    def __init__(self, nodes, arcs, targets, set_mask, output_mask, sample_weights, Adjacency, ArcNode, NodeGraph,
                 aggregation_mode):
        self.nodes = np.ascontiguousarray(nodes, dtype=np.float32)
        self.arcs = np.ascontiguousarray(arcs, dtype=np.float32)
        self.targets = np.asarray(targets, dtype=np.float32)
        self.sample_weights = np.asarray(sample_weights, dtype=np.float32)
        self.set_mask = np.asarray(set_mask, dtype=bool)
        self.output_mask = np.asarray(output_mask, dtype=bool)
        self.aggregation_mode = aggregation_mode
        self.NodeGraph = None if NodeGraph is None else np.asarray(NodeGraph, dtype=np.float32)
        self.Adjacency = Adjacency
        self.ArcNode = ArcNode
        if not np.array_equal(Adjacency[0], ArcNode[0]):
            raise ValueError('Adjacency and ArcNode describe different in-degree sequences')
        self._device_graph = None

    def copy(self):
        return GraphTensor(nodes=self.nodes, arcs=self.arcs, targets=self.targets, set_mask=self.set_mask,
                           output_mask=self.output_mask, sample_weights=self.sample_weights, Adjacency=self.Adjacency,
                           ArcNode=self.ArcNode, NodeGraph=self.NodeGraph, aggregation_mode=self.aggregation_mode)

    @classmethod
    def fromGraphObject(cls, g: GraphObject):
        return cls(nodes=g.nodes, arcs=g.arcs, targets=g.targets, set_mask=g.set_mask, output_mask=g.output_mask,
                   sample_weights=g.sample_weights, NodeGraph=g.NodeGraph, Adjacency=cls.COO2SparseTransposedTensor(g.Adjacency),
                   ArcNode=cls.COO2SparseTransposedTensor(g.ArcNode), aggregation_mode=g.aggregation_mode)

    @classmethod
    def fromArcs(cls, nodes, arcs, targets, problem_based: str = 'n', set_mask=None, output_mask=None, sample_weights=1,
                 NodeGraph=None, aggregation_mode: str = 'average', device: int = 0):
        """GraphTensor built on the DEVICE from the arc list (``gnn_graph_create_from_arcs``): same tensors as
        ``fromGraphObject(GraphObject(...))`` without the host-side sparse matrices (reference graph_class.py:90-121,
        :365-372), for graphs where that Python would take minutes.  ``arcs`` is [E, 2 + AL] as in GraphObject, used as given
        (GraphObject sorts and de-duplicates its arcs; pass them that way if the arc order matters to you)."""
        from GNN import _engine
        nodes = np.ascontiguousarray(nodes, dtype=np.float32)
        arcs = np.ascontiguousarray(arcs, dtype=np.float32)
        n, e = nodes.shape[0], arcs.shape[0]
        lens = {'n': n, 'a': e, 'g': n}[problem_based]
        set_mask = np.ones(lens, bool) if set_mask is None else np.asarray(set_mask, bool)
        output_mask = np.ones(lens, bool) if output_mask is None else np.asarray(output_mask, bool)
        mask = np.logical_and(set_mask, output_mask) if lens == n else np.ones(n, bool)
        targets = np.asarray(targets, dtype=np.float32)
        if np.isscalar(sample_weights): sample_weights = np.full(targets.shape[0], sample_weights, dtype=np.float32)
        dev, indptr, adj_src, adj_w, arc_id, arc_w = _engine.Graph.from_arcs(n, arcs[:, 0].astype(np.int32), arcs[:, 1].astype(np.int32),
                                                                              arcs[:, 2:], aggregation_mode, nodes, mask, device)
        gt = cls(nodes=nodes, arcs=arcs, targets=targets, set_mask=set_mask, output_mask=output_mask, sample_weights=sample_weights,
                 Adjacency=(indptr, adj_src, adj_w), ArcNode=(indptr, arc_id, arc_w), NodeGraph=NodeGraph, aggregation_mode=aggregation_mode)
        gt._device_graph = dev
        return gt

    @staticmethod
    def COO2SparseTransposedTensor(coo):
        """Transposed, row-major reordered sparse matrix as a CSR triple (reference graph_class.py:365-372)."""
        coo = coo.tocoo()
        indptr, inner, data, _ = _transposed_csr(coo.row, coo.col, coo.data, coo.shape[1])
        return indptr, inner, data

    # ---- device side -------------------------------------------------------------------------------------------------
    def loop_mask(self) -> np.ndarray:
        """set_mask & output_mask (reference GNN.py:275)."""
        return np.logical_and(self.set_mask, self.output_mask)

    def device_graph(self, device: int = 0):
        from GNN import _engine
        if self._device_graph is None:
            indptr, adj_src, adj_w = self.Adjacency
            _, arc_id, arc_w = self.ArcNode
            mask = self.loop_mask()
            if len(mask) != self.nodes.shape[0]:       # arc-based problem: the node mask is unused by the per-arc readout
                mask = np.ones(self.nodes.shape[0], dtype=bool)
            self._device_graph = _engine.Graph(self.nodes.shape[0], indptr, adj_src, adj_w, arc_w,
                                               self.arcs[:, 2:][arc_id], self.nodes, mask, device=device)
        return self._device_graph

    def edge_readout_arrays(self):
        """Inputs of gnn_loop_set_edge_readout: CSR row (destination) of every Adjacency^T entry, the arc labels in ORIGINAL
        arc order, and set_mask & output_mask over arcs (reference GNN.py:289-302 pairs entries and arcs by position)."""
        indptr = self.Adjacency[0]
        entry_dst = np.repeat(np.arange(len(indptr) - 1, dtype=np.int32), np.diff(indptr))
        mask = self.loop_mask()
        if len(mask) != self.arcs.shape[0]:
            raise ValueError('edge-based GNN needs set_mask / output_mask over the arcs (problem_based == \'a\')')
        return entry_dst, self.arcs[:, 2:], mask

    def adjacency_by_source(self):
        """Adjacency (untransposed) as CSR over SOURCE nodes, destinations ascending: the operand of the transposed
        aggregation in the backward pass (d state[src] += w * d agg[dst])."""
        indptr, src, w = self.Adjacency
        n = len(indptr) - 1
        dst = np.repeat(np.arange(n), np.diff(indptr))
        order = np.lexsort((dst, src))
        sip = np.zeros(n + 1, dtype=np.int32)
        np.cumsum(np.bincount(src, minlength=n), out=sip[1:])
        return sip, dst[order].astype(np.int32), np.asarray(w, np.float32)[order]

    def nodegraph_csr(self):
        """NodeGraph^T as CSR over graphs, ascending node inside a graph (input of gnn_loop_readout)."""
        ng = self.NodeGraph
        cached = self.__dict__.get('_ng_csr')
        if cached is not None and cached[0] is ng:          # (np.nonzero over the dense [N, G] matrix is 0.1 ms per call on a MUTAG batch: once per matrix)
            return cached[1]
        cols, rows = np.nonzero(ng.T)
        indptr = np.zeros(ng.shape[1] + 1, dtype=np.int32)
        np.cumsum(np.bincount(cols, minlength=ng.shape[1]), out=indptr[1:])
        csr = (indptr, rows.astype(np.int32), ng[rows, cols].astype(np.float32))
        self.__dict__['_ng_csr'] = (ng, csr)
        return csr
