"""Graph build at BASELINE config 3 size (1M nodes / 10M arcs): device-side gnn_graph_create_from_arcs against the host chain
(NumPy restatement of buildArcNode / buildAdiacency / COO2SparseTransposedTensor + upload).  Run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd')):
    sys.path.insert(0, p)
from GNN import GNN_utils as utils, _engine
from GNN.graph_class import GraphTensor
from oracle import gnn_oracle as orc

s = utils.syntheticGraph(1_000_000, 10.0, 3, 1, 2, seed=1)
arcs = np.concatenate([s['src'][:, None].astype(np.float32), s['dst'][:, None].astype(np.float32), s['arc_labels']], axis=1)
n = s['n_nodes']
_engine.require_device(0)
GraphTensor.fromArcs(s['nodes'][:1000], arcs[:10][:, :3] * 0, np.zeros((1000, 2)))      # warm up the library
t = time.perf_counter()
gt = GraphTensor.fromArcs(s['nodes'], arcs, s['targets'], aggregation_mode='average')
t_dev = time.perf_counter() - t
t = time.perf_counter()
adjT, arcT = orc.graph_matrices(arcs, n, 'average')
t_host_build = time.perf_counter() - t
t = time.perf_counter()
g = _engine.Graph(n, adjT[0], adjT[1], adjT[2], arcT[2], arcs[:, 2:][arcT[1]], s['nodes'], np.ones(n, np.uint8))
t_upload = time.perf_counter() - t
same = all(np.array_equal(a, b) for a, b in zip(gt.Adjacency, adjT)) and all(np.array_equal(a, b) for a, b in zip(gt.ArcNode, arcT))
print(f'N={n} E={len(arcs)}: device build (incl. H2D of the arc list and D2H of the mirrors) {t_dev:.3f} s; '
      f'host build {t_host_build:.3f} s + upload {t_upload:.3f} s; identical arrays: {same}')
