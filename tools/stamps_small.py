"""Phase stamps of the fused kernel on one MUTAG batch (per-body launches, body 1).  Run on the GPU box with
GNN_FUSED_STAMPS=<file> set; then tools/stamps.py <file>."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e
from GNN.graph_class import GraphObject, GraphTensor
from util import make_mlp
import load_MUTAG
rng = np.random.default_rng(1)
b = GraphObject.merge(load_MUTAG.load(limit=32), problem_based='g', aggregation_mode='average')
st, ou = make_mlp(rng, 31, [32, 32, 14], 'selu', gain=0.7), make_mlp(rng, 14, [2], 'softmax')
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
loop = e.Loop(GraphTensor.fromGraphObject(b).device_graph(), mst, mou, 0, 50, 0.01)
loop.set_impl(2)
for _ in range(3):
    print(loop.run())
