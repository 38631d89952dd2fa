"""Phase stamps of the persistent small-graph loop on one MUTAG batch (DIAG build, workgroup 0).  Run on the GPU box:
GNN_SMALL_STAMPS=gpurun_out/small_stamps.bin python tools/stamps_small.py"""
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
loop.set_impl(1)
for _ in range(3):
    k = loop.run()
f = os.environ.get('GNN_SMALL_STAMPS')
if f and os.path.exists(f):
    a = np.fromfile(f, dtype=np.uint64).astype(np.int64)
    a = a[a > 0]
    d = np.diff(a)
    k = int(k)
    print('k', k, 'stamps', len(a), 'total ticks', a[-1] - a[0], '(s_memtime ticks: 100 MHz => 10 ns each)')
    print('set-up loads', d[0], ' init + condition', d[1], ' gate 0', d[2])
    body = d[3:3 + 4 * k].reshape(k, 4)
    print('per body (median / first / last):')
    for i, n in enumerate(['gather', 'dense layers', 'condition + store drain', 'barrier + gate']):
        print(f'  {n:26s} {np.median(body[:, i]):8.0f} {body[0, i]:8d} {body[-1, i]:8d}')
    print('output stage', d[3 + 4 * k:])
