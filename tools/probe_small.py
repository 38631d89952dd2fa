import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e
from GNN.graph_class import GraphObject, GraphTensor
from util import make_mlp
import load_MUTAG
rng = np.random.default_rng(1)
graphs = load_MUTAG.load(limit=32)
b = GraphObject.merge(graphs, problem_based='g', aggregation_mode='average')
st, ou = make_mlp(rng, 31, [32, 32, 14], 'selu', gain=0.7), make_mlp(rng, 14, [2], 'softmax')
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
gt = GraphTensor.fromGraphObject(b)
for max_it in (50, 13, 1):
    loop = e.Loop(gt.device_graph(), mst, mou, 0, max_it, 0.01)
    for _ in range(3): loop.run()
    t = time.perf_counter()
    dev = 0
    for _ in range(200):
        k = loop.run(); dev += loop.timing()['total_ms']
    dt = time.perf_counter() - t
    print(f'max_it={max_it} k={k} wall/loop={1e6*dt/200:.1f} us  device(total_ms)/loop={1e3*dev/200:.1f} us')
