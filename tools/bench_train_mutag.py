"""MUTAG training steps with a net whose loop converges before max_iteration (k about 12 of 50, as in bench.py's other_configs): the case the
iteration-count hint of train_forward is for.  Engine level (gnn_loop_train_step + armed Adam), 10 batches of 32, host-timed."""
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
graphs = load_MUTAG.load(limit=320)
batches = [GraphTensor.fromGraphObject(GraphObject.merge(graphs[i:i + 32], problem_based='g', aggregation_mode='average')) for i in range(0, 320, 32)]
st, ou = make_mlp(rng, 31, [32, 32, 14], 'selu', gain=0.7), make_mlp(rng, 14, [2], 'softmax', batch_normalization=False)
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], False)
items = []
for b in batches:
    lp = e.Loop(b.device_graph(), mst, mou, 0, 50, 0.01)
    g = b.targets.shape[0]
    items.append((lp, b.nodegraph_csr(), np.asarray(b.targets, np.float32), np.full(g, 1.0 / g, np.float32)))
def epoch():
    ks = 0.0
    for lp, ng, t, w in items:
        lp.arm_optimizer(1, [1e-4, 0.9, 0.999, 1e-7], True)
        ks += lp.train_step(mst, mou, None, t, w, 0, ng, bn_state=None, bn_output=None)['k']
    return ks / len(items)
for _ in range(3): epoch()
t0 = time.perf_counter()
reps = 20
k = sum(epoch() for _ in range(reps)) / reps
dt = time.perf_counter() - t0
print(f'MUTAG batch-32 training step, converging net (mean k = {k:.1f} of 50): {1e3 * dt / (reps * len(items)):.3f} ms per step   [GNN_TRAIN_K_HINT={os.environ.get("GNN_TRAIN_K_HINT", "default")}]')
