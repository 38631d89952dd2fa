"""Wall time of one training step (gnn_loop_train_step through GNNgraphBased.training_step) on MUTAG batches of 32, on a 100k-node
synthetic graph and (C3=1, or C3_ONLY=1 to skip the others) on the BASELINE configs[2] shape: 1M nodes / 10M arcs, state_dim 64,
net_state 135 -> 128 -> 128 -> 64, 5 bodies.  Run on the GPU box: python tools/bench_train.py"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import losses, optimizers, GNN_utils as utils
from GNN.GNN import GNNgraphBased, GNNnodeBased
from GNN.MLP import MLP, set_seed
from GNN.graph_class import GraphObject, GraphTensor
import load_MUTAG

def c3_shaped():
    """gnn_loop_train_step on the bench workload (bench.py: graph, nets, initial state), 5 bodies, no optimizer step between the calls
    (random-init weights give an expansive map: an update would change k), all-true masks, categorical cross-entropy."""
    sys.path.insert(0, ROOT)
    import bench
    from GNN import _engine as e
    N, d = 1_000_000, 64
    s = utils.syntheticGraph(N, 10, 3, 1, 2, seed=20261003)
    n = s['n_nodes']
    rng = np.random.default_rng(20261003)
    st = bench.make_net(rng, 1 + 2 * (3 + d), [128, 128, d], 'selu')
    ou = bench.make_net(rng, 3 + d, [2], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    loop = e.Loop(graph, mst, mou, d, 5, 0.0)
    loop.set_state0(s0)
    targets = np.eye(2, dtype=np.float32)[rng.integers(0, 2, n)]
    weights = np.full(n, 1.0 / n, np.float32)
    bn_s, bn_o = np.concatenate(st['weights'][-4:-2]), np.concatenate(ou['weights'][-4:-2])
    step = lambda: loop.train_step(mst, mou, None, targets, weights, 0, None, dropout_state=[0, 0, 0, 0], dropout_output=[0, 0], bn_state=bn_s, bn_output=bn_o)
    r = step()
    reps = int(os.environ.get('C3_REPS', 3))
    t = time.perf_counter()
    for _ in range(reps):
        r = step()
    dt = (time.perf_counter() - t) / reps
    print(f'C3-shaped gnn_loop_train_step (1M nodes / {s["n_arcs"]} arcs, 135->128->128->64 selu+BN, 67->2 softmax+BN, k={r["k"]}, loss {r["loss"]:.4f}): '
          f'{1e3 * dt:.1f} ms/step, {n * r["k"] / dt:.3e} node-updates/s (fwd+bwd)   [GNN_TRAIN_MFMA={os.environ.get("GNN_TRAIN_MFMA", "default")}]', flush=True)


set_seed(0)
if os.environ.get('C3_ONLY'):
    c3_shaped()
    sys.exit(0)
graphs = load_MUTAG.load(limit=128)
batches = [GraphTensor.fromGraphObject(GraphObject.merge(graphs[i:i + 32], problem_based='g', aggregation_mode='average')) for i in range(0, 128, 32)]
st = MLP(3 + 2 * 14, [32, 32, 14], 'selu', 'glorot_normal', 'zeros', dropout_rate=0.1, dropout_pos=0)
ou = MLP(14, [2], 'softmax', 'glorot_normal', 'zeros', batch_normalization=False)
gnn = GNNgraphBased(net_state=st, net_output=ou, optimizer=optimizers.Adam(0.001), loss_function=losses.categorical_crossentropy, loss_arguments=None,
                    state_vect_dim=0, max_iteration=10, threshold=0.001, addressed_problem='c')
for b in batches: gnn.training_step(b, True)
from GNN import _engine as _e
_orig = _e.Loop.train_step
_acc = [0.0]
def _timed(self, *a, **k):
    t0 = time.perf_counter(); r = _orig(self, *a, **k); _acc[0] += time.perf_counter() - t0; return r
_e.Loop.train_step = _timed
t = time.perf_counter(); n = 0
for _ in range(10):
    for b in batches:
        r = gnn.training_step(b, True); n += 1
print('  last k', r['k'], 'loss', r['loss'])
dt = time.perf_counter() - t
print(f'MUTAG batch-32 training_step: {1e3 * dt / n:.2f} ms/step  (k={r["k"]}), {32 * n / dt:.0f} graphs/s; inside gnn_loop_train_step: {1e3 * _acc[0] / n:.2f} ms')
_acc[0] = 0.0

if os.environ.get("SMALL_ONLY"): sys.exit(0)
N = 100_000
s = utils.syntheticGraph(N, 10, 3, 1, 2, seed=3)
arcs = np.concatenate([s['src'][:, None].astype(np.float32), s['dst'][:, None].astype(np.float32), s['arc_labels']], axis=1)
go = GraphObject(arcs=arcs, nodes=s['nodes'], targets=s['targets'], problem_based='n', aggregation_mode='average')
if True:
    gt = GraphTensor.fromGraphObject(go)
    st = MLP(1 + 2 * (3 + 16), [64, 16], 'selu', 'glorot_normal', 'zeros')
    ou = MLP(3 + 16, [2], 'softmax', 'glorot_normal', 'zeros', batch_normalization=False)
    g2 = GNNnodeBased(net_state=st, net_output=ou, optimizer=optimizers.Adam(0.001), loss_function=losses.categorical_crossentropy, loss_arguments=None,
                      state_vect_dim=16, max_iteration=5, threshold=0.0, addressed_problem='c')
    g2.training_step(gt, True)
    t = time.perf_counter()
    for _ in range(5):
        r = g2.training_step(gt, True); print('  k', r['k'], 'loss', r['loss'])
    dt = (time.perf_counter() - t) / 5
    print(f'100k-node training_step: {1e3 * dt:.1f} ms/step (k={r["k"]}), {N * r["k"] / dt:.3e} node-updates/s (fwd+bwd)')

if os.environ.get('C3'):
    c3_shaped()
