"""debug: gradient errors of one wide-layer training step against the float64 oracle (per array, relative to its largest entry)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e
from oracle import gnn_oracle as orc, gnn_train_oracle as tro
from util import make_mlp, random_arcs
from test_gpu_train import _by_source_csr
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6000
rng = np.random.default_rng(n)
d, nl, al, max_it = 64, 3, 1, 3
arcs = random_arcs(rng, n, 4 * n, al)
nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
g = orc.make_graph_dict(arcs, nodes, 'average')
g['set_mask'] = rng.random(n) < 0.9
ACT = os.environ.get('ACT', 'selu')
st = make_mlp(rng, al + 2 * (d + nl), [128, 128, d], ACT, gain=0.7, bn_random=True)
ou = make_mlp(rng, d + nl, [2], 'softmax', batch_normalization=False)
st['dropout'], ou['dropout'] = {}, {}
mask = g['set_mask'] & g['output_mask']
m = int(mask.sum())
targets = np.eye(2)[rng.integers(0, 2, m)].astype(np.float32)
weights = (rng.uniform(0.5, 1.5, m) / m).astype(np.float32)
s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
ref = tro.train_step(g, st, ou, d, max_it, 0.0, s0, [{} for _ in range(max_it)], {}, targets, weights, loss='categorical_crossentropy', mean=False, graph_based=False)
graph = e.Graph(n, g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], np.asarray(g['arcs'])[:, 2:][g['arcT'][1]], nodes, mask)
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], False)
loop = e.Loop(graph, mst, mou, d, max_it, 0.0)
loop.set_state0(s0)
res = loop.train_step(mst, mou, _by_source_csr(g, n), targets, weights, 0, None, dropout_state=[0, 0, 0, 0], dropout_output=[0, 0],
                      bn_state=np.concatenate(st['weights'][-4:-2]), bn_output=None)
print('GNN_TRAIN_MFMA =', os.environ.get('GNN_TRAIN_MFMA', '(default)'), ' k', res['k'], ref['k'], ' loss', res['loss'], ref['loss'])
for name, gl, wl in (('state', res['grads_state'], ref['grads_state']), ('output', res['grads_output'], ref['grads_output'])):
    for got, want in zip(gl, wl):
        print(f'  {name} {str(got.shape):12s} rel err {np.max(np.abs(got - want)) / max(1e-12, np.max(np.abs(want))):.2e}   max |want| {np.max(np.abs(want)):.3e}')
