"""debug: test_train_step_matches_oracle[8-False-tanh-...] per-array errors vs float64 and the float32 oracle's own error"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
from GNN import _engine as e
from oracle import gnn_oracle as orc, gnn_train_oracle as tro
from util import make_mlp, random_arcs
from test_gpu_train import _by_source_csr
d, act, loss = 8, 'tanh', 'categorical_crossentropy'
rng = np.random.default_rng(100 + d)
n, nl, al, max_it = 500, 3, 2, 6
arcs = random_arcs(rng, n, 1500, al)
nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
g = orc.make_graph_dict(arcs, nodes, 'average')
g['set_mask'] = rng.random(n) < 0.8
ds, nlc = d, nl
st = make_mlp(rng, al + 2 * (ds + nlc), [16, ds], act, gain=0.8, bn_random=True)
ou = make_mlp(rng, ds + nlc, [9, 2], act, out_activation='softmax', bn_random=True)
st['dropout'], ou['dropout'] = {0: 0.2}, {0: 0.1, 1: 0.3}
mask = g['set_mask'] & g['output_mask']
m = int(mask.sum())
in_s = st['weights'][0].shape[0]
masks_s = [{0: (rng.random((n, in_s)) > 0.2)} for _ in range(max_it)]
masks_o = {0: rng.random((m, ds + nlc)) > 0.1, 1: rng.random((m, 9)) > 0.3}
targets = np.eye(2)[rng.integers(0, 2, m)].astype(np.float32)
weights = rng.uniform(0.5, 1.5, m).astype(np.float32)
s0 = (0.1 * rng.standard_normal((n, ds))).astype(np.float32)
ref = tro.train_step(g, st, ou, d, max_it, 0.0, s0, masks_s, masks_o, targets, weights, loss=loss, mean=False, graph_based=False)
r32 = tro.train_step(g, st, ou, d, max_it, 0.0, s0, masks_s, masks_o, targets, weights, loss=loss, mean=False, graph_based=False, dtype=np.float32)
graph = e.Graph(n, g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], np.asarray(g['arcs'])[:, 2:][g['arcT'][1]], nodes, mask)
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
loop = e.Loop(graph, mst, mou, d, max_it, 0.0)
loop.set_state0(s0)
ms = np.concatenate([masks_s[k][0].astype(np.uint8).ravel() for k in range(max_it)])
mo = np.concatenate([masks_o[0].astype(np.uint8).ravel(), masks_o[1].astype(np.uint8).ravel()])
res = loop.train_step(mst, mou, _by_source_csr(g, n), targets, weights, 0, None, dropout_state=[0.2, 0, 0], dropout_output=[0.1, 0.3, 0], masks_state=ms, masks_output=mo,
                      bn_state=np.concatenate(st['weights'][-4:-2]), bn_output=np.concatenate(ou['weights'][-4:-2]))
print('BN_SMALL =', os.environ.get('GNN_TRAIN_BN_SMALL', '(default)'), 'k', res['k'], ref['k'], 'loss', res['loss'], ref['loss'])
for name, gl, wl, w3 in (('s', res['grads_state'], ref['grads_state'], r32['grads_state']), ('o', res['grads_output'], ref['grads_output'], r32['grads_output'])):
    for got, want, x3 in zip(gl, wl, w3):
        mx = float(np.max(np.abs(want)))
        print(f'  {name} {str(got.shape):9s} rel err {np.max(np.abs(got - want)) / mx:.2e}   float32 oracle {np.max(np.abs(x3 - want)) / mx:.2e}   max {mx:.3g}')
