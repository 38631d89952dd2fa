"""debug: replay tests/test_gpu_train.py::test_train_step_random_shapes case by case with per-array errors"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
from GNN import _engine as e
from oracle import gnn_oracle as orc, gnn_train_oracle as tro
from util import make_mlp, random_arcs
from test_gpu_train import _by_source_csr
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
rng = np.random.default_rng(20261006)
for case in range(14):
    d = int(rng.choice([0, 1, 4, 8, 16, 33, 64]))
    nl = int(rng.integers(1, 6)) if d else int(rng.choice([2, 7, 16]))
    al = int(rng.integers(1, 4))
    hidden = [int(x) for x in rng.choice([1, 9, 32, 64, 100, 128], size=int(rng.integers(0, 3)))]
    n = int(rng.choice([40, 333, 2000, 5000, 9000]))
    act = ['tanh', 'sigmoid', 'linear'][case % 3]
    bn = bool(case % 4 != 3)
    max_it = int(rng.integers(1, 5))
    arcs = random_arcs(rng, n, int(rng.choice([1, 3, 8])) * n, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    mode = str(rng.choice(['average', 'sum', 'normalized']))
    g = orc.make_graph_dict(arcs, nodes, mode)
    g['set_mask'] = rng.random(n) < 0.7
    ds, nlc = (d if d else nl), (nl if d else 0)
    st = make_mlp(rng, al + 2 * (ds + nlc), hidden + [ds], act, gain=0.5, bn_random=True, batch_normalization=bn)
    ou = make_mlp(rng, ds + nlc, [2], 'softmax', batch_normalization=False)      # (BatchNormalization behind a softmax leaves [0, 1]: the loss clips, its gradient is ill-conditioned in float32 and float64 alike)
    st['dropout'], ou['dropout'] = {}, {}
    mask = g['set_mask'] & g['output_mask']
    m = int(mask.sum())
    targets = np.eye(2)[rng.integers(0, 2, m)].astype(np.float32)
    weights = (rng.uniform(0.5, 1.5, m) / m).astype(np.float32)
    s0 = (0.1 * rng.standard_normal((n, ds))).astype(np.float32) if d else None
    if only >= 0 and case != only: continue
    ref = tro.train_step(g, st, ou, d, max_it, 0.0, s0, [{} for _ in range(max_it)], {}, targets, weights, loss='categorical_crossentropy', mean=False, graph_based=False)
    ref32 = tro.train_step(g, st, ou, d, max_it, 0.0, s0, [{} for _ in range(max_it)], {}, targets, weights, loss='categorical_crossentropy', mean=False, graph_based=False, dtype=np.float32)
    graph = e.Graph(n, g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], np.asarray(g['arcs'])[:, 2:][g['arcT'][1]], nodes, mask)
    mst, mou = e.Mlp(st['weights'], st['activations'], bn), e.Mlp(ou['weights'], ou['activations'], False)
    loop = e.Loop(graph, mst, mou, d, max_it, 0.0)
    if d: loop.set_state0(s0)
    kw = dict(dropout_state=[0.0] * (len(hidden) + 2), dropout_output=[0.0, 0.0],
              bn_state=np.concatenate(st['weights'][-4:-2]) if bn else None, bn_output=None)
    res = loop.train_step(mst, mou, _by_source_csr(g, n), targets, weights, 0, None, **kw)
    print(f'case {case}: d={d} nl={nl} al={al} hidden={hidden} n={n} act={act} bn={bn} it={max_it} mode={mode} k={res["k"]}/{ref["k"]} loss {res["loss"]:.6f}/{ref["loss"]:.6f} max|state| {np.max(np.abs(loop.state())):.3g}')
    for name, gl, wl, w32 in (('s', res['grads_state'], ref['grads_state'], ref32['grads_state']), ('o', res['grads_output'], ref['grads_output'], ref32['grads_output'])):
        for got, want, w3 in zip(gl, wl, w32):
            mx = max(1e-30, float(np.max(np.abs(want))))
            print(f'   {name} {str(got.shape):10s} rel err {np.max(np.abs(got - want)) / mx:.2e}  (float32 oracle vs float64: {np.max(np.abs(w3 - want)) / mx:.2e})  max {mx:.3g}')
    loop.close()
