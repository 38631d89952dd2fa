"""Independent semantics cross-check of the UNPINNED half of the oracle (the TensorFlow/Keras ops the reference calls:
sparse_dense_matmul, Dense, selu / tanh / sigmoid / softmax, BatchNormalization(epsilon=1e-3) at inference, the while-loop
of GNN/GNN.py:202-280) against PyTorch-CPU equivalents.  Run in the BUILD container only (torch is never on the product's
path): it writes tests/golden/torch_crosscheck.npz = inputs + what torch computed; tests/test_oracle.py compares the
oracle (NumPy f32 / f64 and the C restatement) with those vectors.

This does NOT pin parity with TensorFlow (nothing here is reference-held); it removes the single-author risk on constants
and op semantics: selu scale / alpha, softmax axis, BatchNormalization formula and epsilon, strict '>' of the convergence
test, state_old = ones, concat order [state | nodes? | agg_state | agg_nodes | agg_arcs] (GNN.py:228-237).

    python tests/golden/make_torch_crosscheck.py
"""
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
for p in (ROOT, os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from oracle import gnn_oracle as orc      # noqa: E402  (only its GRAPH half: matrices pinned by the reference's own fixtures)
from util import make_mlp, random_arcs    # noqa: E402

ACT = {'linear': lambda x: x, 'relu': torch.relu, 'selu': torch.selu, 'elu': F.elu, 'tanh': torch.tanh, 'sigmoid': torch.sigmoid,
       'softmax': lambda x: torch.softmax(x, dim=-1)}


def t_mlp(x, net, dt):
    w = [torch.tensor(a, dtype=dt) for a in net['weights']]
    n = len(net['activations'])
    for l in range(n):
        x = ACT[net['activations'][l]](F.linear(x, w[2 * l].T, w[2 * l + 1]))          # Dense: act(x . W + b)
    if net['batch_normalization']:
        gamma, beta, mean, var = w[2 * n:2 * n + 4]
        x = F.batch_norm(x, mean, var, gamma, beta, training=False, eps=1e-3)          # Keras default epsilon
    return x


def t_csr(csr, n_cols, dt):
    indptr, inner, val = csr
    return torch.sparse_csr_tensor(torch.tensor(indptr, dtype=torch.int64), torch.tensor(inner, dtype=torch.int64), torch.tensor(val, dtype=dt),
                                   size=(len(indptr) - 1, n_cols))


def t_loop(g, st, ou, d, max_it, thr, s0, dt):
    """GNN/GNN.py:251-280 in torch."""
    nodes = torch.tensor(g['nodes'], dtype=dt)
    arcs = torch.tensor(np.asarray(g['arcs'])[:, 2:], dtype=dt)
    n = nodes.shape[0]
    adj = t_csr(g['adjT'], n, dt)
    arcn = t_csr(g['arcT'], arcs.shape[0], dt)
    agg_arcs = arcn @ arcs if arcs.shape[1] else torch.zeros((n, 0), dtype=dt)          # :259
    if d > 0:
        state = torch.tensor(s0, dtype=dt)
        agg_nodes = adj @ nodes                                                           # :263
    else:
        state = nodes.clone()                                                             # :265
        agg_nodes = torch.zeros((n, 0), dtype=dt)
    state_old = torch.ones_like(state)                                                    # :266
    k = 0

    def cond():
        dist = torch.sqrt(torch.sum((state - state_old) ** 2, dim=1))                    # :209-211
        norm = torch.sqrt(torch.sum(state_old ** 2, dim=1))                              # :212
        return bool(torch.any(dist > thr * norm)) and k < max_it                         # :215-220

    while cond():
        comps = [state] + ([nodes] if d > 0 else [])                                      # :228-230
        inp = torch.cat(comps + [adj @ state, agg_nodes, agg_arcs], dim=1)                # :234-237
        state_old, state = state, t_mlp(inp, st, dt)                                      # :240-242
        k += 1
    mask = torch.tensor(np.logical_and(g['set_mask'], g['output_mask']))
    feats = torch.cat([state, nodes], dim=1) if d > 0 else state                         # :245-248
    out = t_mlp(feats[mask], ou, dt)                                                      # :279
    return k, state.numpy(), out.numpy()


def main():
    torch.set_num_threads(1)
    out = {}
    cases = [('selu_d0', 0, 3, 1, (7,), 'selu', 'average'), ('tanh_d8', 8, 3, 2, (16,), 'tanh', 'sum'), ('sigmoid_d5', 5, 2, 1, (6, 9), 'sigmoid', 'normalized'),
             ('selu_d16_deep', 16, 4, 1, (24, 24), 'selu', 'average')]
    for ci, (name, d, nl, al, hidden, act, mode) in enumerate(cases):
        rng = np.random.default_rng(900 + ci)
        n = 150
        arcs = random_arcs(rng, n, 4 * n, al)
        nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
        g = orc.make_graph_dict(arcs, nodes, mode)
        g['set_mask'] = rng.random(n) < 0.8
        ds, nlc = (d if d else nl), (nl if d else 0)
        st = make_mlp(rng, al + 2 * (ds + nlc), list(hidden) + [ds], act, gain=0.6, bn_random=True)
        ou = make_mlp(rng, ds + nlc, [2], 'softmax', bn_random=True)
        s0 = (0.1 * rng.standard_normal((n, ds))).astype(np.float32) if d else np.zeros((0, 0), np.float32)
        out[f'{name}/arcs'], out[f'{name}/nodes'], out[f'{name}/set_mask'], out[f'{name}/s0'] = arcs, nodes, g['set_mask'], s0
        out[f'{name}/cfg'] = np.array([d, nl, al, 25, len(hidden)], np.int64)
        out[f'{name}/mode'], out[f'{name}/act'] = np.array(mode), np.array(act)
        for i, w in enumerate(st['weights']): out[f'{name}/st{i}'] = w
        for i, w in enumerate(ou['weights']): out[f'{name}/ou{i}'] = w
        for dt, tag in ((torch.float64, 'f64'), (torch.float32, 'f32')):
            k, s, o = t_loop(g, st, ou, d, 25, 0.01, s0 if d else None, dt)
            out[f'{name}/{tag}/k'], out[f'{name}/{tag}/state'], out[f'{name}/{tag}/out'] = np.array(k), s, o
        print(name, 'k =', int(out[f'{name}/f64/k']), int(out[f'{name}/f32/k']))
    # single ops on saturating inputs
    rng = np.random.default_rng(77)
    x = (3 * rng.standard_normal((64, 11))).astype(np.float32)
    x[0], x[1], x[2] = 0, 40, -40
    out['ops/x'] = x
    for act in ACT:
        out[f'ops/{act}'] = ACT[act](torch.tensor(x, dtype=torch.float64)).numpy()
    bn = [rng.uniform(0.5, 1.5, 11), rng.uniform(-0.2, 0.2, 11), rng.uniform(-0.2, 0.2, 11), rng.uniform(0.5, 1.5, 11)]
    out['ops/bn_params'] = np.stack(bn)
    out['ops/bn'] = F.batch_norm(torch.tensor(x, dtype=torch.float64), torch.tensor(bn[2]), torch.tensor(bn[3]), torch.tensor(bn[0]), torch.tensor(bn[1]),
                                 training=False, eps=1e-3).numpy()
    np.savez_compressed(os.path.join(HERE, 'torch_crosscheck.npz'), **out)
    print('wrote', os.path.join(HERE, 'torch_crosscheck.npz'), f'torch {torch.__version__}')


if __name__ == '__main__':
    main()
