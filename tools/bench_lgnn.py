"""BASELINE configs[4]: 5-layer LGNN (get_output=True) on the 1M-node synthetic graph, state_dim 64, max_iter 30, threshold 0.
Layers 1-4 see node labels widened by the previous layer's 2 outputs (NL' = 5: net_state 139->128->128->64, net_output 69->2).
Everything stays on the device (gnn_graph_derive / gnn_graph_update_labels between layers).  python tools/bench_lgnn.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd')):
    sys.path.insert(0, p)
from GNN import _engine as e, GNN_utils as utils      # noqa: E402
from bench import make_net                             # noqa: E402


def main():
    layers, d, nl, al, t, max_it = 5, 64, 3, 1, 2, 30
    s = utils.syntheticGraph(1_000_000, 10.0, nl, al, t, seed=20261003)
    n = s['n_nodes']
    rng = np.random.default_rng(7)
    base = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    derived = base.derive(t)
    loops = []
    for layer in range(layers):
        nll = nl + (t if layer else 0)
        st = make_net(rng, al + 2 * (nll + d), [128, 128, d], 'selu')
        ou = make_net(rng, nll + d, [t], 'softmax')
        g = base if layer == 0 else derived
        loop = e.Loop(g, e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, max_it, 0.0)
        loop.set_state0((0.1 * rng.standard_normal((n, d))).astype(np.float32))
        loops.append(loop)

    def run_stack():
        ks = 0.0
        for layer, loop in enumerate(loops):
            ks += loop.run()
            if layer < layers - 1:
                derived.update_labels(base, loop, False, True)
        return ks

    run_stack()
    reps = 3
    t0 = time.perf_counter()
    ks = sum(run_stack() for _ in range(reps))
    dt = time.perf_counter() - t0
    print(f'LGNN x{layers}: {1e3 * dt / reps:.1f} ms per LGNN.Loop, {ks / reps:.0f} iterations, {n * ks / dt:.3e} node-state-updates/s')


if __name__ == '__main__':
    main()
