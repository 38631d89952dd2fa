"""k_fused_pair (wave pair per tile) against k_fused (one wave per tile) on the default path: states, outputs and k must be identical bit for bit
(the two forms evaluate the same arithmetic per node).  GPU box only.  python tools/check_pair.py [big]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)
from GNN import _engine as e, GNN_utils as utils      # noqa: E402
sys.path.insert(0, os.path.join(ROOT, 'tools'))
from gate_study import make_net                        # noqa: E402

e.require_device(0)
bad = 0
cases = [(4096, (128, 128), 'selu', 3, 6, 0.0), (333, (128, 128), 'tanh', 3, 5, 0.0), (1000, (128,), 'selu', 3, 4, 0.0), (40_000, (128, 128), 'relu', 5, 8, 0.001),
         (100_003, (128, 128), 'selu', 3, 12, 0.01), (65_536, (128,), 'sigmoid', 3, 3, 0.0)]
if len(sys.argv) > 1:
    cases.append((1_000_000, (128, 128), 'selu', 3, 30, 0.0))
for n, hidden, act, nl, max_it, thr in cases:
    s = utils.syntheticGraph(n, 10.0, nl, 1, 2, seed=n)
    rng = np.random.default_rng(n)
    st = make_net(rng, 1 + 2 * (nl + 64), list(hidden) + [64], act, 0.6 if thr else 1.0)
    ou = make_net(rng, nl + 64, [2], 'softmax', 1.0)
    s0 = (0.1 * rng.standard_normal((n, 64))).astype(np.float32)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    res = {}
    for form in (1, 2):
        lp = e.Loop(graph, mst, mou, 64, max_it, thr)
        assert lp.set_impl(2) == 2
        used = lp.set_tile_form(form)
        lp.set_state0(s0)
        k = lp.run()
        t = time.perf_counter()
        k = lp.run()
        dt = time.perf_counter() - t
        res[form] = (used, k, lp.state(), lp.output(), dt)
        lp.close()
    (u1, k1, s1, o1, t1), (u2, k2, s2, o2, t2) = res[1], res[2]
    same = k1 == k2 and np.array_equal(s1, s2) and np.array_equal(o1, o2)
    bad += not same or u2 != 2
    print(f'N={n} hidden={hidden} act={act} NL={nl} max_it={max_it} thr={thr}: forms used {u1}/{u2}, k {k1}/{k2}, identical {same}, '
          f'max|ds| {float(np.max(np.abs(s1 - s2))):.3e}, NaNs {int(np.isnan(s2).sum())}, ms per Loop {1e3 * t1:.3f} / {1e3 * t2:.3f}', flush=True)
    graph.close()
sys.exit(1 if bad else 0)
