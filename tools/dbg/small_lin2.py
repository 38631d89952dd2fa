import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
import test_gpu_parity as T
from oracle import c_oracle as corc, gnn_oracle as orc
e = T._engine()
rng = np.random.default_rng(5)
n, d, nl, al = 300, 7, 7, 2
for act in ('linear', 'relu'):
    rng = np.random.default_rng(5)
    g, st, ou, s0 = T._case(rng, n=n, d=d, nl=nl, al=al, hidden=(), act=act, deg=4)
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    kc, sc, oc = corc.loop_node(g, st, ou, d, 1, 0.0, s0)
    loop = e.Loop(T._device_graph(g), mst, mou, d, 1, 0.0)
    loop.set_impl(1); loop.set_persistent(True); loop.set_state0(s0)
    k = loop.run(); s = loop.state()
    print(act, 'mismatches per feature', (s != sc).sum(axis=0), 'of', n)
    W, b = st['weights'][0], st['weights'][1]
    gam, bet, mu, var = st['weights'][2:6]
    r = int(np.nonzero(np.any(s != sc, axis=1))[0][0]) if np.any(s != sc) else 0
    # pre-BN value implied by got / want for feature 3
    sc_, sh_ = gam / np.sqrt(var + 1e-3), bet - mu * gam / np.sqrt(var + 1e-3)
    print('  row', r, 'pre-BN got', (s[r] - sh_) / sc_, '\n        pre-BN want', (sc[r] - sh_) / sc_, '\n  bias', b)
    loop.close()
