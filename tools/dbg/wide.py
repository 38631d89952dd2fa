import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
import test_gpu_parity as T
from oracle import c_oracle as corc
e = T._engine()
for d, hidden in ((60, (128,)), (68, (96,))):
    rng = np.random.default_rng(300 + d)
    g, st, ou, s0 = T._case(rng, n=333, d=d, nl=3, al=2, hidden=hidden, act='tanh', gain=0.5)
    kc, sc, oc = corc.loop_node(g, st, ou, d, 10, 0.01, s0)
    loop = e.Loop(T._device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 10, 0.01)
    for impl in (1, 2, 2):
        print('impl', loop.set_impl(impl))
        loop.set_state0(s0)
        k = loop.run()
        print(d, 'k', k, kc, 'err', np.max(np.abs(loop.state() - sc)), 'scale', np.max(np.abs(sc)), 'nan', np.isnan(loop.state()).sum())
