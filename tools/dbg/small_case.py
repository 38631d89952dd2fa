"""debug: one persistent-loop case against the C oracle, with variations"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
import test_gpu_parity as T
from oracle import c_oracle as corc
e = T._engine()
def run(seed, n, d, nl, al, hidden, act, deg, max_it, thr):
    rng = np.random.default_rng(seed)
    g, st, ou, s0 = T._case(rng, n=n, d=d, nl=nl, al=al, hidden=hidden, act=act, deg=deg)
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    kc, sc, oc = corc.loop_node(g, st, ou, d, max_it, thr, s0)
    loop = e.Loop(T._device_graph(g), mst, mou, d, max_it, thr)
    loop.set_impl(1)
    p = loop.set_persistent(True)
    if d: loop.set_state0(s0)
    k = loop.run()
    s = loop.state()
    bad = np.nonzero(np.any(s != sc, axis=1))[0]
    indeg = np.diff(g['adjT'][0])
    print(f'n={n} d={d} nl={nl} al={al} hidden={hidden} act={act} deg={deg} it={max_it} thr={thr}: persistent={p} k={k}/{kc} bad rows {len(bad)} first {bad[:8]} maxdeg {indeg.max()} tiles-with-bad {sorted(set((bad // 16).tolist()))[:10]}')
    loop.close()
for n in (4096, 4095, 2000, 300):
    for deg in (1, 4, 11):
        run(5, n, 7, 7, 2, (32,), 'selu', deg, 3, 0.0)
run(5, 4096, 8, 7, 2, (32,), 'selu', 11, 3, 0.0)
run(5, 4096, 7, 7, 2, (), 'selu', 11, 3, 0.0)
run(5, 4096, 7, 7, 2, (32,), 'selu', 11, 1, 0.0)
print('--- linear')
for n in (4096, 4080, 2048, 4095):
    run(5, n, 7, 7, 2, (32,), 'linear', 4, 3, 0.0)
run(5, 4096, 7, 7, 2, (32,), 'relu', 4, 3, 0.0)
run(5, 4096, 7, 7, 2, (32,), 'linear', 4, 1, 0.0)
run(5, 4096, 7, 7, 2, (32,), 'linear', 4, 2, 0.0)
run(5, 4096, 0, 7, 2, (32,), 'linear', 4, 3, 0.0)
run(5, 4096, 12, 2, 3, (), 'linear', 4, 6, 0.01)
run(5, 4096, 12, 2, 3, (), 'tanh', 4, 6, 0.01)
