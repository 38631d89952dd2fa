import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
import test_gpu_parity as T
from oracle import c_oracle as corc
e = T._engine()
def run(n, act, hidden=(32,), d=7, nl=7, al=2, it=1, show=False):
    rng = np.random.default_rng(5)
    g, st, ou, s0 = T._case(rng, n=n, d=d, nl=nl, al=al, hidden=hidden, act=act, deg=4)
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    kc, sc, oc = corc.loop_node(g, st, ou, d, it, 0.0, s0)
    loop = e.Loop(T._device_graph(g), mst, mou, d, it, 0.0)
    loop.set_impl(1); p = loop.set_persistent(True)
    if d: loop.set_state0(s0)
    k = loop.run(); s = loop.state()
    bad = np.nonzero(np.any(s != sc, axis=1))[0]
    print(f'n={n} act={act} hidden={hidden} d={d}: persistent={p} bad rows {len(bad)} max|diff| {np.max(np.abs(s - sc)):.3g} max|s| {np.max(np.abs(sc)):.3g}')
    if show and len(bad):
        r = bad[0]
        print('  row', r, 'got ', s[r]); print('  row', r, 'want', sc[r])
        # what would the state be with a zero aggregate, or with the aggregate of another buffer?
    loop.close()
for n in (6000, 4096, 2048, 2047, 2032, 1999):
    run(n, 'linear', show=(n == 4096))
run(4096, 'linear', hidden=())
run(4096, 'linear', hidden=(16,))
run(4096, 'linear', hidden=(17,))
