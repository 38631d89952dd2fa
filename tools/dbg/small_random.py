"""debug: the sequence of tests/test_gpu_parity.py::test_persistent_small_graph_loop_random_shapes, reporting every case"""
import sys, os
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
import test_gpu_parity as T
from oracle import c_oracle as corc
e = T._engine()
rng = np.random.default_rng(20261004)
acts = ['selu', 'tanh', 'relu', 'sigmoid', 'elu', 'linear']
only = int(sys.argv[1]) if len(sys.argv) > 1 else -1
for case in range(30):
    d = int(rng.choice([0, 0, 1, 3, 4, 7, 12, 16, 17, 24, 31, 32]))
    nl = int(rng.integers(1, 9)) if d else int(rng.integers(1, 33))
    al = int(rng.integers(1, 5))
    hidden = tuple(int(x) for x in rng.integers(1, 33, size=int(rng.integers(0, 3))))
    n = int(rng.choice([17, 33, 100, 640, 1999, 4096, 4097, 6000]))
    act = acts[case % len(acts)]
    g, st, ou, s0 = T._case(rng, n=n, d=d, nl=nl, al=al, hidden=hidden, act=act, deg=int(rng.choice([1, 4, 11])))
    max_it, thr = int(rng.integers(1, 12)), float(rng.choice([0.0, 0.01, 0.1]))
    if only >= 0 and case != only: continue
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    kc, sc, oc = corc.loop_node(g, st, ou, d, max_it, thr, s0)
    for rep in range(2):
        loop = e.Loop(T._device_graph(g), mst, mou, d, max_it, thr)
        loop.set_impl(1)
        p = loop.set_persistent(bool(1 - rep))
        if d: loop.set_state0(s0)
        k = loop.run()
        s, o = loop.state(), loop.output()
        bad = np.nonzero(np.any(s != sc, axis=1))[0]
        print(f'case {case} persistent={p}: n={n} d={d} nl={nl} al={al} hidden={hidden} act={act} it={max_it} thr={thr} maxdeg={np.diff(g["adjT"][0]).max()}: k={k}/{kc} bad rows {len(bad)} first {bad[:6]} max|diff| {np.max(np.abs(s - sc)):.3g} out_equal {np.array_equal(o, oc)}', flush=True)
        loop.close()
