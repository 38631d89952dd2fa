"""debug: the wide-state case of tests/test_gpu_parity.py::test_wide_states_and_fallback repeated in one process, alternating with a sharded
(loopback) run like the tests that ran before it when it failed once; prints every deviation with the rows it affects"""
import sys, os, gc
sys.path.insert(0, 'tests'); sys.path.insert(0, '.'); sys.path.insert(0, 'gnn_tf_2.x_amd')
import numpy as np
import test_gpu_parity as T
import test_gpu_sharded as S
from oracle import c_oracle as corc
e = T._engine()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 100
d, hidden = 60, (128,)
rng = np.random.default_rng(300 + d)
g, st, ou, s0 = T._case(rng, n=333, d=d, nl=3, al=2, hidden=hidden, act='tanh', gain=0.5)
kc, sc, oc = corc.loop_node(g, st, ou, d, 10, 0.01, s0)
gs, sts, ous, s0s = S._case(7, 1000, 8, hidden=(16,))
ip, asrc, aw, _, _ = S._csr_parts(gs)
bad = 0
for rep in range(reps):
    if rep % 2 == 0:      # a sliced loopback run, objects left to the garbage collector
        comms, graphs, loops, ranges = S._sharded_loops(e, gs, sts, ous, 8, 30, 0.01, s0s, 4, 1)
        for gr, lp in zip(graphs, loops):
            gr.set_full_adjacency(1000, ip, asrc, aw)
            lp.set_slice_exchange(1)
        e.Loop.run_group(loops)
    loop = e.Loop(T._device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 10, 0.01)
    loop.set_impl(1); loop.set_state0(s0)
    k1 = loop.run(); s1 = loop.state()
    loop.set_impl(2)
    k2 = loop.run(); s2 = loop.state()
    e1, e2 = float(np.max(np.abs(s1 - sc))), float(np.max(np.abs(s2 - sc)))
    if k1 != kc or e1 != 0 or k2 != kc or not e2 < 1e-6:
        bad += 1
        rows = np.nonzero(np.max(np.abs(s2 - sc), axis=1) > 1e-6)[0]
        print(f'rep {rep}: k1 {k1} e1 {e1} k2 {k2} e2 {e2} nan {int(np.isnan(s2).sum())} rows {rows[:40]} (n={len(rows)})', flush=True)
print(f'{bad} deviations in {reps} repetitions')
