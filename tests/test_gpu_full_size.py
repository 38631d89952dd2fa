"""Oracle parity at BASELINE size (configs[2] / configs[4]: 1,000,000 nodes, ~10,000,000 arcs, state_dim 64) and the regression
test of the round-2 intermittent parity failure.

The smaller parity suites (test_gpu_parity.py) compare every case with the C oracle at sizes of 10^2 .. 10^5 nodes; here the SAME
check runs once on the bench's own graph, weights and initial state (bench.py, seed 20261003): all 64 M state values, the outputs
and k of a 3-body GNN.Loop (reference GNN/GNN.py:251-280) bit for bit on the exact path, within BASELINE.json's 1e-5 on the default
path; then one LGNN layer > 0 (reference GNN/LGNN.py:263-290: labels widened by the previous layer's output, 139 -> 128 -> 128 -> 64).
"""
import os
import sys

import numpy as np
import pytest

from oracle import c_oracle as corc
from oracle import gnn_oracle as orc

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def c3():
    """bench.py's workload, built exactly as bench.py builds it (same generator, seed and draw order)."""
    sys.path.insert(0, ROOT)
    import bench
    from GNN import _engine as e, GNN_utils as utils
    d, nl, al, t = 64, 3, 1, 2
    s = utils.syntheticGraph(1_000_000, 10.0, nl, al, t, seed=20261003)
    n = s['n_nodes']
    rng = np.random.default_rng(20261003)
    st = bench.make_net(rng, al + 2 * (nl + d), [128, 128, d], 'selu')
    ou = bench.make_net(rng, nl + d, [t], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    og = bench.oracle_graph(s)
    # the oracle's answer for 3 bodies (threshold 0): computed once, shared by the tests below
    kc, sc, oc = corc.loop_node(og, st, ou, d, 3, 0.0, s0)
    assert kc == 3 and sc.shape == (n, d) and oc.shape == (n, t)
    yield dict(e=e, s=s, n=n, d=d, st=st, ou=ou, s0=s0, graph=graph, mst=mst, mou=mou, og=og, kc=kc, sc=sc, oc=oc, rng=rng)
    graph.close()


def test_c3_full_size_bit_exact_vs_c_oracle(c3):
    """configs[2] at full size: impl 1 (f32 MFMA) and impl 0 (one kernel per op) equal the C oracle on every one of the 64 M state
    values, on the outputs and on k; the default path (impl 2) is within 1e-5 x max|state| of it (BASELINE.json: 1e-5 fp32)."""
    e = c3['e']
    for impl in (1, 0, 2):
        loop = e.Loop(c3['graph'], c3['mst'], c3['mou'], c3['d'], 3, 0.0)
        assert loop.set_impl(impl) == impl
        loop.set_state0(c3['s0'])
        k = loop.run()
        state, out = loop.state(), loop.output()
        loop.close()
        assert k == c3['kc']
        if impl < 2:
            assert np.array_equal(state, c3['sc']), f'impl {impl}: {int(np.sum(state != c3["sc"]))} of {state.size} state values differ'
            assert np.array_equal(out, c3['oc'])
        else:
            scale = max(1.0, float(np.max(np.abs(c3['sc']))))
            ds_, do_ = float(np.max(np.abs(state - c3['sc']))), float(np.max(np.abs(out - c3['oc'])))
            assert ds_ <= 1e-5 * scale and do_ <= 1e-5, (ds_, do_, scale)


def test_c3_headline_depth_parity(c3):
    """The headline's own size AND depth (bench.py: 1,000,000 nodes, 30 bodies, threshold 0; reference loop GNN/GNN.py:271): the exact
    path (impl 1) equals the C oracle on all 64 M state values, on the outputs and on k after 30 bodies; the default path (impl 2) stops
    at the same k and is no further from the float64 shadow (oracle/gnn_oracle_f64.c) than 1.5 x the exact float32 chain is - with
    random-init weights the state map is expansive, so after 30 bodies NO float32 order is within 1e-5 of float64 (measured and asserted
    below: the oracle's own chain is 1e-5 .. 1e-4 away); what can be required of the default path at this depth is that it adds nothing
    to the float32 noise, and that is what is asserted."""
    e, d = c3['e'], c3['d']
    bodies = 30
    kc, sc, oc = corc.loop_node(c3['og'], c3['st'], c3['ou'], d, bodies, 0.0, c3['s0'])
    k64, s64, o64 = corc.loop_node_f64(c3['og'], c3['st'], c3['ou'], d, bodies, 0.0, c3['s0'])
    assert kc == k64 == bodies
    res = {}
    for impl in (1, 2):
        loop = e.Loop(c3['graph'], c3['mst'], c3['mou'], d, bodies, 0.0)
        assert loop.set_impl(impl) == impl
        loop.set_state0(c3['s0'])
        k = loop.run()
        res[impl] = (k, loop.state(), loop.output(), loop.gate_info())
        loop.close()
    k1, s1, o1, _ = res[1]
    assert k1 == kc and np.array_equal(s1, sc) and np.array_equal(o1, oc), f'impl 1: {int(np.sum(s1 != sc))} of {s1.size} state values differ after {bodies} bodies'
    k2, s2, o2, (repeated, _) = res[2]
    assert k2 == kc and not repeated                    # threshold 0 with moving states: every gate has a robust mover
    e1s, e2s = float(np.max(np.abs(s1 - s64))), float(np.max(np.abs(s2 - s64)))
    e1o, e2o = float(np.max(np.abs(o1 - o64))), float(np.max(np.abs(o2 - o64)))
    assert e2s <= 1.5 * e1s and e2o <= 1.5 * e1o + 1e-7, (e1s, e2s, e1o, e2o)
    assert e1s < 1e-3 and e2s < 1e-3                    # (sanity: float32 noise, not a wrong result; max |state| is about 3)
    del s1, s2, s64, sc


def test_c5_layer_full_size_bit_exact_vs_c_oracle(c3):
    """One layer > 0 of configs[4] at full size: the labels of the ORIGINAL graph widened by the previous layer's output
    (get_state=False, get_output=True: NL' = 5, reference LGNN.py:227-260, starter.py:78-79), relabelled on the device, then a
    3-body Loop with net_state 139 -> 128 -> 128 -> 64 / net_output 69 -> 2."""
    import bench
    e, n, d = c3['e'], c3['n'], c3['d']
    rng = np.random.default_rng(4)
    ins, ls = orc.get_inout_dims('state', 3, 1, 2, 'n', d, [128, 128], layer=1, get_state=False, get_output=True)
    ino, lo = orc.get_inout_dims('output', 3, 1, 2, 'n', d, None, layer=1, get_state=False, get_output=True)
    assert (ins, ls, ino, lo) == (139, [128, 128, 64], 69, [2])
    st1, ou1 = bench.make_net(rng, ins, ls, 'selu'), bench.make_net(rng, ino, lo, 'softmax')
    s01 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    # layer 0 on the device (exact path), relabel, layer 1
    loop0 = e.Loop(c3['graph'], c3['mst'], c3['mou'], d, 3, 0.0)
    loop0.set_impl(1)
    loop0.set_state0(c3['s0'])
    assert loop0.run() == 3
    derived = c3['graph'].derive(2)
    derived.update_labels(c3['graph'], loop0, False, True)
    g1 = orc.update_graph(c3['og'], c3['sc'], c3['oc'], False, True)
    assert np.array_equal(derived.nodes(), g1['nodes'])
    m1s, m1o = e.Mlp(st1['weights'], st1['activations'], True), e.Mlp(ou1['weights'], ou1['activations'], True)
    k1c, s1c, o1c = corc.loop_node(g1, st1, ou1, d, 3, 0.0, s01)
    for impl in (1, 2):
        loop1 = e.Loop(derived, m1s, m1o, d, 3, 0.0)
        assert loop1.set_impl(impl) == impl
        loop1.set_state0(s01)
        k1 = loop1.run()
        s1, o1 = loop1.state(), loop1.output()
        loop1.close()
        assert k1 == k1c
        if impl == 1:
            assert np.array_equal(s1, s1c) and np.array_equal(o1, o1c)
        else:
            scale = max(1.0, float(np.max(np.abs(s1c))))
            assert float(np.max(np.abs(s1 - s1c))) <= 1e-5 * scale and float(np.max(np.abs(o1 - o1c))) <= 1e-5
    loop0.close()
    derived.close()


def _relabel_rounds(e, graph, loop, expect, extra, rounds):
    """derive + relabel + read back, `rounds` times over; returns [(round, values that differ, of which zeroed)]"""
    wiped = []
    for rnd in range(rounds):
        derived = graph.derive(extra)
        derived.update_labels(graph, loop, True, True)
        got = derived.nodes()
        derived.close()
        if not np.array_equal(got, expect):
            wiped.append((rnd, int(np.sum(got != expect)), int(np.sum((got == 0) & (expect != 0)))))
    return wiped


def test_relabelling_is_ordered_behind_the_creation_fill(c3):
    """Regression test of the round-2 intermittent LGNN parity failure ("Initialisation order").

    Diagnosed ordering: gnn_graph_derive zero-filled the new label array with hipMemset - queued on the NULL stream, returning before
    the fill had run - and gnn_graph_update_labels wrote the labels at once on the loop's hipStreamNonBlocking stream, which the null
    stream does not order: the fill could land after (part of) the relabelling and wipe it (tools/memset_race_probe.hip: 2 % of the
    trials at the failing test's size, 45 - 85 % at 256 MB back to back; the loop_create -> run sequence of the state buffers: 0 of
    820).  It is a race of two queues, so the test repeats the sequence derive -> relabel -> read back many times in ONE process at
    three sizes: the failing test's (4,133 nodes), a mid size and BASELINE size (276 MB of labels).  With the fill ordered before the
    relabelling by the graph's ready event no round can differ; run against the diagnostic build with GNN_LEGACY_NULL_MEMSET=1 (the
    unordered fill) to see the test catch the old behaviour (profiles/r03_memset_race.txt)."""
    from GNN import GNN_utils as utils
    e, d = c3['e'], c3['d']
    report = {}
    for n_small, rounds in ((4133, 1500), (65_000, 300)):
        s = utils.syntheticGraph(n_small, 10.0, 3, 1, 2, seed=5)
        n = s['n_nodes']
        rng = np.random.default_rng(n)
        s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
        graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
        loop = e.Loop(graph, c3['mst'], c3['mou'], d, 1, 0.0)
        loop.set_impl(1)
        loop.set_state0(s0)
        assert loop.run() == 1
        expect = np.concatenate([s['nodes'], loop.state(), loop.output()], axis=1)      # LGNN.py:241-259 with all-true masks
        report[n] = _relabel_rounds(e, graph, loop, expect, d + 2, rounds)
        loop.close()
        graph.close()
    loop = e.Loop(c3['graph'], c3['mst'], c3['mou'], d, 3, 0.0)
    loop.set_impl(1)
    loop.set_state0(c3['s0'])
    assert loop.run() == 3
    expect = orc.update_graph(c3['og'], c3['sc'], c3['oc'], True, True)['nodes']
    assert expect.shape == (c3['n'], 3 + d + 2)
    report[c3['n']] = _relabel_rounds(e, c3['graph'], loop, expect, d + 2, 6)
    loop.close()
    bad = {n: (len(w), w[:5]) for n, w in report.items() if w}
    assert not bad, f'relabelled labels differ from [nodes | state | output]: nodes -> (rounds affected, first (round, values, of which zeroed)): {bad}'
