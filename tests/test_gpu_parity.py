"""Parity tests proper: the HIP path, called through the C ABI (libgnn_hip.so), against the oracle on the same seeded
inputs.  Bar: the C oracle pins the floating-point evaluation order, so states / outputs / iteration counts must be
BIT-IDENTICAL to it (np.array_equal); against the float64 shadow the tolerance is 1e-5 absolute (BASELINE.json
north_star) on contractive state maps."""
import os

import numpy as np
import pytest

from oracle import c_oracle as corc
from oracle import gnn_oracle as orc
from util import make_mlp, random_arcs

pytestmark = pytest.mark.gpu

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'graph_fixtures.npz'))


def _engine():
    from GNN import _engine
    return _engine


def _device_graph(g):
    e = _engine()
    mask = np.logical_and(g['set_mask'], g['output_mask'])
    arc_labels = np.asarray(g['arcs'], np.float32)[:, 2:]
    return e.Graph(g['nodes'].shape[0], g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], arc_labels[g['arcT'][1]],
                   g['nodes'], mask)


def _run_hip(g, st, ou, d, max_it, thr, s0, impl=1):
    e = _engine()
    loop = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], st['batch_normalization']),
                  e.Mlp(ou['weights'], ou['activations'], ou['batch_normalization']), d, max_it, thr)
    loop.set_impl(impl)
    if d:
        loop.set_state0(s0)
    k = loop.run()
    return k, loop.state(), loop.output()


def _case(rng, n=300, d=8, nl=3, al=2, hidden=(16,), act='selu', gain=0.6, sort=True, mode='average', deg=3):
    arcs = random_arcs(rng, n, deg * n, al, sort=sort)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    g = orc.make_graph_dict(arcs, nodes, mode)
    ds, nls = (d if d else nl), (nl if d else 0)
    st = make_mlp(rng, al + 2 * (ds + nls), list(hidden) + [ds], act, gain=gain, bn_random=True)
    ou = make_mlp(rng, ds + nls, [2], 'softmax', bn_random=True)
    s0 = (0.1 * rng.standard_normal((n, ds))).astype(np.float32) if d else None
    return g, st, ou, s0


@pytest.mark.parametrize('act', ['linear', 'relu', 'selu', 'elu', 'tanh', 'sigmoid', 'softmax'])
@pytest.mark.parametrize('bn', [True, False])
def test_mlp_forward_bit_exact(act, bn):
    rng = np.random.default_rng(sum(map(ord, act)) + int(bn))
    net = make_mlp(rng, 37, [65, 129, 11], act, batch_normalization=bn, bn_random=True)
    x = (3 * rng.standard_normal((517, 37))).astype(np.float32)
    x[0, :] = 0
    x[1, :] = 50      # saturating inputs
    x[2, :] = -50
    y = _engine().Mlp(net['weights'], net['activations'], bn).forward(x)
    assert np.array_equal(y, corc.mlp_forward(x, net['weights'], net['activations'], bn))
    assert np.max(np.abs(y - orc.mlp_forward(x, net['weights'], net['activations'], bn, np.float64))) < 2e-4


@pytest.mark.parametrize('impl', [0, 1])
@pytest.mark.parametrize('d,nl,al,hidden,act,mode,sort', [
    (0, 3, 1, (), 'selu', 'average', True),            # BASELINE config 1 shape: in 7 -> [3]
    (0, 14, 3, (32, 32), 'selu', 'average', True),     # MUTAG shape: 31 -> [32, 32, 14]
    (8, 3, 2, (16,), 'tanh', 'sum', False),
    (5, 2, 1, (7, 9), 'relu', 'normalized', False),
    (64, 3, 1, (128, 128), 'selu', 'average', True),   # BASELINE config 3 shape: 135 -> [128, 128, 64]
    (64, 5, 1, (128, 128), 'selu', 'average', True),   # LGNN layer > 0 shape: 139 -> [128, 128, 64]
    (32, 4, 0, (64,), 'sigmoid', 'average', True),     # no arc labels
])
def test_loop_bit_exact_vs_c_oracle(impl, d, nl, al, hidden, act, mode, sort):
    rng = np.random.default_rng(1000 + 7 * d + nl)
    g, st, ou, s0 = _case(rng, n=700, d=d, nl=nl, al=al, hidden=hidden, act=act, sort=sort, mode=mode)
    g['set_mask'] = rng.random(700) < 0.8
    g['output_mask'] = rng.random(700) < 0.7
    kc, sc, oc = corc.loop_node(g, st, ou, d, 30, 0.01, s0)
    k, s, o = _run_hip(g, st, ou, d, 30, 0.01, s0, impl)
    assert k == kc
    assert np.array_equal(s, sc)
    assert o.shape == oc.shape and np.array_equal(o, oc)
    k64, s64, o64 = orc.loop_node(g, st, ou, d, 30, 0.01, s0, np.float64)
    assert k == k64
    assert np.max(np.abs(s - s64)) < 1e-5 and np.max(np.abs(o - o64)) < 1e-5


@pytest.mark.parametrize('d,nl,al,hidden,act,mode,sort', [
    (0, 3, 1, (), 'selu', 'average', True),
    (0, 14, 3, (32, 32), 'selu', 'average', True),
    (8, 3, 2, (16,), 'tanh', 'sum', False),
    (5, 2, 1, (7, 9), 'relu', 'normalized', False),
    (64, 3, 1, (128, 128), 'selu', 'average', True),   # BASELINE config 3 shape
    (64, 5, 1, (128, 128), 'selu', 'average', True),
    (32, 4, 0, (64,), 'sigmoid', 'average', True),
    (16, 3, 1, (40, 72), 'elu', 'average', True),      # K = 39 -> 3 chunks (odd), hidden widths that are not tile multiples
])
def test_loop_split_arithmetic_within_tolerance(d, nl, al, hidden, act, mode, sort):
    """impl 2: the dense layers run on the bf16 MFMA with every fp32 operand cut into three exact bf16 pieces (six piece
    products per term, fp32 accumulate).  Same k; states / outputs within 1e-5 of the float64 oracle (BASELINE north_star
    tolerance) and within fp32 rounding noise of the exact path."""
    rng = np.random.default_rng(1000 + 7 * d + nl)
    g, st, ou, s0 = _case(rng, n=700, d=d, nl=nl, al=al, hidden=hidden, act=act, sort=sort, mode=mode)
    g['set_mask'] = rng.random(700) < 0.8
    g['output_mask'] = rng.random(700) < 0.7
    k, s, o = _run_hip(g, st, ou, d, 30, 0.01, s0, 2)
    k1, s1, o1 = _run_hip(g, st, ou, d, 30, 0.01, s0, 1)
    k64, s64, o64 = orc.loop_node(g, st, ou, d, 30, 0.01, s0, np.float64)
    assert k == k64 == k1
    assert np.max(np.abs(s - s64)) < 1e-5 and np.max(np.abs(o - o64)) < 1e-5
    assert np.max(np.abs(s - s1)) < 2e-6 * max(1.0, np.max(np.abs(s1)))


@pytest.mark.parametrize('impl', [0, 1])
def test_loop_edge_cases(impl):
    rng = np.random.default_rng(5)
    g, st, ou, s0 = _case(rng, n=257, d=4, gain=1.0)
    zero = dict(st, weights=[np.zeros_like(w) for w in st['weights'][:-4]] + [np.ones(4, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32), np.ones(4, np.float32)])
    k, s, _ = _run_hip(g, zero, ou, 4, 30, 0.01, s0, impl)
    assert k == 2 and np.all(s == 0)                               # zero net_state: converges at k == 2
    k, s, o = _run_hip(g, st, ou, 4, 9, 0.0, s0, impl)
    kc, sc, oc = corc.loop_node(g, st, ou, 4, 9, 0.0, s0)
    assert k == kc == 9 and np.array_equal(s, sc) and np.array_equal(o, oc)   # threshold 0 -> max_iteration
    k, s, o = _run_hip(g, st, ou, 4, 0, 0.01, s0, impl)
    assert k == 0 and np.array_equal(s, s0)                        # max_iteration 0: body never runs
    assert np.array_equal(o, corc.loop_node(g, st, ou, 4, 0, 0.01, s0)[2])
    # state == ones at entry (D == 0 with all-ones labels): first condition is false
    arcs = np.array([[0, 1, .5], [1, 0, .5]], dtype=np.float32)
    g1 = orc.make_graph_dict(arcs, np.ones((3, 2), np.float32))    # node 2 is isolated
    st1, ou1 = make_mlp(rng, 1 + 2 * 2, [2], 'linear'), make_mlp(rng, 2, [2], 'softmax')
    assert _run_hip(g1, st1, ou1, 0, 5, 0.01, None, impl)[0] == 0
    # isolated node + empty mask + no arcs at all
    g2 = orc.make_graph_dict(arcs, rng.random((3, 2)).astype(np.float32))
    g2['set_mask'] = np.zeros(3, bool)
    k, s, o = _run_hip(g2, st1, ou1, 0, 5, 0.01, None, impl)
    kc, sc, oc = corc.loop_node(g2, st1, ou1, 0, 5, 0.01, None, want_out=False)
    assert k == kc and np.array_equal(s, sc) and o.shape == (0, 2)
    g3 = orc.make_graph_dict(np.zeros((0, 3), np.float32), rng.random((5, 2)).astype(np.float32))
    k, s, o = _run_hip(g3, st1, ou1, 0, 4, 0.01, None, impl)
    kc, sc, oc = corc.loop_node(g3, st1, ou1, 0, 4, 0.01, None)
    assert k == kc and np.array_equal(s, sc) and np.array_equal(o, oc)


def test_loop_edge_cases_split_arithmetic():
    """The same edge cases on the default path (impl 2): iteration counts identical, values within fp32 rounding noise."""
    rng = np.random.default_rng(5)
    g, st, ou, s0 = _case(rng, n=257, d=4, gain=0.7)              # partial last tile (257 = 8 x 32 + 1)
    zero = dict(st, weights=[np.zeros_like(w) for w in st['weights'][:-4]] + [np.ones(4, np.float32), np.zeros(4, np.float32), np.zeros(4, np.float32), np.ones(4, np.float32)])
    k, s, _ = _run_hip(g, zero, ou, 4, 30, 0.01, s0, 2)
    assert k == 2 and np.all(s == 0)
    for max_it, thr in ((9, 0.0), (0, 0.01), (30, 0.05)):
        k, s, o = _run_hip(g, st, ou, 4, max_it, thr, s0, 2)
        kc, sc, oc = corc.loop_node(g, st, ou, 4, max_it, thr, s0)
        assert k == kc and np.max(np.abs(s - sc)) < 2e-6 and np.max(np.abs(o - oc)) < 2e-6, (max_it, thr)
    arcs = np.array([[0, 1, .5], [1, 0, .5]], dtype=np.float32)
    st1, ou1 = make_mlp(rng, 1 + 2 * 2, [2], 'linear'), make_mlp(rng, 2, [2], 'softmax')
    g1 = orc.make_graph_dict(arcs, np.ones((3, 2), np.float32))
    assert _run_hip(g1, st1, ou1, 0, 5, 0.01, None, 2)[0] == 0
    g2 = orc.make_graph_dict(arcs, rng.random((3, 2)).astype(np.float32))
    g2['set_mask'] = np.zeros(3, bool)
    k, s, o = _run_hip(g2, st1, ou1, 0, 5, 0.01, None, 2)
    kc, sc, _ = corc.loop_node(g2, st1, ou1, 0, 5, 0.01, None, want_out=False)
    assert k == kc and np.max(np.abs(s - sc)) < 2e-6 and o.shape == (0, 2)
    g3 = orc.make_graph_dict(np.zeros((0, 3), np.float32), rng.random((5, 2)).astype(np.float32))
    k, s, o = _run_hip(g3, st1, ou1, 0, 4, 0.01, None, 2)
    kc, sc, oc = corc.loop_node(g3, st1, ou1, 0, 4, 0.01, None)
    assert k == kc and np.max(np.abs(s - sc)) < 2e-6 and np.max(np.abs(o - oc)) < 2e-6
    # large magnitudes: the piece split is exact for any finite fp32 value (no overflow of the remainders)
    g4, st4, ou4, s04 = _case(rng, n=100, d=8, hidden=(16,), act='relu', gain=0.5)
    big = dict(st4, weights=[w * (64.0 if i == 0 else 1.0 / 64.0 if i == 2 else 1.0) for i, w in enumerate(st4['weights'])])
    k, s, o = _run_hip(g4, big, ou4, 8, 10, 0.01, s04, 2)
    kc, sc, oc = corc.loop_node(g4, big, ou4, 8, 10, 0.01, s04)
    assert k == kc and np.max(np.abs(s - sc)) < 5e-6 * max(1.0, float(np.max(np.abs(sc))))


def test_iteration_count_tracks_threshold():
    """k is decided on the device by the fused '>' test; sweep thresholds so that k changes and must track the oracle."""
    rng = np.random.default_rng(8)
    g, st, ou, s0 = _case(rng, n=500, d=16, hidden=(32,), gain=0.5)
    seen = set()
    for thr in [0.5, 0.1, 0.03, 0.01, 0.003, 1e-3, 1e-4]:
        kc, sc, _ = corc.loop_node(g, st, ou, 16, 40, thr, s0)
        k, s, _ = _run_hip(g, st, ou, 16, 40, thr, s0)
        assert k == kc and np.array_equal(s, sc)
        seen.add(k)
    assert len(seen) >= 4


def _models(st, ou, d, max_it, thr, cls):
    from GNN.MLP import MLP
    def build(net):
        w = net['weights']
        n_dense = len(net['activations'])
        m = MLP(input_dim=w[0].shape[0], layers=[w[2 * i].shape[1] for i in range(n_dense)], activations=net['activations'],
                kernel_initializer='zeros', bias_initializer='zeros', dropout_rate=0.1, dropout_pos=0,
                batch_normalization=net['batch_normalization'])
        m.set_weights(w)
        return m
    from GNN import losses
    gnn = cls(net_state=build(st), net_output=build(ou), optimizer=None, loss_function=losses.categorical_crossentropy,
              loss_arguments=None, state_vect_dim=d, max_iteration=max_it, threshold=thr, addressed_problem='c')
    gnn.impl = 1            # the bit-exact fused path: these tests compare with the C oracle bit for bit
    return gnn


@pytest.mark.parametrize('prefix,mode', [('simple/average/n', 'average'), ('simple/sum/n', 'sum'), ('random/3', 'average'),
                                         ('merge_n/normalized', 'normalized'), ('merge_n/average', 'average')])
def test_facade_on_reference_fixtures(prefix, mode):
    """GraphObject -> GraphTensor -> GNNnodeBased.Loop on graphs whose matrices were produced by the reference's own code."""
    from GNN.GNN import GNNnodeBased
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(2)
    arcs, nodes = GOLD[f'{prefix}/arcs'], GOLD[f'{prefix}/nodes']
    go = GraphObject(arcs=arcs, nodes=nodes, targets=GOLD[f'{prefix}/targets'], aggregation_mode=mode)
    nl, al = go.DIM_NODE_LABEL, go.DIM_ARC_LABEL
    scale = 1.0 / max(1.0, float(np.abs(nodes).max()))
    st = make_mlp(rng, al + 2 * nl, [6, nl], 'tanh', gain=0.5 * scale)
    ou = make_mlp(rng, nl, [2], 'softmax')
    gnn = _models(st, ou, 0, 20, 0.01, GNNnodeBased)
    k, s, o = gnn.Loop(go)
    kc, sc, oc = corc.loop_node(orc.make_graph_dict(arcs, nodes, mode), st, ou, 0, 20, 0.01)
    assert k == kc and np.array_equal(s, sc) and np.array_equal(o, oc)
    assert np.array_equal(gnn(go), oc)
    it, loss, targs, out = gnn.evaluate_single_graph(go, training=False)
    assert it == kc and np.array_equal(out, oc) and targs.shape == oc.shape and np.isfinite(loss)
    gnn.extra_metrics = {}
    metrics = gnn.test(go)
    assert metrics['It'] == int(kc) and np.isfinite(metrics['Loss'])


def test_graph_based_readout_and_errors():
    from GNN.GNN import GNNgraphBased, GNNnodeBased
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(4)
    p = 'merge_g/average'
    go = GraphObject(arcs=GOLD[f'{p}/arcs'], nodes=GOLD[f'{p}/nodes'], targets=GOLD[f'{p}/targets'], problem_based='g',
                     NodeGraph=GOLD[f'{p}/NodeGraph'])
    nl, al = go.DIM_NODE_LABEL, go.DIM_ARC_LABEL
    st, ou = make_mlp(rng, al + 2 * nl, [8, nl], 'selu', gain=0.5), make_mlp(rng, nl, [2], 'softmax')
    gnn = _models(st, ou, 0, 15, 0.01, GNNgraphBased)
    k, s, o = gnn.Loop(go)
    gd = orc.make_graph_dict(go.arcs, go.nodes, 'average', NodeGraph=go.NodeGraph)
    kc, sc, on = corc.loop_node(gd, st, ou, 0, 15, 0.01)
    assert k == kc and np.array_equal(s, sc)
    assert o.shape == (3, 2) and np.array_equal(o, corc.readout(go.NodeGraph, on))
    np.testing.assert_allclose(o, orc.loop_graph(gd, st, ou, 0, 15, 0.01, dtype=np.float64)[2], atol=1e-5)
    # repeated Loops on ONE device-resident graph: from the second on the NodeGraph is cached with the device loop and the persistent
    # small-graph launch folds the readout in (gnn_small.hip: one more grid barrier, result in pinned host memory) - same bits as k_readout
    from GNN.graph_class import GraphTensor
    gt = GraphTensor.fromGraphObject(go)
    for _ in range(3):
        k2, s2, o2 = gnn.Loop(gt)
        assert k2 == kc and np.array_equal(s2, sc) and np.array_equal(o2, o)
    # ... and a changed NodeGraph (other weights) on the same device graph is noticed: separate readout once, then folded in again
    gt.NodeGraph = (np.asarray(go.NodeGraph) * 2.0).astype(np.float32)
    if hasattr(gt, '_ng_csr'): gt._ng_csr = None
    for _ in range(3):
        assert np.array_equal(gnn.Loop(gt)[2], corc.readout(gt.NodeGraph, on))
    node_only = GraphObject(arcs=go.arcs, nodes=go.nodes, targets=np.zeros((go.nodes.shape[0], 2)))
    with pytest.raises(ValueError):
        gnn.Loop(node_only)                                        # reference GNN.py:322
    with pytest.raises(ValueError):
        _models(st, make_mlp(rng, nl + 1, [2], 'softmax'), 0, 5, 0.01, GNNnodeBased).Loop(go)   # wrong net_output width


@pytest.mark.parametrize('d', [0, 6])
def test_edge_based_readout(d):
    """GNNedgeBased.Loop (reference GNN.py:286-302): per-arc readout [F(i0) | F(i1) | arc label] through set/output masks
    over the arcs, against the oracle (states from the C oracle, readout rows per the reference, net_output by the C oracle)."""
    from GNN.GNN import GNNedgeBased
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(40 + d)
    n, nl, al = 300, 3, 2
    arcs = random_arcs(rng, n, 900, al)                      # symmetric, lexicographically sorted (the consistent case)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    e = len(arcs)
    set_mask, output_mask = rng.random(e) < 0.8, rng.random(e) < 0.7
    go = GraphObject(arcs=arcs, nodes=nodes, targets=np.zeros((int(output_mask.sum()), 2)), problem_based='a',   # one target per output_mask arc
                     set_mask=set_mask, output_mask=output_mask)
    ins, ls = orc.get_inout_dims('state', nl, al, 2, 'a', d, [12])
    ino, lo = orc.get_inout_dims('output', nl, al, 2, 'a', d, None)
    ds, nlc = (d if d else nl), (nl if d else 0)
    assert ino == 2 * (ds + nlc) + al
    st, ou = make_mlp(rng, ins, ls, 'tanh', gain=0.6), make_mlp(rng, ino, lo, 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32) if d else None
    gnn = _models(st, ou, d, 20, 0.01, GNNedgeBased)
    k, s, o = gnn.Loop(go, state0=s0)
    gd = orc.make_graph_dict(arcs, nodes, 'average')
    node_out = make_mlp(rng, ds + nlc, [2], 'softmax')       # only to satisfy the node-based oracle signature
    kc, sc, _ = corc.loop_node(gd, st, node_out, d, 20, 0.01, s0, want_out=False)
    gd['set_mask'], gd['output_mask'] = set_mask, output_mask
    feats = orc.edge_features(gd, sc, d)
    oc = corc.mlp_forward(feats, ou['weights'], ou['activations'], True)
    assert k == kc and np.array_equal(s, sc)
    assert o.shape == oc.shape == (int((set_mask & output_mask).sum()), 2) and np.array_equal(o, oc)
    it, loss, targs, out = gnn.evaluate_single_graph(go, training=False)
    assert targs.shape == out.shape
    k2, s2, o2 = gnn.Loop(go, state0=s0)                      # cached loop, second call
    assert k2 == k and np.array_equal(o2, o)


def test_mutag_batches_graph_based():
    """BASELINE config 2: MUTAG batches of 32 graphs, graph-based, state = node labels (14), net_state 31 -> [32, 32, 14],
    max_iteration 50: k, states and graph outputs against the C oracle; LGNN on the same batch."""
    import load_MUTAG
    from GNN.GNN import GNNgraphBased
    from GNN.LGNN import LGNN
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(31)
    graphs = load_MUTAG.load(limit=96)
    st, ou = make_mlp(rng, 3 + 2 * 14, [32, 32, 14], 'selu', gain=0.7), make_mlp(rng, 14, [2], 'softmax')
    gnn = _models(st, ou, 0, 50, 0.01, GNNgraphBased)
    for b in range(3):
        batch = GraphObject.merge(graphs[32 * b:32 * b + 32], problem_based='g', aggregation_mode='average')
        k, s, o = gnn.Loop(batch)
        gd = orc.make_graph_dict(batch.arcs, batch.nodes, 'average', NodeGraph=batch.NodeGraph)
        kc, sc, on = corc.loop_node(gd, st, ou, 0, 50, 0.01)
        assert k == kc and np.array_equal(s, sc)
        assert o.shape == (32, 2) and np.array_equal(o, corc.readout(batch.NodeGraph, on))
        it, loss, targs, out = gnn.evaluate_single_graph(batch, training=False)
        assert targs.shape == (32, 2) and np.array_equal(out, o)
    gnn.impl = 0
    assert np.array_equal(gnn.Loop(batch)[2], o)          # per-op kernels give the same bits
    # graph-based LGNN, 2 layers, outputs propagated
    ins, ls = orc.get_inout_dims('state', 14, 3, 2, 'g', 0, [32], layer=1, get_output=True)
    ino, lo = orc.get_inout_dims('output', 14, 3, 2, 'g', 0, None, layer=1, get_output=True)
    st1, ou1 = make_mlp(rng, ins, ls, 'selu', gain=0.7), make_mlp(rng, ino, lo, 'softmax')
    lgnn = LGNN([_models(st, ou, 0, 50, 0.01, GNNgraphBased), _models(st1, ou1, 0, 50, 0.01, GNNgraphBased)], False, True, None, None, None, 'c')
    K, state, outs = lgnn.Loop(batch)
    gnns = [dict(net_state=st, net_output=ou, state_vect_dim=0, max_iteration=50, threshold=0.01),
            dict(net_state=st1, net_output=ou1, state_vect_dim=0, max_iteration=50, threshold=0.01)]
    K64, s64, o64 = orc.lgnn_loop(gd, gnns, False, True, True, None, np.float64)
    assert K == K64 and len(outs) == 2 and np.max(np.abs(outs[1] - o64[1])) < 1e-5 and np.max(np.abs(outs[0] - o64[0])) < 1e-5


def test_lgnn_stack_on_device():
    from GNN.GNN import GNNnodeBased
    from GNN.LGNN import LGNN
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(6)
    arcs = random_arcs(rng, 400, 1200, 1)
    nodes = (2 * rng.random((400, 3)) - 1).astype(np.float32)
    set_mask = rng.random(400) < 0.75
    go = GraphObject(arcs=arcs, nodes=nodes, targets=np.zeros((400, 2)), set_mask=set_mask)
    gd = orc.make_graph_dict(arcs, nodes, 'average', set_mask=set_mask)
    for get_state, get_output in [(False, True), (True, True), (True, False)]:
        layers, gnns, models = 3, [], []
        for layer in range(layers):
            ins, ls = orc.get_inout_dims('state', 3, 1, 2, 'n', 8, [16], layer=layer, get_state=get_state, get_output=get_output)
            ino, lo = orc.get_inout_dims('output', 3, 1, 2, 'n', 8, None, layer=layer, get_state=get_state, get_output=get_output)
            st, ou = make_mlp(rng, ins, ls, 'selu', gain=0.5), make_mlp(rng, ino, lo, 'softmax')
            gnns.append(dict(net_state=st, net_output=ou, state_vect_dim=8, max_iteration=12, threshold=0.01))
            models.append(_models(st, ou, 8, 12, 0.01, GNNnodeBased))
        s0s = [(0.1 * rng.standard_normal((400, 8))).astype(np.float32) for _ in range(layers)]
        lgnn = LGNN(models, get_state, get_output, None, None, None, 'c')
        K, state, outs = lgnn.Loop(go, state0=s0s)
        # bit-exact chain: C oracle per layer + the reference's relabelling rule
        gtmp, Kc, outs_c = dict(gd), [], []
        for gnn, s0 in zip(gnns, s0s):
            kc, sc, oc = corc.loop_node(gtmp, gnn['net_state'], gnn['net_output'], 8, 12, 0.01, s0)
            Kc.append(kc)
            outs_c.append(oc)
            gtmp = orc.update_graph(gd, sc, oc, get_state, get_output)
        assert K == Kc and np.array_equal(state, sc)
        for a, b in zip(outs, outs_c):
            assert np.array_equal(a, b)
        K64, s64, o64 = orc.lgnn_loop(gd, gnns, get_state, get_output, False, s0s, np.float64)
        assert K == K64 and np.max(np.abs(outs[-1] - o64[-1])) < 1e-5
    # __call__ / predict on a stack without random initial states (state_vect_dim == 0: state starts from the labels)
    gnns, models = [], []
    for layer in range(2):
        ins, ls = orc.get_inout_dims('state', 3, 1, 2, 'n', 0, [8], layer=layer, get_output=True)
        ino, lo = orc.get_inout_dims('output', 3, 1, 2, 'n', 0, None, layer=layer, get_output=True)
        st, ou = make_mlp(rng, ins, ls, 'tanh', gain=0.5), make_mlp(rng, ino, lo, 'softmax')
        gnns.append(dict(net_state=st, net_output=ou, state_vect_dim=0, max_iteration=10, threshold=0.01))
        models.append(_models(st, ou, 0, 10, 0.01, GNNnodeBased))
    lgnn = LGNN(models, False, True, None, None, None, 'c')
    k0, s0_, o0 = corc.loop_node(gd, gnns[0]['net_state'], gnns[0]['net_output'], 0, 10, 0.01)
    k1, s1_, o1 = corc.loop_node(orc.update_graph(gd, s0_, o0, False, True), gnns[1]['net_state'], gnns[1]['net_output'], 0, 10, 0.01)
    assert np.array_equal(lgnn(go), o1) and np.array_equal(lgnn.predict(go, 0), o0)
    assert [np.array_equal(a, b) for a, b in zip(lgnn.predict(go, 'all'), [o0, o1])] == [True, True]


def test_engine_rng_and_call_order_errors():
    e = _engine()
    rng = np.random.default_rng(9)
    g, st, ou, _ = _case(rng, n=2048, d=8)
    loop = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), 8, 0, 0.01)
    with pytest.raises(e.EngineError):
        loop.state()                                               # before run
    with pytest.raises(e.EngineError):
        loop.run()                                                 # state_vect_dim > 0 without state0
    loop.set_state0(None, seed=3)
    assert loop.run() == 0
    s = loop.state()
    assert abs(float(s.mean())) < 0.01 and abs(float(s.std()) - 0.1) < 0.01   # N(0, 0.1^2) as GNN.py:262
    with pytest.raises(NotImplementedError):
        loop.run(training=True)
    with pytest.raises(ValueError):
        e.Loop(_device_graph(g), e.Mlp(ou['weights'], ou['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), 8, 5, 0.01)


def test_rccl_path_single_rank():
    """The multi-GPU code path (lazy librccl load, ncclCommInitRank, grouped all-gather of state rows + flag block after
    every body, allreduce-max used by bench.py) with a 1-rank communicator: must not change a bit."""
    e = _engine()
    rng = np.random.default_rng(12)
    g, st, ou, s0 = _case(rng, n=1500, d=64, hidden=(128, 128), gain=0.6)
    comm = e.Comm(e.Comm.unique_id(), 0, 1, 0)
    assert comm.allreduce_max(3.5) == 3.5
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    res = []
    for c in (None, comm):
        for impl in (1, 0):
            loop = e.Loop(_device_graph(g), mst, mou, 64, 30, 0.01, c)
            loop.set_impl(impl)
            loop.set_state0(s0)
            res.append((loop.run(), loop.state(), loop.output()))
            loop.close()
    kc, sc, oc = corc.loop_node(g, st, ou, 64, 30, 0.01, s0)
    for k, s, o in res:
        assert k == kc and np.array_equal(s, sc) and np.array_equal(o, oc)
    comm.close()


def test_full_size_properties():
    """BASELINE config 3 size (1M nodes / ~10M arcs, D=64, 135->128->128->64): properties that do not need the oracle on
    the full graph: sampled rows of one step against NumPy float64, run-to-run determinism, fused == unfused bit for bit."""
    from GNN import GNN_utils as utils
    e = _engine()
    rng = np.random.default_rng(123)
    s = utils.syntheticGraph(1_000_000, 10.0, seed=20261003)
    n, d, nl, al = s['n_nodes'], 64, 3, 1
    st = make_mlp(rng, al + 2 * (nl + d), [128, 128, d], 'selu')
    ou = make_mlp(rng, nl + d, [2], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    results = {}
    for impl in (2, 1, 0):
        loop = e.Loop(graph, mst, mou, d, 3, 0.0)
        used = loop.set_impl(impl)
        assert used == impl
        loop.set_state0(s0)
        assert loop.run() == 3
        results[impl] = (used, loop.state(), loop.output())
        if impl >= 1:
            assert loop.run() == 3 and np.array_equal(loop.state(), results[impl][1])     # deterministic
        loop.close()
    assert np.array_equal(results[0][1], results[1][1]) and np.array_equal(results[0][2], results[1][2])
    # the default split-arithmetic path: fp32 rounding noise away from the exact one on every one of the 64M state values
    ds_, do_ = float(np.max(np.abs(results[2][1] - results[1][1]))), float(np.max(np.abs(results[2][2] - results[1][2])))
    scale = max(1.0, float(np.max(np.abs(results[1][1]))))
    assert ds_ < 1e-5 * scale and do_ < 1e-5, (ds_, do_, scale)
    # one step on a row sample, float64 NumPy: rows are independent given the previous iterate
    loop = e.Loop(graph, mst, mou, d, 1, 0.0)
    loop.set_impl(1)
    loop.set_state0(s0)
    assert loop.run() == 1
    s1 = loop.state()
    rows = rng.choice(n, 3000, replace=False)
    ip, src, w = s['indptr'], s['adj_src'], s['adj_w']
    inp = np.zeros((len(rows), 135))
    for t, r in enumerate(rows):
        e0, e1 = ip[r], ip[r + 1]
        inp[t, :64] = s0[r]
        inp[t, 64:67] = s['nodes'][r]
        inp[t, 67:131] = (w[e0:e1, None].astype(np.float64) * s0[src[e0:e1]]).sum(0)
        inp[t, 131:134] = (w[e0:e1, None].astype(np.float64) * s['nodes'][src[e0:e1]]).sum(0)
        inp[t, 134] = (s['arc_w'][e0:e1].astype(np.float64) * s['arc_labels_csr'][e0:e1, 0]).sum()
    ref = orc.mlp_forward(inp, st['weights'], st['activations'], True, np.float64)
    assert np.max(np.abs(s1[rows] - ref)) < 1e-5


@pytest.mark.parametrize('mode', ['sum', 'normalized', 'average'])
@pytest.mark.parametrize('sort', [True, False])
def test_device_graph_build_matches_host_build(mode, sort):
    """gnn_graph_create_from_arcs (radix sorts + histogram on the GPU) against the host chain buildArcNode / buildAdiacency /
    COO2SparseTransposedTensor (reference graph_class.py:90-121, :365-372): identical index arrays and weights, and a Loop
    on either handle gives identical bits."""
    from GNN.graph_class import GraphObject, GraphTensor
    rng = np.random.default_rng(5 + sort)
    n, nl, al = 900, 3, 2
    arcs = random_arcs(rng, n, 4000, al, sort=sort)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    set_mask = rng.random(n) < 0.8
    g = orc.make_graph_dict(arcs, nodes, mode)           # host restatement of the reference's matrices, arcs used as given
    dev = GraphTensor.fromArcs(nodes, arcs, np.zeros((n, 2)), set_mask=set_mask, aggregation_mode=mode)
    for got, want in zip(dev.Adjacency, g['adjT']):
        assert np.array_equal(got, want)
    for got, want in zip(dev.ArcNode, g['arcT']):
        assert np.array_equal(got, want)
    if sort:     # GraphObject keeps sorted unique arcs as they are: the two GraphTensors must coincide
        host = GraphTensor.fromGraphObject(GraphObject(arcs=arcs, nodes=nodes, targets=np.zeros((n, 2)), set_mask=set_mask, aggregation_mode=mode))
        assert all(np.array_equal(a, b) for a, b in zip(host.Adjacency, dev.Adjacency))
        assert all(np.array_equal(a, b) for a, b in zip(host.ArcNode, dev.ArcNode))
    st, ou = make_mlp(rng, al + 2 * (nl + 8), [16, 8], 'tanh', gain=0.6), make_mlp(rng, nl + 8, [2], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, 8))).astype(np.float32)
    e = _engine()
    g['set_mask'] = set_mask
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    res = []
    for graph in (dev.device_graph(), _device_graph(g)):
        loop = e.Loop(graph, mst, mou, 8, 20, 0.01)
        loop.set_impl(1)
        loop.set_state0(s0)
        res.append((loop.run(), loop.state(), loop.output()))
    assert res[0][0] == res[1][0] and np.array_equal(res[0][1], res[1][1]) and np.array_equal(res[0][2], res[1][2])
    kc, sc, oc = corc.loop_node(g, st, ou, 8, 20, 0.01, s0)
    assert res[0][0] == kc and np.array_equal(res[0][1], sc) and np.array_equal(res[0][2], oc)
    with pytest.raises(ValueError):
        GraphTensor.fromArcs(nodes, arcs, np.zeros((n, 2)), aggregation_mode='mean')     # reference graph_class.py:86


@pytest.mark.parametrize('get_state,get_output', [(True, True), (False, True), (True, False)])
def test_edge_based_lgnn_stack_on_device(get_state, get_output):
    """LGNN of GNNedgeBased layers (reference LGNN.py:253-254): the output of a layer widens the ARC labels (scattered through
    the arc mask, original arc order), its state the node labels; relabelling on the device (gnn_graph_derive_edge +
    gnn_graph_update_labels), against the C oracle per layer chained by the reference's rule."""
    from GNN.GNN import GNNedgeBased
    from GNN.LGNN import LGNN
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(77)
    n, nl, al, d, t, layers = 250, 3, 2, 6, 2, 3
    arcs = random_arcs(rng, n, 700, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    e = len(arcs)
    set_mask, output_mask = rng.random(e) < 0.8, rng.random(e) < 0.75
    go = GraphObject(arcs=arcs, nodes=nodes, targets=np.zeros((int(output_mask.sum()), t)), problem_based='a', set_mask=set_mask, output_mask=output_mask)
    gd = orc.make_graph_dict(arcs, nodes, 'average')
    gd['set_mask'], gd['output_mask'] = set_mask, output_mask
    gnns, models = [], []
    for layer in range(layers):
        ins, ls = orc.get_inout_dims('state', nl, al, t, 'a', d, [12], layer=layer, get_state=get_state, get_output=get_output)
        ino, lo = orc.get_inout_dims('output', nl, al, t, 'a', d, None, layer=layer, get_state=get_state, get_output=get_output)
        st, ou = make_mlp(rng, ins, ls, 'tanh', gain=0.5), make_mlp(rng, ino, lo, 'softmax')
        gnns.append(dict(net_state=st, net_output=ou))
        models.append(_models(st, ou, d, 12, 0.01, GNNedgeBased))
    s0s = [(0.1 * rng.standard_normal((n, d))).astype(np.float32) for _ in range(layers)]
    lgnn = LGNN(models, get_state, get_output, None, None, None, 'c')
    K, state, outs = lgnn.Loop(go, state0=s0s)
    gtmp, Kc = dict(gd), []
    for gnn, s0, got in zip(gnns, s0s, outs):
        nn_, al_ = np.asarray(gtmp['nodes']).shape[1], np.asarray(gtmp['arcs']).shape[1] - 2
        assert gnn['net_state']['weights'][0].shape[0] == al_ + 2 * (nn_ + d)       # get_inout_dims agrees with the relabelled widths
        node_side = dict(gtmp, set_mask=np.ones(n, bool), output_mask=np.ones(n, bool))
        kc, sc, _ = corc.loop_node(node_side, gnn['net_state'], make_mlp(rng, nn_ + d, [1], 'linear'), d, 12, 0.01, s0, want_out=False)
        oc = corc.mlp_forward(orc.edge_features(gtmp, sc, d), gnn['net_output']['weights'], gnn['net_output']['activations'], True)
        Kc.append(kc)
        assert got.shape == oc.shape == (int((set_mask & output_mask).sum()), t) and np.array_equal(got, oc)
        gtmp = orc.update_graph_edge(gd, sc, oc, get_state, get_output)
    assert K == Kc and np.array_equal(state, sc)
    # float64 restatement of the same chain
    g64 = dict(gd)
    for gnn, s0 in zip(gnns, s0s):
        k64, s64, o64 = orc.loop_edge(g64, gnn['net_state'], gnn['net_output'], d, 12, 0.01, s0, np.float64)
        g64 = orc.update_graph_edge(gd, s64, o64, get_state, get_output, np.float64)
    assert np.max(np.abs(outs[-1] - o64)) < 1e-5
    # host form of update_graph agrees with the oracle's
    upd = lgnn.update_graph(__import__('GNN.graph_class', fromlist=['GraphTensor']).GraphTensor.fromGraphObject(go), sc, oc)
    assert np.array_equal(upd.nodes, gtmp['nodes']) and np.array_equal(upd.arcs, gtmp['arcs'])


def test_fast_path_on_skewed_degrees():
    """The Ds == 64 full-tile gather (batches of 16 entries per 16-lane group, row-boundary flushes, tail batches) on a degree
    distribution it was not tuned for: hubs with thousands of in-arcs, long runs of isolated nodes, rows of exactly 16 / 17 entries."""
    rng = np.random.default_rng(99)
    n, d, nl, al = 2048, 64, 3, 1
    deg = np.zeros(n, np.int64)
    deg[rng.choice(n, 6, replace=False)] = rng.integers(1500, 4000, 6)          # hubs
    body = rng.choice(n, 900, replace=False)
    deg[body] = np.maximum(deg[body], rng.integers(1, 40, 900))
    deg[100:132] = 16; deg[132:164] = 17; deg[164:260] = 0                       # whole tiles of 16s, 17s and isolated nodes
    dst = np.repeat(np.arange(n), deg)
    src = rng.integers(0, n, len(dst))
    keep = src != dst
    pairs = np.unique(np.stack([src[keep], dst[keep]], 1), axis=0)               # (src, dst)-sorted, duplicate free
    arcs = np.concatenate([pairs.astype(np.float32), (2 * rng.random((len(pairs), al)) - 1).astype(np.float32)], axis=1)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    g = orc.make_graph_dict(arcs, nodes, 'average')
    g['set_mask'] = rng.random(n) < 0.9
    st, ou = make_mlp(rng, al + 2 * (nl + d), [128, 128, d], 'selu', gain=0.6), make_mlp(rng, nl + d, [2], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    kc, sc, oc = corc.loop_node(g, st, ou, d, 12, 0.01, s0)
    for impl in (1, 0):
        k, s, o = _run_hip(g, st, ou, d, 12, 0.01, s0, impl)
        assert k == kc and np.array_equal(s, sc) and np.array_equal(o, oc), impl
    k, s, o = _run_hip(g, st, ou, d, 12, 0.01, s0, 2)
    assert k == kc and np.max(np.abs(s - sc)) < 2e-6 * max(1.0, float(np.max(np.abs(sc)))) and np.max(np.abs(o - oc)) < 2e-6


@pytest.mark.parametrize('d,hidden,expect_fused', [(60, (128,), True), (68, (96,), True), (128, (128,), False), (66, (64,), False), (16, (200,), False)])
def test_wide_states_and_fallback(d, hidden, expect_fused):
    """State widths around the tuned shape: Ds = 60 / 68 run fused through the generic gather (68: three feature tiles of state),
    Ds = 128 (the concat no longer fits eight LDS tiles), Ds = 66 (neither a multiple of 4 nor <= 64) and a hidden width above
    128 fall back to the per-op kernels - same bits either way."""
    e = _engine()
    rng = np.random.default_rng(300 + d)
    g, st, ou, s0 = _case(rng, n=333, d=d, nl=3, al=2, hidden=hidden, act='tanh', gain=0.5)
    kc, sc, oc = corc.loop_node(g, st, ou, d, 10, 0.01, s0)
    loop = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 10, 0.01)
    assert (loop.set_impl(1) == 1) == expect_fused
    loop.set_state0(s0)
    k = loop.run()
    assert k == kc and np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc)
    assert (loop.set_impl(2) == 2) == expect_fused
    k = loop.run()
    s2 = loop.state()
    err = float(np.max(np.abs(s2 - sc)))
    assert k == kc and err < 2e-6 * max(1.0, float(np.max(np.abs(sc)))), \
        f'k {k} (oracle {kc}), max |state - oracle| {err}, NaNs {int(np.isnan(s2).sum())}, rows off by > 1e-5: {np.nonzero(np.max(np.abs(s2 - sc), axis=1) > 1e-5)[0][:20]}'


@pytest.mark.parametrize('d,nl,al,hidden,act,n', [(0, 14, 3, (32, 32), 'selu', 970), (0, 3, 1, (), 'selu', 880), (8, 3, 2, (16,), 'tanh', 4099),
                                                   (5, 2, 1, (7, 9), 'relu', 33), (16, 3, 1, (24,), 'sigmoid', 8192),
                                                   (24, 3, 2, (32, 20), 'selu', 700), (20, 2, 1, (16,), 'tanh', 5000)])
def test_persistent_small_graph_loop(d, nl, al, hidden, act, n):
    """Small graphs run initial state, first condition and every body of the loop (GNN.py:266-271) inside ONE persistent launch
    with a grid barrier between the bodies.  k, states and outputs: bit-identical to the C oracle and to one launch per body,
    for both fused modes, over several thresholds (so that the loop leaves at different bodies) and on repeated runs.  Up to 4,096
    nodes the launch works on 16-node tiles (gnn_small16.hip), above on 32-node tiles (gnn_small.hip); states of up to 16 floats travel
    in 64-byte exchange rows, wider ones in 128-byte rows: the shapes cover all four combinations."""
    e = _engine()
    rng = np.random.default_rng(7000 + n + d)
    g, st, ou, s0 = _case(rng, n=n, d=d, nl=nl, al=al, hidden=hidden, act=act)
    g['set_mask'] = rng.random(n) < 0.8
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    graph = _device_graph(g)
    for thr, max_it in ((0.01, 30), (0.2, 30), (0.0, 7), (0.01, 0), (1e9, 5)):
        kc, sc, oc = corc.loop_node(g, st, ou, d, max_it, thr, s0)
        for impl in (1, 2):
            loop = e.Loop(graph, mst, mou, d, max_it, thr)
            loop.set_impl(impl)
            assert loop.set_persistent(True)
            if d: loop.set_state0(s0)
            for _ in range(2):
                k = loop.run()
                assert k == kc and np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc), (thr, max_it, impl)
            assert not loop.set_persistent(False)
            k = loop.run()                                   # one launch per body
            if impl == 1: assert k == kc and np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc)
            loop.close()
    big = e.Loop(_device_graph(_case(rng, n=8193, d=4)[0]), *[e.Mlp(x['weights'], x['activations'], True) for x in _case(np.random.default_rng(1), n=8193, d=4)[1:3]], 4, 5, 0.01)
    assert not big.set_persistent(True)                   # 257 tiles: not all resident at once by the conservative rule


def test_persistent_small_graph_loop_random_shapes():
    """The persistent launch on 30 seeded random shapes (state width 0 / 1 .. 32, label widths, 0 - 2 hidden layers of 1 .. 32 units,
    every activation, 17 .. 6,000 nodes, sparse and dense rows): k, states and outputs bit-identical to the C oracle.  The tile layout of
    the kernels depends on all of these (K-steps of layer 0, row stride of the LDS tile, 64- or 128-byte exchange rows, one or two
    feature tiles in the last layer, arcs per gather round, 16- or 32-node tiles)."""
    e = _engine()
    rng = np.random.default_rng(20261004)
    acts = ['selu', 'tanh', 'relu', 'sigmoid', 'elu', 'linear']
    for case in range(30):
        d = int(rng.choice([0, 0, 1, 3, 4, 7, 12, 16, 17, 24, 31, 32]))
        nl = int(rng.integers(1, 9)) if d else int(rng.integers(1, 33))
        al = int(rng.integers(1, 5))
        hidden = tuple(int(x) for x in rng.integers(1, 33, size=int(rng.integers(0, 3))))
        n = int(rng.choice([17, 33, 100, 640, 1999, 4096, 4097, 6000]))
        act = acts[case % len(acts)]
        g, st, ou, s0 = _case(rng, n=n, d=d, nl=nl, al=al, hidden=hidden, act=act, deg=int(rng.choice([1, 4, 11])))
        mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
        graph = _device_graph(g)
        max_it, thr = int(rng.integers(1, 12)), float(rng.choice([0.0, 0.01, 0.1]))
        kc, sc, oc = corc.loop_node(g, st, ou, d, max_it, thr, s0)
        loop = e.Loop(graph, mst, mou, d, max_it, thr)
        loop.set_impl(1)
        if not loop.set_persistent(True):
            loop.close(); continue
        if d: loop.set_state0(s0)
        k = loop.run()
        assert k == kc and np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc), (case, d, nl, al, hidden, n, act, max_it, thr, k, kc)
        loop.close()


def test_fused_kernel_random_shapes():
    """The per-iteration fused kernel (one launch per body) on 24 seeded random shapes: state widths 0 / 1 .. 64, label widths, 0 - 2
    hidden layers of 1 .. 128 units, every activation, partial last tiles, sparse and dense rows.  impl 1 bit-identical to the C oracle,
    impl 2 (split arithmetic) the SAME k (certified gate: a run whose stopping gate is borderline is repeated on impl 1) and within tolerance."""
    e = _engine()
    rng = np.random.default_rng(20261005)
    acts = ['selu', 'tanh', 'relu', 'sigmoid', 'elu', 'linear']
    ran = 0
    for case in range(24):
        d = int(rng.choice([0, 1, 4, 8, 12, 20, 32, 40, 60, 64]))
        nl = int(rng.integers(1, 9)) if d else int(rng.choice([1, 3, 8, 16, 33, 64]))
        al = int(rng.integers(1, 5))
        hidden = tuple(int(x) for x in rng.choice([1, 7, 16, 31, 32, 33, 64, 96, 128], size=int(rng.integers(0, 3))))
        n = int(rng.choice([31, 32, 257, 1000, 3000]))
        act = acts[case % len(acts)]
        g, st, ou, s0 = _case(rng, n=n, d=d, nl=nl, al=al, hidden=hidden, act=act, gain=0.5, deg=int(rng.choice([1, 4, 11])))
        mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
        max_it, thr = int(rng.integers(1, 9)), float(rng.choice([0.0, 0.01]))
        kc, sc, oc = corc.loop_node(g, st, ou, d, max_it, thr, s0)
        loop = e.Loop(_device_graph(g), mst, mou, d, max_it, thr)
        if loop.set_impl(1) != 1:
            loop.close(); continue                      # shape outside the fused kernel (covered by test_wide_states_and_fallback)
        loop.set_persistent(False)
        if d: loop.set_state0(s0)
        k = loop.run()
        assert k == kc and np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc), (case, d, nl, al, hidden, n, act, max_it, thr, k, kc)
        assert loop.set_impl(2) == 2
        k2 = loop.run()
        assert k2 == kc, (case, k2, kc, loop.gate_info())          # certified gate: the default path's k IS the exact chain's (gnn_hip.h, gnn_loop_set_impl)
        err = float(np.max(np.abs(loop.state() - sc)))
        assert err < 1e-5 * max(1.0, float(np.max(np.abs(sc)))), (case, d, nl, al, hidden, n, act, max_it, thr, err)
        loop.close()
        ran += 1
    assert ran >= 12


@pytest.mark.parametrize('d,hidden,n', [(64, (128, 128), 4096), (60, (128,), 333), (40, (64,), 1000)])
def test_default_path_certified_gate_borderline_threshold(d, hidden, n):
    """The k contract of the default path (impl 2; reference GNN/GNN.py:202-220: strict `>` on float32 norms, reduce_any): a threshold
    placed ON the largest distance / norm ratio of some body makes that body's gate a tie that the two arithmetics could break
    differently.  The split path must notice (no robust mover, a borderline node), repeat the Loop on impl 1 and return the exact chain's
    k, state and output bit for bit; a threshold well away from every ratio must NOT trigger the repeat."""
    e = _engine()
    rng = np.random.default_rng(5000 + d)
    g, st, ou, s0 = _case(rng, n=n, d=d, nl=3, al=2, hidden=hidden, act='tanh', gain=0.5, deg=4)
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    graph = _device_graph(g)
    # states after 3 and 4 bodies of the exact chain (threshold 0), the per-node ratio of the gate that follows body 4 in oracle order
    _, s3, _ = corc.loop_node(g, st, ou, d, 3, 0.0, s0)
    _, s4, _ = corc.loop_node(g, st, ou, d, 4, 0.0, s0)
    dist = np.zeros(n, np.float32); nrm = np.zeros(n, np.float32)
    for c in range(d):
        df = s4[:, c] - s3[:, c]
        dist = dist + df * df
        nrm = nrm + s3[:, c] * s3[:, c]
    ratio = np.sqrt(dist) / np.sqrt(nrm)
    thr_tie = float(np.max(ratio))                      # the top node sits on the threshold to within float32 rounding
    for thr, expect_repeat in ((thr_tie, True), (thr_tie * 1.5, False), (thr_tie * 0.5, False)):
        kc, sc, oc = corc.loop_node(g, st, ou, d, 12, thr, s0)
        loop = e.Loop(graph, mst, mou, d, 12, thr)
        assert loop.set_impl(2) == 2
        loop.set_persistent(False)
        loop.set_state0(s0)
        k = loop.run()
        repeated, total = loop.gate_info()
        assert k == kc, (thr, k, kc, repeated)
        assert repeated == expect_repeat, (thr, repeated, k, kc)
        if repeated:
            assert np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc)        # the exact path's results
        else:
            assert float(np.max(np.abs(loop.state() - sc))) < 2e-6 * max(1.0, float(np.max(np.abs(sc))))
        k_again = loop.run()                            # the handle keeps working on the default path afterwards
        assert k_again == kc and loop.gate_info()[1] == total + (1 if expect_repeat else 0)
        loop.close()


def test_wide_states_after_nan_littered_memory():
    """Run-order independence of the generic fused path (state widths around the tuned shape, partial last tiles): first loops whose
    labels, initial state and therefore every state / concat / output buffer are NaN run and are destroyed, so that the device memory
    the allocator hands out next is full of NaNs; then the wide-state shapes must still give the oracle's bits (impl 1) and stay within
    tolerance (impl 2).  A kernel that reads a word nobody wrote - a pad row of a replica, an LDS hole, a weight-image slack row times a
    stale value - turns it into a NaN here instead of passing by luck.  (The diagnostic build has the systematic form: GNN_POISON=1
    fills EVERY allocation and the kernels' LDS with NaN, DESIGN.md "Oracle and parity".)"""
    e = _engine()
    rng = np.random.default_rng(77)
    for d, hidden, n in ((60, (128,), 333), (68, (96,), 333), (64, (128, 128), 1000), (128, (128,), 500)):
        g, st, ou, s0 = _case(rng, n=n, d=d, nl=3, al=2, hidden=hidden, act='tanh', gain=0.5)
        # NaN labels / initial state on the upper half of the nodes: the finite half keeps the gates open (a NaN never "moves"), and the
        # NaNs spread along the arcs into both replicas, the concat, the hidden activations and the outputs within the three bodies
        nodes_nan, s0_nan = g['nodes'].copy(), s0.copy()
        nodes_nan[n // 2:] = np.nan
        s0_nan[n // 2:] = np.nan
        g = dict(g, nodes=nodes_nan)
        litter = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 3, 0.0)
        litter.set_state0(s0_nan)
        for impl in (1, 2, 0):
            litter.set_impl(impl)
            assert litter.run() == 3
        assert np.isnan(litter.state()).mean() > 0.5
        litter.close()                                  # frees NaN-filled replicas, concat, label block, outputs
    for d, hidden, expect_fused in ((60, (128,), True), (68, (96,), True), (64, (128, 128), True), (36, (128,), True)):
        rng2 = np.random.default_rng(300 + d)
        g, st, ou, s0 = _case(rng2, n=333, d=d, nl=3, al=2, hidden=hidden, act='tanh', gain=0.5)
        kc, sc, oc = corc.loop_node(g, st, ou, d, 10, 0.01, s0)
        loop = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 10, 0.01)
        assert (loop.set_impl(1) == 1) == expect_fused
        loop.set_state0(s0)
        k = loop.run()
        assert k == kc and np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc), (d, hidden)
        assert loop.set_impl(2) == 2
        k = loop.run()
        s2 = loop.state()
        err = float(np.max(np.abs(s2 - sc)))
        assert k == kc and err < 2e-6 * max(1.0, float(np.max(np.abs(sc)))), (d, hidden, k, kc, err, int(np.isnan(s2).sum()))
        loop.close()


@pytest.mark.parametrize('n', [65_536, 65_568, 131_072, 131_104, 140_013])
def test_tile_rounds_static_and_ticketed(n):
    """How the fused kernel hands out its 32-node tiles (round 4): wave w of the launch takes tiles w and w + W statically (W = 8 waves x
    256 workgroups = 2,048 on MI355X), further tiles by ticket.  Node counts that end exactly on W tiles, one tile past it, on 2 W, one
    past that (a single ticketed tile) and with a partial last tile in the ticketed round: every row must be computed exactly once -
    impl 1 bit-identical to the C oracle, impl 2 within tolerance, over two bodies (the second reads what the first wrote)."""
    e = _engine()
    from GNN import GNN_utils as utils
    rng = np.random.default_rng(n)
    d, nl, al = 64, 3, 1
    s = utils.syntheticGraph(n, 6.0, nl, al, 2, seed=n)
    st = make_mlp(rng, al + 2 * (nl + d), [128, 128, d], 'selu', gain=0.6, bn_random=True)
    ou = make_mlp(rng, nl + d, [2], 'softmax', bn_random=True)
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    arcs = np.concatenate([np.stack([s['src'], s['dst']], 1).astype(np.float32), s['arc_labels']], axis=1)
    g = dict(nodes=s['nodes'], arcs=arcs, set_mask=np.ones(n, bool), output_mask=np.ones(n, bool),
             adjT=(s['indptr'], s['adj_src'], s['adj_w']), arcT=(s['indptr'], s['arc_perm'], s['arc_w']))
    kc, sc, oc = corc.loop_node(g, st, ou, d, 2, 0.0, s0)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    loop = e.Loop(graph, e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 2, 0.0)
    assert loop.set_impl(1) == 1
    loop.set_state0(s0)
    assert loop.run() == kc and np.array_equal(loop.state(), sc) and np.array_equal(loop.output(), oc)
    assert loop.set_impl(2) == 2
    assert loop.run() == kc
    assert float(np.max(np.abs(loop.state() - sc))) < 2e-6 * max(1.0, float(np.max(np.abs(sc))))
    loop.close(); graph.close()


def test_run_many_runs_small_loops_side_by_side():
    """gnn_loop_run_many: the persistent launches of several small graphs queued on their own streams before any is waited for (they run
    side by side), a graph too large for the persistent path in the same call; every loop's k, state and outputs bit-identical to the C
    oracle, as if run alone - also on a second call, and with the graph readout folded into the launches."""
    e = _engine()
    rng = np.random.default_rng(77)
    cases, loops, want = [], [], []
    shapes = [(570, 0, 14, 3, (32, 32)), (300, 8, 3, 2, (16,)), (1999, 5, 2, 1, (7, 9)), (33, 16, 3, 1, (24,)), (4097, 8, 3, 2, (16,)),
              (700, 24, 3, 2, (32, 20)), (9000, 8, 3, 2, (16,)), (640, 0, 5, 2, ())]
    for n, d, nl, al, hidden in shapes:
        g, st, ou, s0 = _case(rng, n=n, d=d, nl=nl, al=al, hidden=hidden, act='selu')
        lp = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 20, 0.01)
        lp.set_impl(1)
        if d: lp.set_state0(s0)
        loops.append(lp)
        want.append(corc.loop_node(g, st, ou, d, 20, 0.01, s0))
    assert sum(lp.set_persistent(True) for lp in loops) == len(shapes) - 1          # all but the 9,000-node graph
    for _ in range(2):
        ks = e.Loop.run_many(loops)
        for lp, k, (kc, sc, oc) in zip(loops, ks, want):
            assert k == kc and np.array_equal(lp.state(), sc) and np.array_equal(lp.output(), oc)
    with pytest.raises((e.EngineError, ValueError)):
        e.Loop.run_many([loops[0], loops[0]])
    for lp in loops: lp.close()


def test_run_many_limits_what_it_queues_side_by_side():
    """gnn_loop_run_many queues persistent launches side by side only while their workgroups sum to at most three per CU (a grid barrier
    needs every workgroup of every launch in flight resident); past that it collects the queued ones before it queues the next.  Thirty
    MUTAG-sized batches (about 1,100 workgroups by the call's own bound, more than 3 x 256) in one call: every loop's k, state and outputs
    as if run alone, on two calls in a row, and none of them fell back to one launch per body (the barrier never had to give up)."""
    e = _engine()
    rng = np.random.default_rng(78)
    loops, want = [], []
    for i in range(30):
        n = int(rng.integers(540, 640))
        g, st, ou, s0 = _case(rng, n=n, d=0, nl=14, al=3, hidden=(32, 32), act='selu')
        lp = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), 0, 20, 0.01)
        lp.set_impl(1)
        assert lp.set_persistent(True)
        loops.append(lp)
        want.append(corc.loop_node(g, st, ou, 0, 20, 0.01, None))
    for _ in range(2):
        ks = e.Loop.run_many(loops)
        for lp, k, (kc, sc, oc) in zip(loops, ks, want):
            assert k == kc and np.array_equal(lp.state(), sc) and np.array_equal(lp.output(), oc)
    assert all(lp.set_persistent(True) for lp in loops)      # still on the persistent path: no launch gave up
    for lp in loops: lp.close()


def test_lgnn_run_in_one_call_and_work_counters():
    """gnn_lgnn_run = LGNN.Loop (reference LGNN.py:263-290) of a whole stack through ONE C-ABI call: layer i on graphs[i], relabelling
    from the ORIGINAL graph in between; bit-identical to the C oracle chain.  gnn_counters_get: the algorithmic work of one iteration."""
    e = _engine()
    rng = np.random.default_rng(16)
    n, d, layers = 900, 8, 3
    arcs = random_arcs(rng, n, 3 * n, 1)
    nodes = (2 * rng.random((n, 3)) - 1).astype(np.float32)
    gd = orc.make_graph_dict(arcs, nodes, 'average')
    base = _device_graph(gd)
    derived = base.derive(2)                      # get_state=False, get_output=True: NL' = 3 + 2; one derived graph serves layers 1..
    nets, loops, s0s = [], [], []
    for layer in range(layers):
        ins, ls = orc.get_inout_dims('state', 3, 1, 2, 'n', d, [16], layer=layer, get_state=False, get_output=True)
        ino, lo = orc.get_inout_dims('output', 3, 1, 2, 'n', d, None, layer=layer, get_state=False, get_output=True)
        st, ou = make_mlp(rng, ins, ls, 'selu', gain=0.5), make_mlp(rng, ino, lo, 'softmax')
        nets.append((st, ou))
        lp = e.Loop(base if layer == 0 else derived, e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 12, 0.01)
        lp.set_impl(1)
        s0s.append((0.1 * rng.standard_normal((n, d))).astype(np.float32))
        lp.set_state0(s0s[-1])
        loops.append(lp)
    K = e.Loop.lgnn_run(loops, [base] + [derived] * (layers - 1), False, True)
    gtmp, Kc = dict(gd), []
    for (st, ou), s0, lp in zip(nets, s0s, loops):
        kc, sc, oc = corc.loop_node(gtmp, st, ou, d, 12, 0.01, s0)
        Kc.append(kc)
        assert np.array_equal(lp.output(), oc)
        gtmp = orc.update_graph(gd, sc, oc, False, True)
    assert K == Kc and np.array_equal(loops[-1].state(), sc)
    with pytest.raises(ValueError):                # loops[1] does not belong to graphs[1] = base
        e.Loop.lgnn_run(loops, [base] * layers, False, True)
    c = loops[0].counters()
    E, nl, al = len(arcs), 3, 1
    assert c['bytes_per_iteration'] == E * (4 * d + 8) + 4 * (n + 1) + n * (8 * d + 4 * (2 * nl + al))
    assert c['flops_per_iteration'] == n * 2 * ((al + 2 * (nl + d)) * 16 + 16 * d) + 2 * E * d and c['iterations'] == int(K[0])
    for lp in loops: lp.close()
    derived.close(); base.close()


def test_communicator_may_be_destroyed_before_its_loops():
    """gnn_comm_destroy with loops alive only marks the communicator closed; the last gnn_loop_destroy releases it."""
    e = _engine()
    rng = np.random.default_rng(3)
    g, st, ou, s0 = _case(rng, n=200, d=8)
    comm = e.Comm(e.Comm.unique_id(), 0, 1, 0)
    loop = e.Loop(_device_graph(g), e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), 8, 10, 0.01, comm)
    loop.set_state0(s0)
    comm.close()                                   # before the loop: must stay usable
    k = loop.run()
    kc, sc, oc = corc.loop_node(g, st, ou, 8, 10, 0.01, s0)
    assert k == kc and np.array_equal(loop.state(), sc)
    loop.close()


@pytest.mark.parametrize('graph_based,d', [(True, 0), (False, 6)])
def test_evaluate_runs_the_graphs_side_by_side_with_the_same_results(graph_based, d):
    """GNN_BaseClass.evaluate over a list of GraphTensors sends their Loops through gnn_loop_run_many; metrics, iteration counts and every
    output are those of evaluating the graphs one by one (reference GNN_BaseClass.py:165-189)."""
    import sys, os
    from GNN.graph_class import GraphObject, GraphTensor
    from GNN.GNN import GNNgraphBased, GNNnodeBased
    from GNN.MLP import MLP, get_inout_dims
    from GNN import optimizers, losses, GNN_utils as utils
    rng = np.random.default_rng(5 + d)
    graphs = [utils.randomGraph(int(rng.integers(15, 40)), 3, 1, 2, 0.3, problem_based='g' if graph_based else 'n') for _ in range(24)]
    batches = [GraphTensor.fromGraphObject(GraphObject.merge(graphs[i:i + 6], problem_based='g' if graph_based else 'n', aggregation_mode='average')) for i in range(0, 24, 6)]
    pb = 'g' if graph_based else 'n'
    ins, ls = get_inout_dims('state', 3, 1, 2, pb, d, [12])
    ino, lo = get_inout_dims('output', 3, 1, 2, pb, d, [])
    cls = GNNgraphBased if graph_based else GNNnodeBased
    gnn = cls(MLP(input_dim=ins, layers=ls, activations=['tanh'] * len(ls), kernel_initializer='lecun_normal', bias_initializer='lecun_normal'),
              MLP(input_dim=ino, layers=lo, activations=['softmax'], kernel_initializer='glorot_normal', bias_initializer='glorot_normal', batch_normalization=False),
              optimizers.Adam(1e-3), losses.categorical_crossentropy, {}, d, 12, 0.01, 'c', path_writer='/tmp/gnn_test_eval/', namespace='t')
    gnn.seed = 11
    one_by_one = [gnn.evaluate_single_graph(b, training=False) for b in batches]
    metrics, y_true, y_pred, targets, y_score = gnn.evaluate(batches)
    assert np.array_equal(y_score, np.concatenate([o[3] for o in one_by_one]))
    assert metrics['It'] == int(np.mean(np.asarray([o[0] for o in one_by_one], np.float32)))
    assert abs(metrics['Loss'] - float(np.mean(np.asarray([o[1] for o in one_by_one], np.float32)))) < 1e-6
    assert all(getattr(lp, '_fresh', None) is None for b in batches for (_, lp) in b.device_graph(gnn.device).__dict__.get('_loops', {}).values())


@pytest.mark.parametrize('n,hidden,act,nl,max_it,thr', [(4096, (128, 128), 'selu', 3, 6, 0.0), (333, (128, 128), 'tanh', 3, 5, 0.0), (1000, (128,), 'selu', 3, 4, 0.0),
                                                     (40_000, (128, 128), 'relu', 5, 8, 0.001), (100_003, (128, 128), 'selu', 3, 12, 0.01),
                                                     (65_552, (128,), 'sigmoid', 3, 3, 0.0), (17, (128, 128), 'selu', 3, 3, 0.0)])
def test_tile_forms_of_the_default_path_are_bit_identical(n, hidden, act, nl, max_it, thr):
    """gnn_loop_set_tile_form: one wave per 32-node tile (k_fused) and a wave pair per tile (k_fused_pair: each wave gathers 16 nodes and
    produces half of every layer's output features, activations cross as bf16 pieces through LDS) evaluate the same arithmetic per node
    (reference GNN/GNN.py:223-242): k, states and outputs identical bit for bit - full tiles, a partial last tile (333, 1000, 100,003),
    a range shorter than one side of a tile (17), two- and three-layer nets, wider label blocks (NL 5: concat 139 + hole)."""
    from GNN import GNN_utils as utils
    e = _engine()
    s = utils.syntheticGraph(n, 10.0 if n > 100 else 3.0, nl, 1, 2, seed=n)
    n = s['n_nodes']
    rng = np.random.default_rng(n)
    st = make_mlp(rng, 1 + 2 * (nl + 64), list(hidden) + [64], act, gain=0.6 if thr else 1.0, bn_random=True)
    ou = make_mlp(rng, nl + 64, [2], 'softmax', bn_random=True)
    s0 = (0.1 * rng.standard_normal((n, 64))).astype(np.float32)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    res = {}
    for form in (1, 2):
        lp = e.Loop(graph, mst, mou, 64, max_it, thr)
        assert lp.set_impl(2) == 2
        assert lp.set_tile_form(form) == form
        lp.set_state0(s0)
        k = lp.run()
        res[form] = (k, lp.state(), lp.output(), lp.gate_info()[0])
        assert lp.run() == k                     # and again on the same handle
        assert np.array_equal(lp.state(), res[form][1])
        lp.close()
    graph.close()
    (k1, s1, o1, r1), (k2, s2, o2, r2) = res[1], res[2]
    assert k1 == k2 and r1 == r2
    assert not np.isnan(s2).any()
    assert np.array_equal(s1, s2) and np.array_equal(o1, o2), f'{int(np.sum(s1 != s2))} of {s1.size} state values differ'
