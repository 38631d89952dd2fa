"""The engine's DEFAULT arithmetic (impl 2: fp32 operands as three exact bf16 pieces on the bf16 MFMA) through the facade and
at the depth / shapes of the BASELINE configurations, next to the bit-exact path (impl 1).

Bars (BASELINE.json north_star: "node states/outputs within 1e-5 fp32"):
  * facade objects with their default `impl` (GNNnodeBased / graphBased / edgeBased, LGNN): same iteration counts as the oracle,
    states and outputs within 1e-5 of the float64 oracle (contractive maps);
  * a configs[4]-shaped stack (LGNN x5, state_dim 64, 139 -> 128 -> 128 -> 64 for layers > 0): impl 1 bit-identical to the C
    oracle layer by layer, impl 2 within 1e-5 of float64;
  * configs[2] depth (30 bodies, bench weights, non-contractive): the distance of impl 2 to float64 must not exceed 1.5 x the
    distance of the exact fp32 chain (impl 1) to float64 - "fp32-noise-equivalent" measured, not argued;
  * starter.py (configs[0]) imports and runs under pytest.
Reference call sites: GNN/GNN.py:251-280, :286-302, :318-333; GNN/LGNN.py:227-290; starter.py:135-194."""
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import c_oracle as corc
from oracle import gnn_oracle as orc
from util import make_mlp, random_arcs
from test_gpu_parity import GOLD, _models

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _with_impl(model, impl):
    if impl is not None:
        model.impl = impl
    return model


@pytest.mark.parametrize('impl', [1, 2, None])          # None: whatever the class defaults to (must be the engine default, 2)
def test_mutag_batch_graph_based_default_arithmetic(impl):
    import load_MUTAG
    from GNN.GNN import GNNgraphBased
    from GNN.LGNN import LGNN
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(31)
    graphs = load_MUTAG.load(limit=64)
    st, ou = make_mlp(rng, 3 + 2 * 14, [32, 32, 14], 'selu', gain=0.7), make_mlp(rng, 14, [2], 'softmax')
    gnn = _models(st, ou, 0, 50, 0.01, GNNgraphBased)
    if impl is None:                                  # what a user gets without touching `impl`: the constructor's default
        gnn = gnn.copy(copy_weights=True)
        assert gnn.impl == 2
    else:
        gnn.impl = impl
    for b in range(2):
        batch = GraphObject.merge(graphs[32 * b:32 * b + 32], problem_based='g', aggregation_mode='average')
        k, s, o = gnn.Loop(batch)
        gd = orc.make_graph_dict(batch.arcs, batch.nodes, 'average', NodeGraph=batch.NodeGraph)
        k64, s64, o64 = orc.loop_graph(gd, st, ou, 0, 50, 0.01, dtype=np.float64)
        assert k == k64
        assert np.max(np.abs(s - s64)) < 1e-5 and np.max(np.abs(o - o64)) < 1e-5
    # graph-based LGNN, 2 layers, outputs propagated
    ins, ls = orc.get_inout_dims('state', 14, 3, 2, 'g', 0, [32], layer=1, get_output=True)
    ino, lo = orc.get_inout_dims('output', 14, 3, 2, 'g', 0, None, layer=1, get_output=True)
    st1, ou1 = make_mlp(rng, ins, ls, 'selu', gain=0.7), make_mlp(rng, ino, lo, 'softmax')
    models = [_with_impl(_models(st, ou, 0, 50, 0.01, GNNgraphBased), impl or 2), _with_impl(_models(st1, ou1, 0, 50, 0.01, GNNgraphBased), impl or 2)]
    lgnn = LGNN(models, False, True, None, None, None, 'c')
    K, state, outs = lgnn.Loop(batch)
    gnns = [dict(net_state=st, net_output=ou, state_vect_dim=0, max_iteration=50, threshold=0.01),
            dict(net_state=st1, net_output=ou1, state_vect_dim=0, max_iteration=50, threshold=0.01)]
    K64, s64, o64 = orc.lgnn_loop(gd, gnns, False, True, True, None, np.float64)
    assert K == K64 and all(np.max(np.abs(a - b)) < 1e-5 for a, b in zip(outs, o64))


@pytest.mark.parametrize('impl', [1, 2])
@pytest.mark.parametrize('get_state,get_output', [(False, True), (True, True), (True, False)])
def test_lgnn_node_stack_default_arithmetic(impl, get_state, get_output):
    from GNN.GNN import GNNnodeBased
    from GNN.LGNN import LGNN
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(6)
    arcs = random_arcs(rng, 400, 1200, 1)
    nodes = (2 * rng.random((400, 3)) - 1).astype(np.float32)
    set_mask = rng.random(400) < 0.75
    go = GraphObject(arcs=arcs, nodes=nodes, targets=np.zeros((400, 2)), set_mask=set_mask)
    gd = orc.make_graph_dict(arcs, nodes, 'average', set_mask=set_mask)
    gnns, models = [], []
    for layer in range(3):
        ins, ls = orc.get_inout_dims('state', 3, 1, 2, 'n', 8, [16], layer=layer, get_state=get_state, get_output=get_output)
        ino, lo = orc.get_inout_dims('output', 3, 1, 2, 'n', 8, None, layer=layer, get_state=get_state, get_output=get_output)
        st, ou = make_mlp(rng, ins, ls, 'selu', gain=0.5), make_mlp(rng, ino, lo, 'softmax')
        gnns.append(dict(net_state=st, net_output=ou, state_vect_dim=8, max_iteration=12, threshold=0.01))
        models.append(_with_impl(_models(st, ou, 8, 12, 0.01, GNNnodeBased), impl))
    s0s = [(0.1 * rng.standard_normal((400, 8))).astype(np.float32) for _ in range(3)]
    K, state, outs = LGNN(models, get_state, get_output, None, None, None, 'c').Loop(go, state0=s0s)
    K64, s64, o64 = orc.lgnn_loop(gd, gnns, get_state, get_output, False, s0s, np.float64)
    assert K == K64 and np.max(np.abs(state - s64)) < 1e-5
    assert all(np.max(np.abs(a - b)) < 1e-5 for a, b in zip(outs, o64))


@pytest.mark.parametrize('impl', [1, 2])
@pytest.mark.parametrize('d', [0, 6])
def test_edge_based_default_arithmetic(impl, d):
    from GNN.GNN import GNNedgeBased
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(40 + d)
    n, nl, al = 300, 3, 2
    arcs = random_arcs(rng, n, 900, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    e = len(arcs)
    set_mask, output_mask = rng.random(e) < 0.8, rng.random(e) < 0.7
    go = GraphObject(arcs=arcs, nodes=nodes, targets=np.zeros((int(output_mask.sum()), 2)), problem_based='a', set_mask=set_mask, output_mask=output_mask)
    ins, ls = orc.get_inout_dims('state', nl, al, 2, 'a', d, [12])
    ino, lo = orc.get_inout_dims('output', nl, al, 2, 'a', d, None)
    ds, nlc = (d if d else nl), (nl if d else 0)
    st, ou = make_mlp(rng, ins, ls, 'tanh', gain=0.6), make_mlp(rng, ino, lo, 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32) if d else None
    gnn = _with_impl(_models(st, ou, d, 20, 0.01, GNNedgeBased), impl)
    k, s, o = gnn.Loop(go, state0=s0)
    gd = orc.make_graph_dict(arcs, nodes, 'average')
    node_out = make_mlp(rng, ds + nlc, [2], 'softmax')
    k64, s64, _ = orc.loop_node(gd, st, node_out, d, 20, 0.01, s0, np.float64)
    gd['set_mask'], gd['output_mask'] = set_mask, output_mask
    o64 = orc.mlp_forward(orc.edge_features(gd, s64, d, np.float64), ou['weights'], ou['activations'], True, np.float64)
    assert k == k64 and np.max(np.abs(s - s64)) < 1e-5 and o.shape == o64.shape and np.max(np.abs(o - o64)) < 1e-5


@pytest.mark.parametrize('impl', [1, 2])
def test_reference_fixture_graphs_default_arithmetic(impl):
    from GNN.GNN import GNNnodeBased
    from GNN.graph_class import GraphObject
    for prefix, mode in [('simple/average/n', 'average'), ('random/3', 'average'), ('merge_n/normalized', 'normalized')]:
        rng = np.random.default_rng(2)
        arcs, nodes = GOLD[f'{prefix}/arcs'], GOLD[f'{prefix}/nodes']
        go = GraphObject(arcs=arcs, nodes=nodes, targets=GOLD[f'{prefix}/targets'], aggregation_mode=mode)
        nl, al = go.DIM_NODE_LABEL, go.DIM_ARC_LABEL
        scale = 1.0 / max(1.0, float(np.abs(nodes).max()))
        st = make_mlp(rng, al + 2 * nl, [6, nl], 'tanh', gain=0.5 * scale)
        ou = make_mlp(rng, nl, [2], 'softmax')
        gnn = _with_impl(_models(st, ou, 0, 20, 0.01, GNNnodeBased), impl)
        k, s, o = gnn.Loop(go)
        k64, s64, o64 = orc.loop_node(orc.make_graph_dict(arcs, nodes, mode), st, ou, 0, 20, 0.01, None, np.float64)
        tol = 1e-5 * max(1.0, float(np.abs(s64).max()))          # unnormalised labels (simple_graph: 11 .. 24)
        assert k == k64 and np.max(np.abs(s - s64)) < tol and np.max(np.abs(o - o64)) < 1e-5


def test_c5_shaped_lgnn_stack():
    """BASELINE configs[4] shape on a 4,133-node graph: 5 GNN layers, state_dim 64, get_state=False / get_output=True
    (starter.py:78-79), so layers > 0 see NL' = 3 + 2 = 5 labels: net_state 139 -> 128 -> 128 -> 64, net_output 69 -> 2."""
    from GNN.GNN import GNNnodeBased
    from GNN.LGNN import LGNN
    from GNN.graph_class import GraphObject
    rng = np.random.default_rng(55)
    n, d, layers = 4133, 64, 5
    arcs = random_arcs(rng, n, 5 * n, 1)
    nodes = (2 * rng.random((n, 3)) - 1).astype(np.float32)
    go = GraphObject(arcs=arcs, nodes=nodes, targets=np.zeros((n, 2)))
    gd = orc.make_graph_dict(arcs, nodes, 'average')
    gnns, nets = [], []
    for layer in range(layers):
        ins, ls = orc.get_inout_dims('state', 3, 1, 2, 'n', d, [128, 128], layer=layer, get_state=False, get_output=True)
        ino, lo = orc.get_inout_dims('output', 3, 1, 2, 'n', d, None, layer=layer, get_state=False, get_output=True)
        assert (ins, ino) == ((135, 67) if layer == 0 else (139, 69)) and ls == [128, 128, 64]
        st, ou = make_mlp(rng, ins, ls, 'selu', gain=0.6), make_mlp(rng, ino, lo, 'softmax')
        gnns.append(dict(net_state=st, net_output=ou, state_vect_dim=d, max_iteration=30, threshold=0.01))
        nets.append((st, ou))
    s0s = [(0.1 * rng.standard_normal((n, d))).astype(np.float32) for _ in range(layers)]
    # impl 1: bit-identical to the C oracle, layer by layer, with the reference's relabelling rule between the layers
    lgnn = LGNN([_with_impl(_models(st, ou, d, 30, 0.01, GNNnodeBased), 1) for st, ou in nets], False, True, None, None, None, 'c')
    K, state, outs = lgnn.Loop(go, state0=s0s)
    gtmp, Kc = dict(gd), []
    for (st, ou), s0, got in zip(nets, s0s, outs):
        kc, sc, oc = corc.loop_node(gtmp, st, ou, d, 30, 0.01, s0)
        Kc.append(kc)
        assert np.array_equal(got, oc)
        gtmp = orc.update_graph(gd, sc, oc, False, True)
    assert K == Kc and np.array_equal(state, sc) and all(1 < k < 30 for k in K)
    # impl 2 (the default): same iteration counts, within 1e-5 of float64
    lgnn2 = LGNN([_with_impl(_models(st, ou, d, 30, 0.01, GNNnodeBased), 2) for st, ou in nets], False, True, None, None, None, 'c')
    K2, state2, outs2 = lgnn2.Loop(go, state0=s0s)
    K64, s64, o64 = orc.lgnn_loop(gd, gnns, False, True, False, s0s, np.float64)
    assert K2 == K64 == K
    diffs = [float(np.max(np.abs(a - b))) for a, b in zip(outs2, o64)] + [float(np.max(np.abs(state2 - s64)))]
    if not all(x < 1e-5 for x in diffs):
        # diagnosis for an intermittent failure seen once (normal values: 1.5e-7 per layer, 4.9e-7 for the state): is a second run of the
        # same objects identical (state of the process) or back to normal (a transient)?
        K3, state3, outs3 = lgnn2.Loop(go, state0=s0s)
        again = [float(np.max(np.abs(a - b))) for a, b in zip(outs3, o64)] + [float(np.max(np.abs(state3 - s64)))]
        raise AssertionError(f'impl 2 vs float64: per-layer output differences + state {diffs}; second run {again}, '
                             f'bit-identical to the first: {bool(np.array_equal(state2, state3))}; k {K2} / {K3}')


def test_config3_depth_fp32_noise_equivalence():
    """configs[2] depth: 30 bodies with the bench's random-init (non-contractive) weights on a 50,000-node graph of the bench
    generator.  Two fp32 evaluation orders diverge by ~1e-5 here (tests/test_oracle.py::test_loop_f32_f64_c_agree), so the bar is
    relative: the default path may be at most 1.5 x as far from float64 as the exact fp32 chain is."""
    sys.path.insert(0, ROOT)
    import bench
    from GNN import _engine as e, GNN_utils as utils
    n, d = 50_000, 64
    s = utils.syntheticGraph(n, 10.0, 3, 1, 2, seed=20261003)
    rng = np.random.default_rng(20261003)
    st = bench.make_net(rng, 1 + 2 * (3 + d), [128, 128, d], 'selu')
    ou = bench.make_net(rng, 3 + d, [2], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    arcs = np.concatenate([np.stack([s['src'], s['dst']], 1).astype(np.float32), s['arc_labels']], axis=1)
    g = dict(nodes=s['nodes'], arcs=arcs, set_mask=np.ones(n, bool), output_mask=np.ones(n, bool),
             adjT=(s['indptr'], s['adj_src'], s['adj_w']), arcT=(s['indptr'], s['arc_perm'], s['arc_w']))
    k64, s64, o64 = orc.loop_node(g, st, ou, d, 30, 0.0, s0, np.float64)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    dist = {}
    for impl in (1, 2):
        loop = e.Loop(graph, mst, mou, d, 30, 0.0)
        assert loop.set_impl(impl) == impl
        loop.set_state0(s0)
        assert loop.run() == k64 == 30
        dist[impl] = (float(np.max(np.abs(loop.state() - s64))), float(np.max(np.abs(loop.output() - o64))))
        loop.close()
    print(f'max |state - float64| after 30 bodies: exact fp32 chain {dist[1][0]:.3e}, split bf16 {dist[2][0]:.3e}; outputs {dist[1][1]:.3e} / {dist[2][1]:.3e}')
    # (outputs: both errors sit at a few fp32 ulps of the 2-class head and swap order from run to run: absolute slack of 5e-6)
    assert dist[2][0] <= 1.5 * dist[1][0] and dist[2][1] <= max(1.5 * dist[1][1], 5e-6), dist


def test_starter_drop_in_runs():
    """BASELINE configs[0]: the TensorFlow-free starter (reference starter.py:135-194 names: graphs, gTr, gVa, gTe, gnn, lgnn)
    imports, tests, trains one epoch and tests again, on the random-graph node-focused problem."""
    code = ("import os, sys, numpy as np\n"
            "sys.path.insert(0, os.path.join(%r, 'gnn_tf_2.x_amd')); os.chdir(os.path.join(%r, 'gnn_tf_2.x_amd'))\n"
            "import importlib.util\n"
            "src = open('starter.py').read().replace('use_MUTAG: bool = True', 'use_MUTAG: bool = False').replace('seed: Optional[int] = None', 'seed: Optional[int] = 7')\n"
            "np.random.seed(7)\n"
            "ns = {'__name__': 'starter_under_test', '__file__': os.path.abspath('starter.py')}\n"
            "exec(compile(src, 'starter.py', 'exec'), ns)\n"
            "gnn, lgnn, gTr, gVa, gTe = ns['gnn'], ns['lgnn'], ns['gTr'], ns['gVa'], ns['gTe']\n"
            "assert ns['problem_based'] == 'n' and len(ns['graphs']) == 100\n"
            "m0 = gnn.test(gTe)\n"
            "gnn.train(gTr, 1, gVa, update_freq=1, verbose=0)\n"
            "m1 = gnn.test(gTe); l1 = lgnn.test(gTe)\n"
            "k, state, out = gnn.Loop(gTe)\n"
            "assert out.shape[1] == 2 and np.isfinite(out).all() and np.isfinite(m1['Loss']) and np.isfinite(l1['Loss'])\n"
            "K, s, outs = lgnn.Loop(gTe); assert len(K) == 5 and len(outs) == 5\n"
            "print('STARTER_OK', m0['It'], m1['It'])\n") % (ROOT, ROOT)
    r = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'STARTER_OK' in r.stdout, r.stdout[-2000:] + r.stderr[-4000:]
