"""Pins the oracle: (1) graph half against fixtures emitted by the reference's own graph code
(tests/golden/make_fixtures.py), (2) the SURVEY.md 8c known answers, (3) NumPy-f32 vs NumPy-f64 vs plain-C agreement
on the TF half, (4) hand-derived loop cases.  CPU only."""
import os

import numpy as np
import pytest

from oracle import gnn_oracle as orc
from oracle import c_oracle as corc
from util import make_mlp, random_arcs

GOLD = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'graph_fixtures.npz'))


def _prefixes():
    return sorted({k.rsplit('/', 1)[0] for k in GOLD.files if k.endswith('/arcs')})


def _mode_of(prefix):
    for m in ('sum', 'normalized'):
        if f'/{m}' in prefix:
            return m
    return 'average'


@pytest.mark.parametrize('prefix', _prefixes())
def test_graph_matrices_match_reference(prefix):
    arcs, nodes = GOLD[f'{prefix}/arcs'], GOLD[f'{prefix}/nodes']
    mode = _mode_of(prefix)
    n = nodes.shape[0]
    w = orc.arcnode_values(arcs, mode)
    # ArcNode: entry (arc a, dst(a)) = w_a, stored in arc order by the reference (graph_class.py:121)
    assert np.array_equal(GOLD[f'{prefix}/ArcNode_row'], np.arange(len(arcs)))
    assert np.array_equal(GOLD[f'{prefix}/ArcNode_col'], arcs[:, 1].astype(int))
    assert np.array_equal(GOLD[f'{prefix}/ArcNode_data'], w)                       # bit-exact float32
    assert np.array_equal(GOLD[f'{prefix}/Adj_data'], w)
    # transposed CSR == dense Adjacency^T
    (ip, src, val), (ip2, aid, val2) = orc.graph_matrices(arcs, n, mode)
    dense_T = np.zeros((n, n), dtype=np.float32)
    for r in range(n):
        assert np.all(np.diff(src[ip[r]:ip[r + 1]]) >= 0)                          # row-major reorder: ascending src
        assert np.all(np.diff(aid[ip2[r]:ip2[r + 1]]) > 0)                         # ascending arc id
        for e in range(ip[r], ip[r + 1]):
            dense_T[r, src[e]] += val[e]
    assert np.array_equal(dense_T, GOLD[f'{prefix}/Adj_dense'].T)
    assert np.array_equal(ip, ip2)
    assert np.array_equal(arcs[aid, 1].astype(int), np.repeat(np.arange(n), np.diff(ip)))
    # loop-invariant aggregates (float64 SciPy products of the reference-built matrices)
    g = orc.make_graph_dict(arcs, nodes, mode)
    agg_nodes = orc.spmm_csr(g['adjT'], nodes, np.float64)
    agg_arcs = orc.spmm_csr(g['arcT'], arcs[:, 2:], np.float64)
    np.testing.assert_allclose(agg_nodes, GOLD[f'{prefix}/AdjT_nodes'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(agg_arcs, GOLD[f'{prefix}/ArcNodeT_arclabels'], rtol=1e-12, atol=1e-12)
    # ... and the C restatement in float32
    np.testing.assert_allclose(corc.spmm(g['adjT'], nodes), GOLD[f'{prefix}/AdjT_nodes'], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(corc.spmm(g['arcT'], arcs[:, 2:]), GOLD[f'{prefix}/ArcNodeT_arclabels'], rtol=2e-6, atol=1e-6)


def test_simple_graph_known_answers():
    """values computed through the reference's GraphObject on GNN_utils.simple_graph."""
    p = 'simple/average/n'
    arcs, nodes = GOLD[f'{p}/arcs'], GOLD[f'{p}/nodes']
    assert np.array_equal(nodes, [[11, 21], [12, 22], [13, 23], [14, 24]])
    w = orc.arcnode_values(arcs, 'average')
    np.testing.assert_allclose(w, [.5, 1 / 3, .5, 1 / 3, .5, .5, 1, 1 / 3], rtol=1e-7)
    (ip, src, val), (_, aid, _) = orc.graph_matrices(arcs, 4, 'average')
    assert ip.tolist() == [0, 2, 4, 7, 8]
    assert src.tolist() == [1, 2, 0, 2, 0, 1, 3, 2]
    assert aid.tolist() == [2, 4, 0, 5, 1, 3, 7, 6]
    np.testing.assert_allclose(val, [.5, .5, .5, .5, 1 / 3, 1 / 3, 1 / 3, 1], rtol=1e-7)
    g = orc.make_graph_dict(arcs, nodes)
    np.testing.assert_allclose(orc.spmm_csr(g['adjT'], nodes), [[12.5, 22.5], [12, 22], [12.33333, 22.33333], [13, 23]], rtol=1e-6)
    np.testing.assert_allclose(orc.spmm_csr(g['arcT'], arcs[:, 2:]).ravel(), [25, 15, 30, 30], rtol=1e-6)
    assert np.all(orc.arcnode_values(arcs, 'sum') == 1)
    assert np.all(orc.arcnode_values(arcs, 'normalized') == np.float32(0.125))
    # first condition call on D == 0: distances of nodes from ones vs 0.01 * sqrt(2)
    dist = np.sqrt(np.sum((nodes - 1) ** 2, axis=1))
    np.testing.assert_allclose(dist, [22.36, 23.71, 25.06, 26.42], atol=0.01)
    assert orc.not_converged(nodes, np.ones_like(nodes), 0.01).all()
    # graph based: NodeGraph = 0.25 * ones(4,1); merge of two copies -> [8, 2] block diagonal, ids of 2nd copy + 4
    assert np.array_equal(GOLD['simple/average/g/NodeGraph'], np.full((4, 1), 0.25, np.float32))
    assert np.array_equal(orc.nodegraph_single(4), GOLD['simple/average/g/NodeGraph'])
    m = GOLD['merge_simple2/NodeGraph']
    assert m.shape == (8, 2) and np.all(m[:4, 0] == .25) and np.all(m[4:, 1] == .25) and m.sum() == 2
    assert np.array_equal(GOLD['merge_simple2/arcs'][8:, :2], arcs[:, :2] + 4)
    assert GOLD['merge_simple2/targets'].shape == (2, 2)


def _gdict(prefix):
    d = {k: GOLD[f'{prefix}/{k}'] for k in ('arcs', 'nodes', 'targets', 'set_mask', 'output_mask', 'sample_weights')}
    d['NodeGraph'] = GOLD[f'{prefix}/NodeGraph'] if f'{prefix}/NodeGraph' in GOLD.files else None
    return d


def test_merge_matches_reference():
    for pb, parts, merged in [('n', [f'random/{i}' for i in (1, 2, 3)], 'merge_n/average'),
                              ('g', [f'gsingle/{i}' for i in range(3)], 'merge_g/average')]:
        m = orc.merge_graphs([_gdict(p) for p in parts], pb)
        for k in ('arcs', 'nodes', 'targets', 'set_mask', 'output_mask', 'sample_weights'):
            assert np.array_equal(m[k], GOLD[f'{merged}/{k}']), k
        if pb == 'g':
            assert np.array_equal(m['NodeGraph'], GOLD[f'{merged}/NodeGraph'])
    # 'normalized' weight after merging is 1 / E_merged (matrices are rebuilt on the merged arcs, graph_class.py:318)
    e = len(GOLD['merge_n/normalized/arcs'])
    assert np.all(GOLD['merge_n/normalized/ArcNode_data'] == np.float32(1 / e))
    shapes = GOLD['getbatches/shapes']
    sizes = GOLD['getbatches/sizes']
    assert shapes[0].tolist() == sizes[:32].sum(0).tolist() and shapes[2].tolist() == sizes[64:].sum(0).tolist()


def test_get_inout_dims():
    """GNN/MLP.py:68-122 on the BASELINE.json configs (SURVEY.md section 8 header)."""
    assert orc.get_inout_dims('state', 3, 1, 2, 'n', 0, None) == (7, [3])
    assert orc.get_inout_dims('output', 3, 1, 2, 'n', 0, None) == (3, [2])
    assert orc.get_inout_dims('state', 14, 3, 2, 'g', 0, [32, 32]) == (31, [32, 32, 14])
    assert orc.get_inout_dims('state', 3, 1, 2, 'n', 64, [128, 128]) == (135, [128, 128, 64])
    assert orc.get_inout_dims('output', 3, 1, 2, 'n', 64, None) == (67, [2])
    assert orc.get_inout_dims('state', 3, 1, 2, 'n', 64, [128, 128], layer=2, get_state=False, get_output=True) == (139, [128, 128, 64])
    assert orc.get_inout_dims('output', 3, 1, 2, 'n', 64, 0, layer=1, get_output=True) == (69, [2])
    assert orc.get_inout_dims('state', 3, 1, 2, 'n', 0, 5, layer=2, get_state=True, get_output=True) == (1 + 2 * (3 + 6 + 4), [5, 13])
    assert orc.get_inout_dims('output', 3, 2, 4, 'a', 5, None) == (3 + 2 + 5 + 3 + 5, [4])
    with pytest.raises(ValueError):
        orc.get_inout_dims('foo', 1, 1, 1, 'n', 0, None)


def test_expf_accuracy():
    x = np.concatenate([np.linspace(-87, 88, 4001), np.linspace(-1, 1, 2001), [0.0, -0.0, 1e-8, -1e-8]]).astype(np.float32)
    got = corc.expf(x).astype(np.float64)
    ref = np.exp(x.astype(np.float64))
    assert np.all(np.abs(got - ref) / ref < 2.5e-7 + 7e-8 * np.abs(x))     # polynomial 2e-7 + argument rounding 6e-8 |x|
    assert corc.expf(np.float32([-100]))[0] == 0 and np.isinf(corc.expf(np.float32([89]))[0])
    assert corc.expf(np.float32([0]))[0] == 1.0


@pytest.mark.parametrize('act', ['linear', 'relu', 'selu', 'elu', 'tanh', 'sigmoid', 'softmax'])
def test_mlp_c_vs_numpy(act):
    rng = np.random.default_rng(5)
    net = make_mlp(rng, 19, [33, 7], act, batch_normalization=True, bn_random=True)
    x = rng.standard_normal((301, 19)).astype(np.float32) * 2
    y64 = orc.mlp_forward(x, net['weights'], net['activations'], True, np.float64)
    y32 = orc.mlp_forward(x, net['weights'], net['activations'], True, np.float32)
    yc = corc.mlp_forward(x, net['weights'], net['activations'], True)
    assert np.max(np.abs(y32 - y64)) < 2e-5
    assert np.max(np.abs(yc - y64)) < 2e-5


def _small_case(rng, n=200, d=8, nl=3, al=2, thr=0.01, sort=True, gain=0.6):
    arcs = random_arcs(rng, n, 3 * n, al, sort=sort)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    g = orc.make_graph_dict(arcs, nodes, 'average')
    ds = d if d else nl
    nls = nl if d else 0
    st = make_mlp(rng, al + 2 * (ds + nls), [16, ds], 'selu', gain=gain)
    ou = make_mlp(rng, ds + nls, [2], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, ds))).astype(np.float32) if d else None
    return g, st, ou, s0


@pytest.mark.parametrize('d,gain,tol', [(0, 0.6, 1e-5), (8, 0.6, 1e-5), (0, 1.0, 1e-3), (8, 1.0, 1e-3)])
def test_loop_f32_f64_c_agree(d, gain, tol):
    """gain 0.6: contractive state map -> every float32 evaluation order stays within 1e-5 of the float64 shadow.
    gain 1.0 (lecun_normal as in starter.py:52): the map is not a contraction, rounding differences are amplified from
    iteration to iteration (3e-5 seen after 30 iterations on 3-wide states), so only a loose bound is meaningful
    against float64; bit-exactness between the C oracle and the GPU is what pins that regime."""
    rng = np.random.default_rng(11 + d)
    g, st, ou, s0 = _small_case(rng, d=d, gain=gain)
    k64, s64, o64 = orc.loop_node(g, st, ou, d, 30, 0.01, s0, np.float64)
    k32, s32, o32 = orc.loop_node(g, st, ou, d, 30, 0.01, s0, np.float32)
    kc, sc, oc = corc.loop_node(g, st, ou, d, 30, 0.01, s0)
    assert k64 == k32 == kc and 1 < kc <= 30
    for a in (s32, sc):
        assert np.max(np.abs(a - s64)) < tol
    for a in (o32, oc):
        assert np.max(np.abs(a - o64)) < tol


@pytest.mark.parametrize('d,act', [(0, 'selu'), (5, 'tanh'), (7, 'sigmoid'), (4, 'elu')])
def test_c_float64_shadow_equals_numpy_float64(d, act):
    """oracle/gnn_oracle_f64.c (the arbiter of the full-size GPU tests, where the NumPy shadow would take minutes) against
    gnn_oracle.loop_node(dtype=float64): same k, states and outputs to 1e-12 - masks, unsorted arcs, every activation."""
    rng = np.random.default_rng(40 + d)
    g, st, ou, s0 = _small_case(rng, n=180, d=d, sort=False)
    for net in (st,):
        net['activations'] = [act] * len(net['activations'])
    g['set_mask'] = rng.random(180) < 0.7
    k64, s64, o64 = orc.loop_node(g, st, ou, d, 25, 0.01, s0, np.float64)
    kc, sc, oc = corc.loop_node_f64(g, st, ou, d, 25, 0.01, s0)
    assert kc == k64 and sc.dtype == np.float64
    assert np.max(np.abs(sc - s64)) < 1e-12 and np.max(np.abs(oc - o64)) < 1e-12


def test_loop_unsorted_arcs_and_masks():
    rng = np.random.default_rng(3)
    g, st, ou, s0 = _small_case(rng, n=150, d=5, sort=False)
    g['set_mask'] = rng.random(150) < 0.7
    g['output_mask'] = rng.random(150) < 0.6
    k64, s64, o64 = orc.loop_node(g, st, ou, 5, 20, 0.001, s0, np.float64)
    kc, sc, oc = corc.loop_node(g, st, ou, 5, 20, 0.001, s0)
    assert kc == k64
    assert oc.shape[0] == int(np.sum(g['set_mask'] & g['output_mask']))
    assert np.max(np.abs(sc - s64)) < 1e-5 and np.max(np.abs(oc - o64)) < 1e-5


def test_loop_hand_derived_cases():
    rng = np.random.default_rng(0)
    g, st, ou, s0 = _small_case(rng, n=50, d=4)
    # zero-weight net_state: state becomes 0 after one step, second condition compares 0 with 0 -> k == 2
    zero = dict(st, weights=[np.zeros_like(w) for w in st['weights'][:-4]] + st['weights'][-4:])
    for impl in (orc.loop_node, corc.loop_node):
        k, s, _ = impl(g, zero, ou, 4, 30, 0.01, s0)
        assert k == 2 and np.all(s == 0)
        # threshold 0 and a non-stationary map: strict '>' keeps iterating until max_iteration
        k, _, _ = impl(g, st, ou, 4, 7, 0.0, s0)
        assert k == 7
        # max_iteration 0: the body never runs, state is the injected one
        k, s, _ = impl(g, st, ou, 4, 0, 0.01, s0)
        assert k == 0 and np.array_equal(s, s0)
    # isolated destination -> empty row -> aggregate exactly 0 (graph_class.py:120 comment)
    arcs = np.array([[0, 1, .5], [1, 0, .5]], dtype=np.float32)
    nodes = rng.random((3, 2)).astype(np.float32)
    gi = orc.make_graph_dict(arcs, nodes)
    assert np.all(orc.spmm_csr(gi['adjT'], nodes)[2] == 0) and np.all(corc.spmm(gi['adjT'], nodes)[2] == 0)
    # state == ones at entry: first condition false, k == 0
    ones = np.ones((3, 2), np.float32)
    g1 = orc.make_graph_dict(arcs, ones)
    st1 = make_mlp(rng, 1 + 2 * 2, [2], 'linear')
    ou1 = make_mlp(rng, 2, [2], 'softmax')
    for impl in (orc.loop_node, corc.loop_node):
        assert impl(g1, st1, ou1, 0, 5, 0.01)[0] == 0


def test_graph_based_and_lgnn_oracle():
    rng = np.random.default_rng(21)
    parts = [_gdict(f'gsingle/{i}') for i in range(3)]
    m = orc.merge_graphs(parts, 'g')
    g = orc.make_graph_dict(m['arcs'], m['nodes'], 'average', NodeGraph=m['NodeGraph'], targets=m['targets'])
    nl, al, t = 2, 2, 2
    st = make_mlp(rng, al + 2 * nl, [6, nl], 'selu')
    ou = make_mlp(rng, nl, [t], 'softmax')
    k, s, o = orc.loop_graph(g, st, ou, 0, 10, 0.01)
    assert o.shape == (3, t)
    _, _, on = orc.loop_node(g, st, ou, 0, 10, 0.01)
    lens = [p['nodes'].shape[0] for p in parts]
    ref = np.stack([on[sum(lens[:i]):sum(lens[:i + 1])].mean(0) for i in range(3)])
    np.testing.assert_allclose(o, ref, atol=1e-6)
    np.testing.assert_allclose(corc.readout(g['NodeGraph'], on), o, atol=1e-6)
    with pytest.raises(ValueError):
        orc.loop_graph(dict(g, NodeGraph=None), st, ou, 0, 10, 0.01)
    # 3-layer LGNN, get_output only: layer>0 node labels widen by T (MLP.py:96-99)
    gn = orc.make_graph_dict(m['arcs'], m['nodes'], 'average')
    gnns = []
    for layer in range(3):
        ins, ls = orc.get_inout_dims('state', nl, al, t, 'n', 3, [5], layer=layer, get_output=True)
        ino, lo = orc.get_inout_dims('output', nl, al, t, 'n', 3, None, layer=layer, get_output=True)
        gnns.append(dict(net_state=make_mlp(rng, ins, ls, 'tanh'), net_output=make_mlp(rng, ino, lo, 'softmax'),
                         state_vect_dim=3, max_iteration=6, threshold=0.01))
    s0s = [(0.1 * rng.standard_normal((gn['nodes'].shape[0], 3))).astype(np.float32) for _ in range(3)]
    ks, state, outs = orc.lgnn_loop(gn, gnns, False, True, False, s0s)
    assert len(ks) == 3 and len(outs) == 3 and state.shape == (gn['nodes'].shape[0], 3)
    assert all(o.shape == (gn['nodes'].shape[0], t) for o in outs)


def test_oracle_against_torch_crosscheck_vectors():
    """tests/golden/torch_crosscheck.npz holds what PyTorch-CPU computed for the TF/Keras ops of the path (generated by
    tests/golden/make_torch_crosscheck.py in the build container).  Not a reference-held pin (the TF half stays
    "parity unpinned"), but an independent implementation of the same semantics: constants, formulas, loop control."""
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'torch_crosscheck.npz'))
    x = z['ops/x']
    for act in ['linear', 'relu', 'selu', 'elu', 'tanh', 'sigmoid', 'softmax']:
        assert np.max(np.abs(orc.activation(x.astype(np.float64), act) - z[f'ops/{act}'])) < 1e-12, act
        got32 = corc.mlp_forward(x, [np.eye(11, dtype=np.float32), np.zeros(11, np.float32)], [act], False)
        assert np.max(np.abs(got32 - z[f'ops/{act}']) / np.maximum(1.0, np.abs(z[f'ops/{act}']))) < 3e-6, act     # C restatement, fp32
    bn = z['ops/bn_params']
    ident = [np.eye(11), np.zeros(11)] + [bn[0], bn[1], bn[2], bn[3]]
    assert np.max(np.abs(orc.mlp_forward(x, ident, ['linear'], True, np.float64) - z['ops/bn'])) < 1e-12
    for name in ['selu_d0', 'tanh_d8', 'sigmoid_d5', 'selu_d16_deep']:
        d, nl, al, max_it, n_hidden = (int(v) for v in z[f'{name}/cfg'])
        act, mode = str(z[f'{name}/act']), str(z[f'{name}/mode'])
        g = orc.make_graph_dict(z[f'{name}/arcs'], z[f'{name}/nodes'], mode)
        g['set_mask'] = z[f'{name}/set_mask']
        n_st = 2 * (n_hidden + 1) + 4
        st = dict(weights=[z[f'{name}/st{i}'] for i in range(n_st)], activations=[act] * (n_hidden + 1), batch_normalization=True)
        ou = dict(weights=[z[f'{name}/ou{i}'] for i in range(6)], activations=['softmax'], batch_normalization=True)
        s0 = z[f'{name}/s0'] if d else None
        k64, s64, o64 = orc.loop_node(g, st, ou, d, max_it, 0.01, s0, np.float64)
        assert k64 == int(z[f'{name}/f64/k']) and 1 < k64 < max_it
        assert np.max(np.abs(s64 - z[f'{name}/f64/state'])) < 1e-12 and np.max(np.abs(o64 - z[f'{name}/f64/out'])) < 1e-12
        for k32, s32, o32 in (orc.loop_node(g, st, ou, d, max_it, 0.01, s0, np.float32), corc.loop_node(g, st, ou, d, max_it, 0.01, s0)):
            assert k32 == int(z[f'{name}/f32/k'])
            assert np.max(np.abs(s32 - z[f'{name}/f32/state'])) < 1e-5 and np.max(np.abs(o32 - z[f'{name}/f32/out'])) < 1e-5
