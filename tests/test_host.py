"""CPU-only checks of the product's host side: the Python mirror of the reference classes against the golden fixtures,
and that the C-ABI library loads and exports every symbol declared in include/gnn_hip.h (no compute calls)."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import gnn_oracle as orc

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = np.load(os.path.join(ROOT, 'tests', 'golden', 'graph_fixtures.npz'))


def _graph_object(prefix, problem_based='n', mode='average'):
    from GNN.graph_class import GraphObject
    kw = {}
    if f'{prefix}/NodeGraph' in GOLD.files:
        kw['NodeGraph'] = GOLD[f'{prefix}/NodeGraph']
    return GraphObject(arcs=GOLD[f'{prefix}/arcs'], nodes=GOLD[f'{prefix}/nodes'], targets=GOLD[f'{prefix}/targets'],
                       problem_based=problem_based, set_mask=GOLD[f'{prefix}/set_mask'], output_mask=GOLD[f'{prefix}/output_mask'],
                       sample_weights=GOLD[f'{prefix}/sample_weights'], aggregation_mode=mode, **kw)


@pytest.mark.parametrize('prefix,mode', [('simple/average/n', 'average'), ('simple/sum/n', 'sum'), ('simple/normalized/n', 'normalized'),
                                         ('random/1', 'average'), ('random/5', 'average'), ('merge_n/normalized', 'normalized')])
def test_graphobject_matrices_match_reference(prefix, mode):
    from GNN.graph_class import GraphTensor
    g = _graph_object(prefix, mode=mode)
    an, ad = g.ArcNode.tocoo(), g.Adjacency.tocoo()
    assert np.array_equal(an.row, GOLD[f'{prefix}/ArcNode_row']) and np.array_equal(an.col, GOLD[f'{prefix}/ArcNode_col'])
    assert np.array_equal(an.data, GOLD[f'{prefix}/ArcNode_data'])
    assert np.array_equal(ad.data, GOLD[f'{prefix}/Adj_data']) and np.array_equal(g.Adjacency.toarray(), GOLD[f'{prefix}/Adj_dense'])
    assert g.DIM_NODE_LABEL == g.nodes.shape[1] and g.DIM_ARC_LABEL == g.arcs.shape[1] - 2 and g.DIM_TARGET == g.targets.shape[1]
    gt = GraphTensor.fromGraphObject(g)
    (ip, src, w), (ip2, aid, w2) = orc.graph_matrices(g.arcs, g.nodes.shape[0], mode)
    assert np.array_equal(gt.Adjacency[0], ip) and np.array_equal(gt.Adjacency[1], src) and np.array_equal(gt.Adjacency[2], w)
    assert np.array_equal(gt.ArcNode[0], ip2) and np.array_equal(gt.ArcNode[1], aid) and np.array_equal(gt.ArcNode[2], w2)


def test_graphobject_merge_copy_save_load(tmp_path):
    from GNN.graph_class import GraphObject, GraphTensor
    parts = [_graph_object(f'gsingle/{i}', 'g') for i in range(3)]
    m = GraphObject.merge(parts, problem_based='g', aggregation_mode='average')
    for k in ('arcs', 'nodes', 'targets', 'set_mask', 'output_mask', 'sample_weights', 'NodeGraph'):
        assert np.array_equal(getattr(m, k), GOLD[f'merge_g/average/{k}']), k
    assert np.array_equal(m.ArcNode.tocoo().data, GOLD['merge_g/average/ArcNode_data'])
    c = m.copy()
    assert np.array_equal(c.arcs, m.arcs) and np.array_equal(c.NodeGraph, m.NodeGraph)
    m.save(str(tmp_path / 'g'))
    back = GraphObject.load(str(tmp_path / 'g'), problem_based='g', aggregation_mode='average')
    assert np.array_equal(back.arcs, m.arcs) and np.array_equal(back.NodeGraph, m.NodeGraph) and np.array_equal(back.targets, m.targets)
    ip, node, w = GraphTensor.fromGraphObject(m).nodegraph_csr()
    dense = np.zeros_like(m.NodeGraph)
    for gidx in range(len(ip) - 1):
        dense[node[ip[gidx]:ip[gidx + 1]], gidx] = w[ip[gidx]:ip[gidx + 1]]
    assert np.array_equal(dense, m.NodeGraph)
    with pytest.raises(ValueError):
        GraphObject(arcs=m.arcs, nodes=m.nodes, targets=m.targets, aggregation_mode='bogus')
    with pytest.raises(ValueError):
        GraphObject(arcs=m.arcs, nodes=m.nodes, targets=m.targets, set_mask=np.ones(3), output_mask=np.ones(4))
    with pytest.raises(TypeError):
        GraphObject.merge('nope', 'n', 'sum')


def test_randomgraph_reproduces_reference_stream():
    from GNN import GNN_utils as utils
    for seed, n in [(1, 17), (2, 23), (5, 39)]:
        np.random.seed(seed)
        g = utils.randomGraph(nodes_number=n, dim_node_label=3, dim_arc_label=1, dim_target=2, density=0.7)
        assert np.array_equal(g.arcs, GOLD[f'random/{seed}/arcs']) and np.array_equal(g.nodes, GOLD[f'random/{seed}/nodes'])
        assert np.array_equal(g.targets, GOLD[f'random/{seed}/targets'])
    s = utils.simple_graph('n')
    assert np.array_equal(s.arcs, GOLD['simple/average/n/arcs']) and np.array_equal(s.targets, GOLD['simple/average/n/targets'])
    np.random.seed(20261003)
    many = [utils.randomGraph(int(np.random.choice(range(15, 40))), 3, 1, 2, 0.7) for _ in range(70)]
    batches = utils.getbatches(many, problem_based='n', aggregation_mode='average', batch_size=32)
    assert np.array_equal(np.array([[b.nodes.shape[0], b.arcs.shape[0]] for b in batches]), GOLD['getbatches/shapes'])
    tr, te, va = utils.getindices(70, 0.7, 0.2, seed=3)
    assert len(tr) == 49 and len(te) == 7 and len(va) == 14 and sorted(tr + te + va) == list(range(70))


def test_synthetic_graph_invariants():
    from GNN import GNN_utils as utils
    s = utils.syntheticGraph(5000, 10.0, seed=7)
    n, e = s['n_nodes'], s['n_arcs']
    src, dst = s['src'].astype(np.int64), s['dst'].astype(np.int64)
    assert 0.9 * 50000 < e <= 50000 and np.all(src != dst)
    key = src * n + dst
    assert np.all(np.diff(key) > 0)                                   # lexicographically sorted, duplicate free
    assert np.array_equal(np.sort(dst * n + src), key)                # symmetric
    arcs = np.concatenate([np.stack([src, dst], 1).astype(np.float32), s['arc_labels']], axis=1)
    (ip, a_src, w), (_, aid, _) = orc.graph_matrices(arcs, n, 'average')
    assert np.array_equal(ip, s['indptr']) and np.array_equal(a_src, s['adj_src']) and np.array_equal(aid, s['arc_perm'])
    assert np.array_equal(w, s['adj_w']) and np.array_equal(s['arc_labels'][aid], s['arc_labels_csr'])


def test_mlp_factory_and_dims():
    from GNN.MLP import MLP, get_inout_dims, Dense, Dropout, BatchNormalization, set_seed
    set_seed(0)
    net = MLP(input_dim=7, layers=[5, 3], activations='selu', kernel_initializer='lecun_normal', bias_initializer='lecun_normal',
              dropout_rate=0.1, dropout_pos=0)
    kinds = [type(l) for l in net.layers]
    assert kinds == [Dropout, Dense, Dense, BatchNormalization]        # dropout_pos 0 -> before the first Dense (MLP.py:54-55)
    w = net.get_weights()
    assert [a.shape for a in w] == [(7, 5), (5,), (5, 3), (3,), (3,), (3,), (3,), (3,)]
    assert np.all(w[4] == 1) and np.all(w[5] == 0) and np.all(w[6] == 0) and np.all(w[7] == 1)
    assert np.abs(w[0]).max() <= 2 * np.sqrt(1 / 7) / 0.87962566103423978 + 1e-6
    net2 = MLP(4, [6, 6, 2], ['relu', 'tanh', 'softmax'], 'glorot_normal', 'zeros', dropout_rate=[0.1, 0.2], dropout_pos=[0, 2],
               batch_normalization=False)
    assert [type(l) for l in net2.layers] == [Dropout, Dense, Dense, Dropout, Dense]
    assert net2.activations == ['relu', 'tanh', 'softmax'] and not net2.batch_normalization
    with pytest.raises(ValueError):
        MLP(4, [6, 2], ['relu'], 'zeros', 'zeros')
    for args, kw in [(('state', 3, 1, 2, 'n', 64, [128, 128]), {}), (('output', 3, 1, 2, 'n', 64, None), {}),
                     (('state', 3, 1, 2, 'n', 64, [128, 128]), dict(layer=2, get_output=True)),
                     (('state', 3, 1, 2, 'n', 0, 5), dict(layer=2, get_state=True, get_output=True)),
                     (('output', 3, 2, 4, 'a', 5, None), {}), (('state', 14, 3, 2, 'g', 0, [32, 32]), {})]:
        assert get_inout_dims(*args, **kw) == orc.get_inout_dims(*args, **kw)


def test_model_argument_errors_match_reference():
    from GNN.GNN import GNNnodeBased
    from GNN.LGNN import LGNN
    from GNN.MLP import MLP
    from GNN.GNN_BaseClass import BaseClass
    st = MLP(7, [3], 'selu', 'lecun_normal', 'zeros')
    ou = MLP(3, [2], 'softmax', 'glorot_normal', 'zeros')
    mk = lambda **kw: GNNnodeBased(**{**dict(net_state=st, net_output=ou, optimizer=None, loss_function=None, loss_arguments=None,
                                             state_vect_dim=0, max_iteration=5, threshold=0.01, addressed_problem='c'), **kw})
    with pytest.raises(TypeError):
        mk(state_vect_dim=-1)                       # reference GNN.py:53
    with pytest.raises(ValueError):
        mk(addressed_problem='x')                   # reference GNN_BaseClass.py:40
    with pytest.raises(TypeError):
        mk(extra_metrics=[1])                       # reference GNN_BaseClass.py:41
    with pytest.raises(TypeError):
        BaseClass.checktype(3)                      # reference GNN_BaseClass.py:424
    assert BaseClass.checktype(None) is None
    gnn = mk()
    assert gnn.get_weights()[0][0][0].shape == (7, 3)
    lg = LGNN([gnn], False, True, None, None, None, 'c')
    lg.training_mode = 'parallel'
    with pytest.raises(ValueError):
        lg.train([], 1, training_mode='serial', verbose=0)        # reference LGNN.py:318-319
    assert st.dropout_rates() == [0.0, 0.0] and MLP(7, [5, 3], 'selu', 'zeros', 'zeros', dropout_rate=0.1, dropout_pos=[0, 1]).dropout_rates() == [0.1, 0.1, 0.0]


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, 'include', 'gnn_hip.h')).read()
    declared = sorted(set(re.findall(r'\b(gnn_[a-z0-9_]+)\s*\(', header)))
    assert len(declared) >= 28
    from GNN import _engine
    assert sorted(_engine.EXPORTS) == declared
    so = _engine.LIB_PATH
    if not os.path.exists(so):          # fresh checkout: cross-compile for gfx950 (no GPU needed), exactly what build() does
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'gnn_tf_2.x_amd', 'csrc'), '-j8'], stdout=subprocess.DEVNULL)
    assert os.path.exists(so), 'libgnn_hip.so not built: run __graft_entry__.build()'
    lib = ctypes.CDLL(so)
    for name in declared:
        assert hasattr(lib, name), name
    lib.gnn_version.restype = ctypes.c_int
    assert lib.gnn_version() == 1


def test_mutag_loader_follows_reference_semantics():
    """The loader against a literal (slow, mask-based) restatement of reference load_MUTAG.py:14-52 on the first graphs,
    and the dataset totals of SURVEY.md section 6."""
    import load_MUTAG
    graphs = load_MUTAG.load()
    assert len(graphs) == 4337 and sum(g.nodes.shape[0] for g in graphs) == 131488 and sum(g.arcs.shape[0] for g in graphs) == 266894
    assert graphs[0].DIM_NODE_LABEL == 14 and graphs[0].DIM_ARC_LABEL == 3 and graphs[0].DIM_TARGET == 2
    assert int(sum(g.targets[0, 0] for g in graphs)) == 2401
    edgesIDs, edgesL, nodesL, gIDs_nodes, gtargs = load_MUTAG._raw()
    _, idx = np.unique(gIDs_nodes, return_index=True)
    idx = np.concatenate([idx, [len(gIDs_nodes)]]).tolist()
    edgesIDs = np.unique(edgesIDs, axis=0)
    eL = np.zeros((edgesL.shape[0], len(np.unique(edgesL))), dtype=int)
    eL[range(eL.shape[0]), edgesL] = 1
    for g in range(40):
        i, j = idx[g], idx[g + 1]
        k = (edgesIDs > i) * (edgesIDs <= j)
        mask = (k[:, 0] * k[:, 1]).astype(bool)
        ids = edgesIDs[mask, :].copy()
        for new, old in enumerate(np.unique(ids)):
            ids[ids == old] = new
        arcs = np.concatenate([ids, eL[mask]], axis=1)
        assert np.array_equal(graphs[g].arcs, arcs.astype(np.float32))
        assert graphs[g].nodes.shape == (j - i, 14) and graphs[g].NodeGraph.shape == (j - i, 1)


def test_regularizers_penalty_and_gradient():
    """GNN.regularizers (the tf.keras.regularizers call convention, summed into the loss as reference GNN_BaseClass.py:223-228)."""
    from GNN import regularizers
    from GNN.MLP import MLP
    rng = np.random.default_rng(0)
    net = MLP(5, [4, 3], 'tanh', 'glorot_normal', 'zeros', kernel_regularizer=regularizers.l2(0.1), bias_regularizer=[None, regularizers.l1_l2(0.2, 0.3)],
              batch_normalization=False)
    net.set_weights([rng.standard_normal(w.shape).astype(np.float32) for w in net.get_weights()])
    pen, grads = regularizers.penalty_and_gradients(net.dense_layers)
    w = net.get_weights()
    want = 0.1 * np.sum(w[0].astype(np.float64) ** 2) + 0.1 * np.sum(w[2].astype(np.float64) ** 2) + 0.2 * np.sum(np.abs(w[3])) + 0.3 * np.sum(w[3].astype(np.float64) ** 2)
    assert abs(pen - want) < 1e-6 * max(1.0, abs(want))
    assert grads[0][1] is None and np.allclose(grads[0][0], 0.2 * w[0]) and np.allclose(grads[1][0], 0.2 * w[2])
    assert np.allclose(grads[1][1], 0.2 * np.sign(w[3]) + 0.6 * w[3])
    with pytest.raises(TypeError):
        bad = MLP(5, [2], 'tanh', 'zeros', 'zeros', kernel_regularizer=lambda x: 0.0, batch_normalization=False)
        regularizers.penalty_and_gradients(bad.dense_layers)


def test_halo_plan_on_host():
    """gnn_halo_plan is host-only: block-diagonal batches sharded at graph boundaries have no boundary rows at all, a ring
    has one per cut; shard_halo renumbers sources into [own rows | one block of boundary rows per rank]."""
    from GNN import _engine as e
    n, world = 256, 4                        # shard = 64
    src = np.arange(n, dtype=np.int32)

    def csr(dst):
        order = np.lexsort((src, dst))
        indptr = np.zeros(n + 1, np.int64)
        np.add.at(indptr, dst + 1, 1)
        return np.cumsum(indptr).astype(np.int32), src[order]

    indptr, adj = csr((src // 64) * 64 + (src + 1) % 64)       # four 64-cycles, one per shard
    slot, counts, block = e.halo_plan(n, world, indptr, adj)
    assert block == 0 and not counts.any() and (slot == -1).all()
    indptr, adj = csr((src + 1) % n)                          # one n-cycle: node 64 q - 1 is read by the next rank
    slot, counts, block = e.halo_plan(n, world, indptr, adj)
    assert block == 1 and list(counts) == [1, 1, 1, 1]
    assert sorted(np.nonzero(slot >= 0)[0]) == [63, 127, 191, 255]
    nodes = np.arange(2 * n, dtype=np.float32).reshape(n, 2)
    h = e.shard_halo(n, 1, world, indptr, adj, nodes, (slot, counts, block))
    assert h['row_begin'] == 64 and h['n_rows'] == 64 and list(h['send_rows']) == [63]
    # row 64 (first owned row of rank 1) reads node 63 = the boundary row of rank 0 -> slot 64 + 0 * 1 + 0; the others are own rows
    assert h['adj_src'][0] == 64 and list(h['adj_src'][1:]) == list(range(0, 63))
    assert h['nodes'].shape == (64 + 4, 2) and np.array_equal(h['nodes'][64], nodes[63]) and np.array_equal(h['nodes'][:64], nodes[64:128])


def test_save_load_round_trip(tmp_path):
    """GNN.save / GNN.load and LGNN.save / LGNN.load (reference GNN.py:93-149, LGNN.py:83-141): architecture, weights, optimizer and
    loss come back; no pickled objects in the files."""
    from GNN import losses, optimizers
    from GNN.GNN import GNNnodeBased, GNNgraphBased
    from GNN.LGNN import LGNN
    from GNN.MLP import MLP, AlphaDropout, Dropout, set_seed
    set_seed(3)
    st = MLP(7, [5, 3], 'selu', 'lecun_normal', 'lecun_normal', dropout_rate=0.2, dropout_pos=[0, 1], alphadropout=True)
    ou = MLP(3, [2], 'softmax', 'glorot_normal', 'zeros', dropout_rate=0.1, dropout_pos=0, batch_normalization=False)
    gnn = GNNnodeBased(net_state=st, net_output=ou, optimizer=optimizers.Adam(0.01, beta_1=0.8), loss_function=losses.categorical_crossentropy,
                       loss_arguments={'from_logits': True}, state_vect_dim=0, max_iteration=7, threshold=0.02, addressed_problem='c',
                       path_writer=str(tmp_path / 'w'))
    gnn.save(str(tmp_path / 'm'))
    assert not any(f.endswith('.pkl') for f in os.listdir(tmp_path / 'm'))
    np.load(str(tmp_path / 'm' / 'net_state.npz'), allow_pickle=False).close()
    back = GNNnodeBased.load(str(tmp_path / 'm'), path_writer=str(tmp_path / 'w2'))
    assert (back.max_iteration, back.state_threshold, back.state_vect_dim, back.addressed_problem) == (7, 0.02, 0, 'c')
    assert back.loss_function is losses.categorical_crossentropy and back.loss_args == {'from_logits': True}
    assert isinstance(back.optimizer, optimizers.Adam) and back.optimizer.get_config() == gnn.optimizer.get_config()
    for a, b in ((gnn.net_state, back.net_state), (gnn.net_output, back.net_output)):
        assert [type(l).__name__ for l in a.layers] == [type(l).__name__ for l in b.layers]
        assert a.activations == b.activations and a.batch_normalization == b.batch_normalization
        assert all(np.array_equal(x, y) for x, y in zip(a.get_weights(), b.get_weights()))
        assert a.dropout_rates() == b.dropout_rates()
    assert isinstance(back.net_state.layers[0], AlphaDropout) and isinstance(back.net_output.layers[0], Dropout)
    mk = lambda: GNNgraphBased(net_state=MLP(7, [3], 'tanh', 'glorot_normal', 'zeros'), net_output=MLP(3, [2], 'softmax', 'glorot_normal', 'zeros'),
                               optimizer=optimizers.SGD(0.1) if hasattr(optimizers, 'SGD') else optimizers.Adam(), loss_function=losses.mse, loss_arguments=None,
                               state_vect_dim=0, max_iteration=3, threshold=0.1, addressed_problem='c', path_writer=str(tmp_path / 'w3'))
    lg = LGNN([mk(), mk()], False, True, optimizers.Adam(), losses.mse, None, 'c', path_writer=str(tmp_path / 'w4'))
    lg.save(str(tmp_path / 'l'))
    lb = LGNN.load(str(tmp_path / 'l'), path_writer=str(tmp_path / 'w5'))
    assert lb.LAYERS == 2 and (lb.get_state, lb.get_output) == (False, True) and isinstance(lb.gnns[0], GNNgraphBased)
    for g0, g1 in zip(lg.gnns, lb.gnns):
        assert all(np.array_equal(x, y) for x, y in zip(g0.net_state.get_weights(), g1.net_state.get_weights()))


def test_optimizer_counts_a_device_step_only_after_it_succeeded_and_restarts_on_a_path_switch():
    from GNN import optimizers
    opt = optimizers.Adam(learning_rate=0.01)
    kind, hyper = opt.device_step_args()
    assert kind == 1 and opt.iterations == 0                    # asking for the arguments does not count the step
    kind2, hyper2 = opt.device_step_args()
    assert hyper2 == hyper                                      # ... so a failed step is retried with the same bias correction
    opt.device_step_done()
    assert opt.iterations == 1 and opt.device_step_args()[1][0] != hyper[0]
    token = opt._slot_token
    # the same optimizer now drives a host-side update: slots and counter restart together (a fresh Keras optimizer)
    p = np.ones(3, np.float32)
    new = opt.apply_gradients([(np.ones(3, np.float32), p)])
    assert opt.iterations == 1 and opt._slot_token is not token and np.allclose(new[0], p - 0.01, atol=1e-6)
    opt.device_step_args()                                      # and back: again from zero
    assert opt.iterations == 0 and opt._m is None


def test_stand_in_transport_exports_what_the_engine_binds():
    """tests/mock_rccl (the stand-in transport of tests/test_gpu_multiprocess.py) builds and exports every entry point rccl_load binds
    (csrc/gnn_engine.hip): a missing symbol would only show on the GPU box otherwise."""
    import ctypes, re, subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.run(['make', '-C', os.path.join(root, 'tests', 'mock_rccl')], check=True, capture_output=True)
    src = open(os.path.join(root, 'gnn_tf_2.x_amd', 'csrc', 'gnn_engine.hip')).read()
    bound = set(re.findall(r'"(nccl[A-Za-z]+)"', src))
    assert {'ncclGetUniqueId', 'ncclCommInitRank', 'ncclAllGather', 'ncclSend', 'ncclRecv', 'ncclGroupStart', 'ncclGroupEnd'} <= bound
    lib = ctypes.CDLL(os.path.join(root, 'tests', 'mock_rccl', 'libmock_rccl.so'))
    for name in bound:
        assert hasattr(lib, name), name


def test_slice_exchange_switch_never_implies_the_pipelined_form(monkeypatch):
    """Loop.set_slice_exchange: any truthy `on` - True, 1, numpy.bool_ - selects the ONE-SHOT form (code 2); the block-wise form
    (code 1, never yet run over RCCL with more than one rank) needs the explicit keyword; falsy switches the exchange off.  No GPU: the C
    entry point is replaced by a recorder."""
    from GNN import _engine as e
    calls = []

    class _Lib:
        def gnn_loop_set_slice_exchange(self, handle, code):
            calls.append(code.value)
            return 0

    monkeypatch.setattr(e, 'lib', lambda: _Lib())
    lp = e.Loop.__new__(e.Loop)
    lp._h = None
    for on in (True, 1, np.bool_(True), 2):
        lp.set_slice_exchange(on)
    lp.set_slice_exchange(True, form='oneshot')
    lp.set_slice_exchange(True, form='pipelined')
    for off in (False, 0, np.bool_(False), None):
        lp.set_slice_exchange(off)
    assert calls == [2, 2, 2, 2, 2, 1, 0, 0, 0, 0]
    with pytest.raises(ValueError):
        lp.set_slice_exchange(True, form='fast')
    lp._h = None          # (nothing to destroy)
