"""The engine's SHARDED path on one GPU: a loopback communicator group (gnn_comm_create_loopback) runs `world` ranks of the
node-range sharded loop in one process, with the real HIP kernels (row_begin > 0, padded replicas, a short / empty last
shard, per-rank flag slots, the exchange call sites of the RCCL path) and device-to-device copies in place of RCCL.

Bar: impl 0 / 1 bit-identical to the unsharded run AND to the C oracle (k, owned state rows, outputs); impl 2 with the same k
and within the fp32-noise tolerance of the unsharded run.  The reference has no counterpart (single device): what is sharded
is the row-wise work of convergence() (GNN/GNN.py:223-242) and the global reduce_any of condition() (GNN.py:218)."""
import numpy as np
import pytest

from oracle import c_oracle as corc
from oracle import gnn_oracle as orc
from util import make_mlp, random_arcs

pytestmark = pytest.mark.gpu


def _engine():
    from GNN import _engine
    return _engine


def _graph(rng, n, nl, al, deg=4, mode='average'):
    arcs = random_arcs(rng, n, deg * n, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    return orc.make_graph_dict(arcs, nodes, mode)


def _csr_parts(g):
    arc_labels = np.asarray(g['arcs'], np.float32)[:, 2:]
    return g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], arc_labels[g['arcT'][1]]


def _sharded_loops(e, g, st, ou, d, max_it, thr, s0, world, impl, halo=False, strict=True):
    """One loop per rank of a loopback group.  Returns (comms, graphs, loops, ranges)."""
    n = g['nodes'].shape[0]
    indptr, adj_src, adj_w, arc_w, arc_lab = _csr_parts(g)
    mask = np.logical_and(g['set_mask'], g['output_mask'])
    comms = e.Comm.loopback(world)
    mst = e.Mlp(st['weights'], st['activations'], st['batch_normalization'])
    mou = e.Mlp(ou['weights'], ou['activations'], ou['batch_normalization'])
    plan = e.halo_plan(n, world, indptr, adj_src) if halo else None
    graphs, loops, ranges = [], [], []
    loops_fell_back = False
    for r in range(world):
        rb, nr, ip, src, w, aw, al_ = e.shard_csr(n, r, world, indptr, adj_src, adj_w, arc_w, arc_lab)
        if halo:
            h = e.shard_halo(n, r, world, indptr, adj_src, g['nodes'], plan)
            gr = e.Graph.halo(n, r, world, h['block'], h['send_rows'], ip, h['adj_src'], w, aw, al_, h['nodes'], mask[rb:rb + nr])
        else:
            gr = e.Graph(n, ip, src, w, aw, al_, g['nodes'], mask[rb:rb + nr], row_begin=rb)
        lp = e.Loop(gr, mst, mou, d, max_it, thr, comms[r])
        used = lp.set_impl(impl)
        assert used == impl or nr == 0 or not strict          # a rank without rows has nothing to fuse
        if used != impl and nr: loops_fell_back = True
        if d:
            lp.set_state0(s0[rb:rb + nr])
        graphs.append(gr); loops.append(lp); ranges.append((rb, nr))
    if not strict and loops_fell_back:          # the shape is outside the fused kernel on some rank: every rank on the per-op path (a group runs one path)
        for lp in loops: lp.set_impl(0)
    return comms, graphs, loops, ranges


def _collect(loops, ranges, mask):
    state = np.concatenate([lp.state() for lp in loops])
    out = np.concatenate([lp.output() for lp in loops])
    return state, out


def _unsharded(e, g, st, ou, d, max_it, thr, s0, impl):
    indptr, adj_src, adj_w, arc_w, arc_lab = _csr_parts(g)
    mask = np.logical_and(g['set_mask'], g['output_mask'])
    gr = e.Graph(g['nodes'].shape[0], indptr, adj_src, adj_w, arc_w, arc_lab, g['nodes'], mask)
    lp = e.Loop(gr, e.Mlp(st['weights'], st['activations'], st['batch_normalization']),
                e.Mlp(ou['weights'], ou['activations'], ou['batch_normalization']), d, max_it, thr)
    lp.set_impl(impl)
    if d:
        lp.set_state0(s0)
    k = lp.run()
    return k, lp.state(), lp.output()


def _case(seed, n, d, nl=3, al=1, hidden=(16,), act='selu', gain=0.6):
    rng = np.random.default_rng(seed)
    g = _graph(rng, n, nl, al)
    ds, nlc = (d if d else nl), (nl if d else 0)
    st = make_mlp(rng, al + 2 * (ds + nlc), list(hidden) + [ds], act, gain=gain, bn_random=True)
    ou = make_mlp(rng, ds + nlc, [2], 'softmax', bn_random=True)
    s0 = (0.1 * rng.standard_normal((n, ds))).astype(np.float32) if d else None
    g['set_mask'] = rng.random(n) < 0.8
    g['output_mask'] = rng.random(n) < 0.7
    return g, st, ou, s0


@pytest.mark.parametrize('halo', [False, True])
@pytest.mark.parametrize('world', [2, 3, 8])
@pytest.mark.parametrize('n,d,hidden', [(1000, 8, (16,)), (4099, 64, (128, 128)), (333, 0, (7,))])
def test_sharded_loop_bit_exact(n, d, hidden, world, halo):
    """n = 4099 with the BASELINE net (135 -> 128 -> 128 -> 64): short last shard and padding rows; n = 333 on 8 ranks: the
    last ranks own nothing at all."""
    e = _engine()
    g, st, ou, s0 = _case(100 + n + world, n, d, hidden=hidden)
    kc, sc, oc = corc.loop_node(g, st, ou, d, 30, 0.01, s0)
    for impl in (1, 0):
        comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, d, 30, 0.01, s0, world, impl, halo)
        k = e.Loop.run_group(loops)
        state, out = _collect(loops, ranges, None)
        assert k == kc, (impl, k, kc)
        assert np.array_equal(state, sc), impl
        assert out.shape == oc.shape and np.array_equal(out, oc), impl
        k2 = e.Loop.run_group(loops)                               # second run on the same handles (state restored, flags reset)
        assert k2 == kc and np.array_equal(_collect(loops, ranges, None)[0], sc)
        for lp in loops: lp.close()
        for c in comms: c.close()
    # default arithmetic (impl 2): same k, fp32 rounding noise away from the unsharded run of the same arithmetic
    comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, d, 30, 0.01, s0, world, 2, halo)
    k = e.Loop.run_group(loops)
    state, out = _collect(loops, ranges, None)
    ku, su, ou_ = _unsharded(e, g, st, ou, d, 30, 0.01, s0, 2)
    assert k == ku == kc
    # (small unsharded graphs run the persistent loop, which uses the exact fp32 chain in both fused modes: compare by value)
    tol = 2e-6 * max(1.0, float(np.max(np.abs(sc))))
    assert np.max(np.abs(state - su)) < tol and np.max(np.abs(out - ou_)) < 2e-6
    assert np.max(np.abs(state - sc)) < tol


def test_sharded_large_graph_matches_unsharded():
    """100k nodes / ~1M arcs, the BASELINE net, 8 ranks: owned rows bit-equal to the unsharded run (impl 1) and to the C oracle."""
    e = _engine()
    from GNN import GNN_utils as utils
    n, d = 100_000, 64
    s = utils.syntheticGraph(n, 10.0, 3, 1, 2, seed=11)
    rng = np.random.default_rng(12)
    st = make_mlp(rng, 1 + 2 * (3 + d), [128, 128, d], 'selu', gain=0.6)
    ou = make_mlp(rng, 3 + d, [2], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    arcs = np.concatenate([np.stack([s['src'], s['dst']], 1).astype(np.float32), s['arc_labels']], axis=1)
    g = dict(nodes=s['nodes'], arcs=arcs, set_mask=np.ones(n, bool), output_mask=np.ones(n, bool),
             adjT=(s['indptr'], s['adj_src'], s['adj_w']), arcT=(s['indptr'], s['arc_perm'], s['arc_w']))
    ku, su, ou_ = _unsharded(e, g, st, ou, d, 12, 0.0, s0, 1)
    for world, halo in ((8, False), (3, True)):
        comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, d, 12, 0.0, s0, world, 1, halo)
        k = e.Loop.run_group(loops)
        state, out = _collect(loops, ranges, None)
        assert k == ku == 12 and np.array_equal(state, su) and np.array_equal(out, ou_)
        for lp in loops: lp.close()
    kc, sc, oc = corc.loop_node(g, st, ou, d, 12, 0.0, s0)
    assert np.array_equal(su, sc) and np.array_equal(ou_, oc)


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_graph_readout(world):
    """GNNgraphBased readout (GNN.py:331-332) on shards: per-rank partial sums added in rank order.  Graphs that straddle a
    shard boundary differ from the unsharded fmaf chain by rounding only; graphs inside one shard are bit-identical."""
    e = _engine()
    rng = np.random.default_rng(77 + world)
    n, d, n_graphs = 960, 8, 12
    g, st, ou, s0 = _case(5 + world, n, d)
    g['set_mask'] = np.ones(n, bool); g['output_mask'] = np.ones(n, bool)
    bounds = np.sort(rng.choice(np.arange(1, n), n_graphs - 1, replace=False))
    sizes = np.diff(np.concatenate([[0], bounds, [n]]))
    ng_indptr = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int32)
    ng_node = np.arange(n, dtype=np.int32)
    ng_w = np.repeat(1.0 / sizes, sizes).astype(np.float32)
    comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, d, 20, 0.01, s0, world, 1)
    k = e.Loop.run_group(loops)
    got = e.Loop.readout_group(loops, ng_indptr, ng_node, ng_w)
    kc, sc, on = corc.loop_node(g, st, ou, d, 20, 0.01, s0)
    node_graph = np.zeros((n, n_graphs), np.float32)
    for gi in range(n_graphs):
        node_graph[ng_indptr[gi]:ng_indptr[gi + 1], gi] = ng_w[ng_indptr[gi]]
    want = corc.readout(node_graph, on)
    assert k == kc and got.shape == want.shape
    shard = ((n + world - 1) // world + 31) // 32 * 32
    inside = np.array([ng_indptr[gi] // shard == (ng_indptr[gi + 1] - 1) // shard for gi in range(n_graphs)])
    assert inside.any() and not inside.all()
    assert np.array_equal(got[inside], want[inside])
    assert np.max(np.abs(got - want)) < 1e-6


@pytest.mark.parametrize('sliced', [False, True, 'halo'])
@pytest.mark.parametrize('world', [2, 3])
def test_sharded_lgnn_relabelling(world, sliced):
    """LGNN.update_graph (LGNN.py:227-260) on shards: every rank relabels its own rows, the new label rows are exchanged, the
    next layer runs on them.  Two layers, get_state and get_output both on; bit-equal to the single-GPU stack.  sliced: both layers
    use the feature-sliced exchange; the derived graphs of layer 1 share the whole-graph adjacency of their base (no second upload).
    'halo': boundary-exchange shards - the relabelled BOUNDARY rows are exchanged block-wise, like the state rows of an iteration."""
    e = _engine()
    rng = np.random.default_rng(9 + world)
    n, d, nl, al = 700, 6, 3, 1
    g, st0, ou0, s0 = _case(21 + world, n, d)
    nl1 = nl + d + 2
    st1 = make_mlp(rng, al + 2 * (d + nl1), [16, d], 'selu', gain=0.6, bn_random=True)
    ou1 = make_mlp(rng, d + nl1, [2], 'softmax', bn_random=True)
    s1 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    mask = np.logical_and(g['set_mask'], g['output_mask'])
    indptr, adj_src, adj_w, arc_w, arc_lab = _csr_parts(g)

    halo = sliced == 'halo'
    sliced = sliced is True

    def stack(world_):
        comms = e.Comm.loopback(world_)
        m = [e.Mlp(x['weights'], x['activations'], True) for x in (st0, ou0, st1, ou1)]
        bases, derived, l0, l1 = [], [], [], []
        plan = e.halo_plan(n, world_, indptr, adj_src) if halo else None
        for r in range(world_):
            rb, nr, ip, src, w, aw, al_ = e.shard_csr(n, r, world_, indptr, adj_src, adj_w, arc_w, arc_lab)
            if halo:
                h = e.shard_halo(n, r, world_, indptr, adj_src, g['nodes'], plan)
                bases.append(e.Graph.halo(n, r, world_, h['block'], h['send_rows'], ip, h['adj_src'], w, aw, al_, h['nodes'], mask[rb:rb + nr]))
            else:
                bases.append(e.Graph(n, ip, src, w, aw, al_, g['nodes'], mask[rb:rb + nr], row_begin=rb))
            l0.append(e.Loop(bases[-1], m[0], m[1], d, 20, 0.01, comms[r]))
            l0[-1].set_impl(1); l0[-1].set_state0(s0[rb:rb + nr])
            if sliced and world_ > 1:
                bases[-1].set_full_adjacency(n, indptr, adj_src, adj_w)
                l0[-1].set_slice_exchange(True)
        k0 = e.Loop.run_group(l0)
        for r in range(world_):
            derived.append(bases[r].derive(d + 2))
        e.Graph.update_labels_group(derived, bases, l0, True, True)
        if halo:                                                       # index space [own rows | boundary blocks]: the own rows, in rank order
            labels = np.concatenate([derived[r].nodes()[:e.shard_range(n, r, world_)[1]] for r in range(world_)])
        else:
            labels = derived[0].nodes()
        for r in range(world_):
            if not halo: assert np.array_equal(derived[r].nodes(), labels)          # every rank holds the same relabelled graph
            l0[r].close()                                              # frees the rank's slot in the group
        for r in range(world_):
            rb, nr = e.shard_range(n, r, world_)
            l1.append(e.Loop(derived[r], m[2], m[3], d, 20, 0.01, comms[r]))
            l1[-1].set_impl(1); l1[-1].set_state0(s1[rb:rb + nr])
            if sliced and world_ > 1: l1[-1].set_slice_exchange(True)
        k1 = e.Loop.run_group(l1)
        return k0, k1, labels, np.concatenate([lp.state() for lp in l1]), np.concatenate([lp.output() for lp in l1])

    k0, k1, labels, state, out = stack(world)
    # single-GPU stack through the same C ABI (world 1 loopback group degenerates to plain copies of nothing)
    gr = e.Graph(n, indptr, adj_src, adj_w, arc_w, arc_lab, g['nodes'], mask)
    a = e.Loop(gr, e.Mlp(st0['weights'], st0['activations'], True), e.Mlp(ou0['weights'], ou0['activations'], True), d, 20, 0.01)
    a.set_impl(1); a.set_state0(s0)
    ka = a.run()
    dg = gr.derive(d + 2)
    dg.update_labels(gr, a, True, True)
    b = e.Loop(dg, e.Mlp(st1['weights'], st1['activations'], True), e.Mlp(ou1['weights'], ou1['activations'], True), d, 20, 0.01)
    b.set_impl(1); b.set_state0(s1)
    kb = b.run()
    assert (k0, k1) == (ka, kb)
    assert np.array_equal(labels, dg.nodes())
    assert np.array_equal(state, b.state()) and np.array_equal(out, b.output())


@pytest.mark.parametrize('world', [2, 3])
def test_sharded_edge_based_readout(world):
    """GNNedgeBased.apply_filters + net_output (reference GNN.py:289-302, :279) on node-range shards: a rank reads out the masked arcs of
    its own CSR rows (destination owned, source anywhere in its replica); the per-rank outputs in rank order are the unsharded output."""
    e = _engine()
    rng = np.random.default_rng(70 + world)
    n, d, nl, al = 500, 6, 3, 1
    g, st, _, s0 = _case(33 + world, n, d)
    ou = make_mlp(rng, 2 * (d + nl) + al, [5, 2], 'tanh', out_activation='softmax', bn_random=True)      # per-arc readout: [F(dst) | F(src) | arc label]
    indptr, adj_src, adj_w, arc_w, arc_lab = _csr_parts(g)
    E = len(adj_src)
    entry_dst = np.repeat(np.arange(n, dtype=np.int32), np.diff(indptr))
    arc_labels = (2 * rng.random((E, al)) - 1).astype(np.float32)        # labels paired with the entries by position (GNN.py:294-299)
    arc_mask = rng.random(E) < 0.6
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)

    def run(world_):
        comms = e.Comm.loopback(world_)
        loops, outs = [], []
        for r in range(world_):
            rb, nr, ip, src, w, aw, al_ = e.shard_csr(n, r, world_, indptr, adj_src, adj_w, arc_w, arc_lab)
            e0, e1 = int(indptr[rb]), int(indptr[rb + nr])
            gr = e.Graph(n, ip, src, w, aw, al_, g['nodes'], np.ones(nr, np.uint8), row_begin=rb)
            lp = e.Loop(gr, mst, mou, d, 15, 0.01, comms[r])
            lp.set_impl(1); lp.set_state0(s0[rb:rb + nr])
            lp.set_edge_readout(entry_dst[e0:e1] - rb, arc_labels[e0:e1], arc_mask[e0:e1])
            loops.append(lp)
        k = e.Loop.run_group(loops)
        return k, np.concatenate([lp.state() for lp in loops]), np.concatenate([lp.output() for lp in loops])

    k1, s1, o1 = run(1)
    kw, sw, ow = run(world)
    assert o1.shape == (int(arc_mask.sum()), 2)
    assert kw == k1 and np.array_equal(sw, s1) and np.array_equal(ow, o1)


@pytest.mark.parametrize('world,halo', [(2, False), (3, True)])
def test_sharded_default_path_certified_gate(world, halo):
    """The exact-k contract of the default arithmetic on shards (reference GNN/GNN.py:202-220: the reduce_any spans all nodes): a threshold
    placed ON the largest distance / norm ratio of a body leaves that body's gate without a robust mover and with a borderline node on
    SOME rank; every rank reads the same exchanged flag words (moved / robust / borderline), so all ranks repeat the Loop on the bit-exact
    path together and return the oracle's k, states and outputs bit for bit."""
    e = _engine()
    n, d = 1500, 64
    g, st, ou, s0 = _case(77, n, d, hidden=(128, 128), act='tanh', gain=0.5)
    _, s3, _ = corc.loop_node(g, st, ou, d, 3, 0.0, s0)
    _, s4, _ = corc.loop_node(g, st, ou, d, 4, 0.0, s0)
    dist = np.zeros(n, np.float32); nrm = np.zeros(n, np.float32)
    for c in range(d):
        df = s4[:, c] - s3[:, c]
        dist = dist + df * df
        nrm = nrm + s3[:, c] * s3[:, c]
    thr = float(np.max(np.sqrt(dist) / np.sqrt(nrm)))
    kc, sc, oc = corc.loop_node(g, st, ou, d, 12, thr, s0)
    comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, d, 12, thr, s0, world, 2, halo=halo)
    k = e.Loop.run_group(loops)
    assert k == kc
    assert all(lp.gate_info()[0] for lp in loops)                   # every rank's run was the repeat on impl 1
    state, out = _collect(loops, ranges, None)
    assert np.array_equal(state, sc) and np.array_equal(out, oc)
    k2 = e.Loop.run_group(loops)                                     # and the group keeps working on the default path afterwards
    assert k2 == kc
    for lp in loops: lp.close()
    for gr in graphs: gr.close()
    for c_ in comms: c_.close()


def test_loopback_group_errors():
    e = _engine()
    g, st, ou, s0 = _case(3, 200, 4)
    comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, 4, 5, 0.01, s0, 2, 1)
    with pytest.raises(e.EngineError):
        loops[0].run()                                   # a member of a group cannot run alone
    with pytest.raises(ValueError):
        e.Loop.run_group(loops[::-1])                    # rank order
    with pytest.raises(ValueError):                      # one loop per rank and group
        e.Loop(graphs[0], e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), 4, 5, 0.01, comms[0])
    assert e.Loop.run_group(loops) == corc.loop_node(g, st, ou, 4, 5, 0.01, s0)[0]


@pytest.mark.parametrize('form', ['pipelined', 'oneshot'])
@pytest.mark.parametrize('world', [2, 4, 8])
@pytest.mark.parametrize('n,d,hidden', [(1000, 8, (16,)), (4099, 64, (128, 128)), (333, 16, (7,))])
def test_feature_sliced_exchange_bit_exact(n, d, hidden, world, form):
    """gnn_loop_set_slice_exchange: every rank aggregates ITS columns of the state for all nodes over the whole graph's adjacency
    (two all-to-all steps per iteration instead of the all-gather of rows).  The fmaf chain of an aggregated element is the
    same CSR-ordered chain as in the replicated layouts: impl 0 / 1 bit-identical to the C oracle, impl 2 within the tolerance
    of the unsharded run; n = 333 on 8 ranks: the last ranks own nothing, n = 4099: short last shard, partial tiles.
    'pipelined': the slice is aggregated in one row block per destination rank and each block is sent on a second stream while
    the next one is aggregated; 'oneshot' (the default of set_slice_exchange(True)): whole slice, then one grouped all-to-all.  Same bits."""
    e = _engine()
    g, st, ou, s0 = _case(300 + n + world, n, d, hidden=hidden)
    indptr, adj_src, adj_w, _, _ = _csr_parts(g)
    kc, sc, oc = corc.loop_node(g, st, ou, d, 30, 0.01, s0)

    def sliced(impl):
        comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, d, 30, 0.01, s0, world, impl)
        for gr, lp in zip(graphs, loops):
            gr.set_full_adjacency(n, indptr, adj_src, adj_w)
            lp.set_slice_exchange(True, form=form)
        return comms, graphs, loops, ranges

    for impl in (1, 0):
        comms, graphs, loops, ranges = sliced(impl)
        k = e.Loop.run_group(loops)
        state, out = _collect(loops, ranges, None)
        assert k == kc, (impl, k, kc)
        assert np.array_equal(state, sc), impl
        assert out.shape == oc.shape and np.array_equal(out, oc), impl
        k2 = e.Loop.run_group(loops)
        assert k2 == kc and np.array_equal(_collect(loops, ranges, None)[0], sc)
        for lp in loops: lp.close()
        for c in comms: c.close()
    comms, graphs, loops, ranges = sliced(2)
    k = e.Loop.run_group(loops)
    state, out = _collect(loops, ranges, None)
    ku, su, ou_ = _unsharded(e, g, st, ou, d, 30, 0.01, s0, 2)
    assert k == ku == kc
    tol = 2e-6 * max(1.0, float(np.max(np.abs(sc))))
    assert np.max(np.abs(state - su)) < tol and np.max(np.abs(out - ou_)) < 2e-6


def test_feature_sliced_exchange_argument_errors():
    e = _engine()
    g, st, ou, s0 = _case(7, 200, 6)
    comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, 6, 5, 0.01, s0, 4, 1)
    with pytest.raises((e.EngineError, ValueError)):
        loops[0].set_slice_exchange(True)                    # no full adjacency yet
    indptr, adj_src, adj_w, _, _ = _csr_parts(g)
    graphs[0].set_full_adjacency(200, indptr, adj_src, adj_w)
    with pytest.raises((e.EngineError, ValueError)):
        loops[0].set_slice_exchange(True)                    # 6 columns over 4 ranks
    with pytest.raises((e.EngineError, ValueError)):
        graphs[1].set_full_adjacency(199, indptr[:200], adj_src, adj_w)      # another graph's size


def test_sharded_loop_random_shapes():
    """Loopback groups on 16 seeded random combinations of world size (2 .. 8), exchange form (whole shards, boundary blocks, sliced with the
    pipelined and the one-shot return), state width, net widths, activation and node count (down to ranks without rows): k, states and
    outputs of the exact path bit-identical to the C oracle, the default path within tolerance."""
    e = _engine()
    rng = np.random.default_rng(20261007)
    acts = ['selu', 'tanh', 'relu', 'sigmoid', 'elu', 'linear']
    for case in range(16):
        world = int(rng.choice([2, 3, 4, 5, 8]))
        d = int(rng.choice([0, 4, 8, 16, 24, 40, 64]))
        nl = int(rng.integers(1, 7)) if d else int(rng.choice([4, 8, 16]))
        hidden = tuple(int(x) for x in rng.choice([7, 16, 32, 64, 128], size=int(rng.integers(0, 3))))
        n = int(rng.choice([50, 333, 1000, 4099]))
        ds = d if d else nl
        forms = ['full', 'halo'] + (['slice1', 'slice2'] if ds % world == 0 else [])
        form = str(rng.choice(forms))
        g, st, ou, s0 = _case(9000 + case, n, d, nl=nl, al=int(rng.integers(1, 4)), hidden=hidden, act=acts[case % len(acts)], gain=0.5)
        indptr, adj_src, adj_w, _, _ = _csr_parts(g)
        max_it, thr = int(rng.integers(1, 12)), float(rng.choice([0.0, 0.01]))
        kc, sc, oc = corc.loop_node(g, st, ou, d, max_it, thr, s0)
        for impl in (1, 2):
            comms, graphs, loops, ranges = _sharded_loops(e, g, st, ou, d, max_it, thr, s0, world, impl, halo=(form == 'halo'), strict=False)
            if form.startswith('slice'):
                for gr, lp in zip(graphs, loops):
                    gr.set_full_adjacency(n, indptr, adj_src, adj_w)
                    lp.set_slice_exchange(True, form='pipelined' if form[-1] == '1' else 'oneshot')
            k = e.Loop.run_group(loops)
            state, out = _collect(loops, ranges, None)
            tag = (case, world, form, d, nl, hidden, n, impl, max_it, thr)
            if impl == 1:          # (a shape outside the fused kernel runs the per-op path on every rank: exact as well)
                assert k == kc and np.array_equal(state, sc) and np.array_equal(out, oc), tag
            else:
                assert abs(k - kc) <= 1, tag
                if k == kc:
                    assert np.max(np.abs(state - sc)) < 1e-5 * max(1.0, float(np.max(np.abs(sc)))), tag
            for lp in loops: lp.close()
            for c in comms: c.close()
