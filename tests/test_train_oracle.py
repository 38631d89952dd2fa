"""The training oracle (one training step: unrolled-loop back-propagation) against central finite differences in float64.
CPU only."""
import numpy as np
import pytest

from oracle import gnn_oracle as orc
from oracle import gnn_train_oracle as tro
from util import make_mlp, random_arcs


def _case(rng, d, graph_based=False, act='tanh'):
    n, nl, al = 40, 3, 2
    arcs = random_arcs(rng, n, 90, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    ng = None
    if graph_based:
        ng = np.zeros((n, 2), np.float32); ng[:25, 0] = 1 / 25; ng[25:, 1] = 1 / 15
    g = orc.make_graph_dict(arcs, nodes, 'average', NodeGraph=ng)
    if not graph_based:
        g['set_mask'] = rng.random(n) < 0.8
    ds, nlc = (d if d else nl), (nl if d else 0)
    st = make_mlp(rng, al + 2 * (ds + nlc), [7, ds], act, gain=0.8, bn_random=True)
    ou = make_mlp(rng, ds + nlc, [5, 2], 'softmax' if False else act, out_activation='softmax', bn_random=True)
    st['dropout'], ou['dropout'] = {0: 0.2}, {0: 0.1, 1: 0.3}
    m = n if graph_based else int(np.sum(g['set_mask'] & g['output_mask']))
    masks_s = [{0: (rng.random((n, st['weights'][0].shape[0])) > 0.2)} for _ in range(6)]
    masks_o = {0: rng.random((m, ds + nlc)) > 0.1, 1: rng.random((m, 5)) > 0.3}
    n_t = 2 if graph_based else m
    targets = np.eye(2)[rng.integers(0, 2, n_t)]
    weights = rng.uniform(0.5, 1.5, n_t)
    s0 = 0.1 * rng.standard_normal((n, ds)) if d else None
    return g, st, ou, s0, masks_s, masks_o, targets, weights


@pytest.mark.parametrize('d,graph_based,act,loss,alpha', [(4, False, 'tanh', 'categorical_crossentropy', False), (0, False, 'selu', 'categorical_crossentropy', False),
                                                            (3, True, 'sigmoid', 'mean_squared_error', False), (0, True, 'relu', 'categorical_crossentropy', False),
                                                            (4, False, 'selu', 'categorical_crossentropy_from_logits', True)])
def test_gradients_match_finite_differences(d, graph_based, act, loss, alpha):
    """alpha: AlphaDropout in place of Dropout (MLP(..., alphadropout=True), reference MLP.py:59-61); *_from_logits: the loss of
    starter.py:83 with from_logits=True."""
    rng = np.random.default_rng(3 + d)
    g, st, ou, s0, ms, mo, targets, weights = _case(rng, d, graph_based, act)
    if alpha: st['alphadropout'] = ou['alphadropout'] = True
    # threshold 0 => exactly max_iteration bodies, so that the iteration count cannot flip under the perturbation
    kw = dict(state_vect_dim=d, max_iteration=4, threshold=0.0, state0=s0, masks_state=ms, masks_output=mo, targets=targets,
              sample_weights=weights, loss=loss, mean=False, graph_based=graph_based)
    res = tro.train_step(g, st, ou, **kw)
    assert res['k'] == 4 and np.isfinite(res['loss'])
    eps = 1e-6
    for net, grads in ((st, res['grads_state']), (ou, res['grads_output'])):
        n_tr = len(grads)
        for wi in range(n_tr):
            w = net['weights'][wi] = np.asarray(net['weights'][wi], np.float64)
            for _ in range(3):
                idx = tuple(rng.integers(0, s) for s in w.shape)
                old = w[idx]
                w[idx] = old + eps; lp = tro.train_step(g, st, ou, **kw)['loss']
                w[idx] = old - eps; lm = tro.train_step(g, st, ou, **kw)['loss']
                w[idx] = old
                fd = (lp - lm) / (2 * eps)
                assert abs(fd - grads[wi][idx]) <= 1e-5 * max(1.0, abs(fd)), (wi, idx, fd, grads[wi][idx])
    # mean=True divides the net_state gradients by the iteration count (GNN_BaseClass.py:241), not the net_output ones
    res_m = tro.train_step(g, st, ou, **dict(kw, mean=True))
    for a, b in zip(res_m['grads_state'], res['grads_state']):
        np.testing.assert_allclose(a, b / 4, rtol=1e-12)
    for a, b in zip(res_m['grads_output'], res['grads_output']):
        np.testing.assert_allclose(a, b, rtol=1e-12)


def test_training_forward_semantics():
    rng = np.random.default_rng(0)
    g, st, ou, s0, ms, mo, targets, weights = _case(rng, 4)
    # without dropout and with BatchNormalization statistics equal to the batch's, training forward == inference forward
    x = rng.standard_normal((50, st['weights'][0].shape[0]))
    net = dict(st, dropout={})
    y, cache = tro.mlp_train_forward(x, net, {}, np.float64)
    inf = dict(net, weights=list(net['weights'][:-2]) + [cache['batch_mean'], cache['batch_var']])
    np.testing.assert_allclose(y, orc.mlp_forward(x, inf['weights'], inf['activations'], True, np.float64), atol=1e-12)
    # dropout scales kept units by 1 / (1 - rate)
    y2, c2 = tro.mlp_train_forward(x, dict(st, dropout={0: 0.5}, batch_normalization=False, weights=st['weights'][:-4]), {0: np.ones_like(x)}, np.float64)
    y3 = orc.mlp_forward(2 * x, st['weights'][:-4], st['activations'], False, np.float64)
    np.testing.assert_allclose(y2, y3, atol=1e-12)
    # moving statistics move once per executed body
    res = tro.train_step(g, st, ou, 4, 3, 0.0, s0, ms, mo, targets, weights)
    assert res['k'] == 3
    mv = np.asarray(st['weights'][-2], np.float64)
    assert not np.allclose(res['moving_state'][0], mv)
    # Adam: first step moves every parameter by about lr against the gradient sign
    p, gr = [np.ones(3)], [np.array([0.5, -2.0, 1e-3])]
    new = tro.adam_update(p, gr, [np.zeros(3)], [np.zeros(3)], 1)
    np.testing.assert_allclose(new[0], 1 - 0.001 * np.sign(gr[0]), atol=5e-6)   # epsilon 1e-7 is not bias-corrected in Keras


def _lgnn_case(rng, d, graph_based, get_state, get_output, n_layers=3, loss='mean_squared_error'):
    n, nl, al, t = 30, 3, 2, 2
    arcs = random_arcs(rng, n, 70, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    ng = None
    if graph_based:
        ng = np.zeros((n, 2), np.float32); ng[:18, 0] = 1 / 18; ng[18:, 1] = 1 / 12
    g = orc.make_graph_dict(arcs, nodes, 'average', NodeGraph=ng)
    if not graph_based:
        g['set_mask'] = rng.random(n) < 0.8
    m = int(np.sum(g['set_mask'] & g['output_mask']))
    layers, s0, ms, mo = [], [], [], []
    nl_i = nl
    for i in range(n_layers):
        ds, nlc = (d if d else nl_i), (nl_i if d else 0)
        st = make_mlp(rng, al + 2 * (ds + nlc), [6, ds], 'tanh', gain=0.8, bn_random=True)
        # categorical_crossentropy on a BatchNormalization output is clipped almost everywhere (zero gradient): no BN there
        if loss == 'categorical_crossentropy':
            ou = make_mlp(rng, ds + nlc, [t], 'tanh', out_activation='softmax')
            ou.update(batch_normalization=False, weights=ou['weights'][:2])
        else:
            ou = make_mlp(rng, ds + nlc, [t], 'tanh', out_activation='softmax', bn_random=True)
        st['dropout'], ou['dropout'] = {0: 0.2}, {0: 0.1}
        layers.append(dict(net_state=st, net_output=ou, state_vect_dim=d, max_iteration=3, threshold=0.0))
        s0.append(0.1 * rng.standard_normal((n, ds)) if d else None)
        ms.append([{0: rng.random((n, st['weights'][0].shape[0])) > 0.2} for _ in range(3)])
        mo.append({0: rng.random((m, ds + nlc)) > 0.1})
        nl_i = nl + get_state * ds + get_output * t
    n_t = 2 if graph_based else m
    targets = np.eye(t)[rng.integers(0, t, n_t)]
    weights = rng.uniform(0.5, 1.5, n_t)
    return g, layers, s0, ms, mo, targets, weights


@pytest.mark.parametrize('d,graph_based,get_state,get_output,mode,loss', [
    (3, False, True, True, 'parallel', 'mean_squared_error'), (0, False, True, False, 'residual', 'categorical_crossentropy'),
    (2, True, False, True, 'parallel', 'categorical_crossentropy'), (0, True, True, True, 'residual', 'mean_squared_error')])
def test_lgnn_joint_gradients_match_finite_differences(d, graph_based, get_state, get_output, mode, loss):
    """'parallel' / 'residual' training: the gradient of layer i includes the path through layer i + 1's labels."""
    rng = np.random.default_rng(11 + d)
    g, layers, s0, ms, mo, targets, weights = _lgnn_case(rng, d, graph_based, get_state, get_output, loss=loss)
    kw = dict(loss=loss, get_state=get_state, get_output=get_output, training_mode=mode, state0=s0, masks_state=ms, masks_output=mo,
              targets=targets, sample_weights=weights, mean=False, graph_based=graph_based)
    res = tro.lgnn_train_step(g, layers, **kw)
    assert res['k'] == [3.0] * 3 and np.isfinite(res['loss'])
    eps = 1e-6
    for li, ly in enumerate(layers):
        for net, grads in ((ly['net_state'], res['grads_state'][li]), (ly['net_output'], res['grads_output'][li])):
            for wi in range(len(grads)):
                w = net['weights'][wi] = np.asarray(net['weights'][wi], np.float64)
                for _ in range(2):
                    idx = tuple(rng.integers(0, s) for s in w.shape)
                    old = w[idx]
                    w[idx] = old + eps; lp = tro.lgnn_train_step(g, layers, **kw)['loss']
                    w[idx] = old - eps; lm = tro.lgnn_train_step(g, layers, **kw)['loss']
                    w[idx] = old
                    fd = (lp - lm) / (2 * eps)
                    assert abs(fd - grads[wi][idx]) <= 2e-5 * max(1.0, abs(fd)), (li, wi, idx, fd, grads[wi][idx])
    # the first layer's gradient is NOT what it would get from its own loss alone (cross-layer path exists)
    if mode == 'parallel':
        own = tro.lgnn_train_step(g, layers[:1], **dict(kw, state0=s0[:1], masks_state=ms[:1], masks_output=mo[:1]))
        assert not np.allclose(own['grads_state'][0][0] / 3, res['grads_state'][0][0], rtol=1e-3)


@pytest.mark.parametrize('d', [3, 0])
def test_edge_based_gradients_match_finite_differences(d):
    """GNNedgeBased: net_output runs on [F[dst] | F[src] | arc label] of the masked arcs (GNN.py:289-302); both endpoints of
    every masked arc receive gradient."""
    rng = np.random.default_rng(21 + d)
    n, nl, al, e = 30, 3, 2, 80
    arcs = random_arcs(rng, n, e, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    g = orc.make_graph_dict(arcs, nodes, 'average')
    e = len(g['arcs'])
    g['set_mask'] = rng.random(e) < 0.7
    g['output_mask'] = np.ones(e, bool)
    ds, nlc = (d if d else nl), (nl if d else 0)
    st = make_mlp(rng, al + 2 * (ds + nlc), [6, ds], 'tanh', gain=0.8, bn_random=True)
    ou = make_mlp(rng, 2 * (ds + nlc) + al, [5, 2], 'tanh', out_activation='softmax')
    ou.update(batch_normalization=False, weights=ou['weights'][:4])
    st['dropout'], ou['dropout'] = {0: 0.2}, {1: 0.3}
    m = int(g['set_mask'].sum())
    ms = [{0: rng.random((n, st['weights'][0].shape[0])) > 0.2} for _ in range(3)]
    mo = {1: rng.random((m, 5)) > 0.3}
    targets = np.eye(2)[rng.integers(0, 2, m)]
    weights = rng.uniform(0.5, 1.5, m)
    s0 = 0.1 * rng.standard_normal((n, ds)) if d else None
    kw = dict(state_vect_dim=d, max_iteration=3, threshold=0.0, state0=s0, masks_state=ms, masks_output=mo, targets=targets,
              sample_weights=weights, mean=False, edge_based=True)
    res = tro.train_step(g, st, ou, **kw)
    assert res['k'] == 3 and res['out'].shape == (m, 2)
    eps = 1e-6
    for net, grads in ((st, res['grads_state']), (ou, res['grads_output'])):
        for wi in range(len(grads)):
            w = net['weights'][wi] = np.asarray(net['weights'][wi], np.float64)
            for _ in range(3):
                idx = tuple(rng.integers(0, s_) for s_ in w.shape)
                old = w[idx]
                w[idx] = old + eps; lp = tro.train_step(g, st, ou, **kw)['loss']
                w[idx] = old - eps; lm = tro.train_step(g, st, ou, **kw)['loss']
                w[idx] = old
                fd = (lp - lm) / (2 * eps)
                assert abs(fd - grads[wi][idx]) <= 1e-5 * max(1.0, abs(fd)), (wi, idx, fd, grads[wi][idx])


@pytest.mark.parametrize('d,get_state,get_output,mode', [(3, True, True, 'parallel'), (0, False, True, 'residual'), (2, True, False, 'parallel')])
def test_edge_lgnn_joint_gradients_match_finite_differences(d, get_state, get_output, mode):
    """Edge-based LGNN, joint training: layer i also receives gradient through the ARC labels of layer i + 1 (LGNN.py:253-254)."""
    rng = np.random.default_rng(31 + d)
    n, nl, al, t, L = 24, 3, 2, 2, 3
    arcs = random_arcs(rng, n, 60, al)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    g = orc.make_graph_dict(arcs, nodes, 'average')
    e = len(g['arcs'])
    g['set_mask'], g['output_mask'] = rng.random(e) < 0.75, np.ones(e, bool)
    m = int(g['set_mask'].sum())
    layers, s0, ms, mo = [], [], [], []
    for i in range(L):
        ins, ls = orc.get_inout_dims('state', nl, al, t, 'a', d, [6], layer=i, get_state=get_state, get_output=get_output)
        ino, lo = orc.get_inout_dims('output', nl, al, t, 'a', d, None, layer=i, get_state=get_state, get_output=get_output)
        st = make_mlp(rng, ins, ls, 'tanh', gain=0.8, bn_random=True)
        ou = make_mlp(rng, ino, lo, 'tanh', out_activation='softmax', bn_random=True)
        st['dropout'], ou['dropout'] = {0: 0.2}, {0: 0.1}
        layers.append(dict(net_state=st, net_output=ou, state_vect_dim=d, max_iteration=3, threshold=0.0))
        s0.append(0.1 * rng.standard_normal((n, d)) if d else None)
        ms.append([{0: rng.random((n, ins)) > 0.2} for _ in range(3)])
        mo.append({0: rng.random((m, ino)) > 0.1})
    targets = rng.random((m, t))
    weights = rng.uniform(0.5, 1.5, m)
    kw = dict(get_state=get_state, get_output=get_output, training_mode=mode, state0=s0, masks_state=ms, masks_output=mo, targets=targets,
              sample_weights=weights, loss='mean_squared_error', mean=False, edge_based=True)
    res = tro.lgnn_train_step(g, layers, **kw)
    assert res['k'] == [3.0] * L and np.isfinite(res['loss'])
    eps = 1e-6
    for li, ly in enumerate(layers):
        for net, grads in ((ly['net_state'], res['grads_state'][li]), (ly['net_output'], res['grads_output'][li])):
            for wi in range(len(grads)):
                w = net['weights'][wi] = np.asarray(net['weights'][wi], np.float64)
                for _ in range(2):
                    idx = tuple(rng.integers(0, s_) for s_ in w.shape)
                    old = w[idx]
                    w[idx] = old + eps; lp = tro.lgnn_train_step(g, layers, **kw)['loss']
                    w[idx] = old - eps; lm = tro.lgnn_train_step(g, layers, **kw)['loss']
                    w[idx] = old
                    fd = (lp - lm) / (2 * eps)
                    assert abs(fd - grads[wi][idx]) <= 2e-5 * max(1.0, abs(fd)), (li, wi, idx, fd, grads[wi][idx])
