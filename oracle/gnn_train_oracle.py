"""CPU oracle (NumPy) for ONE training step of GNNnodeBased / GNNgraphBased: training-mode forward through the unrolled
state loop, loss, and back-propagation through all iterations.  TEST INFRASTRUCTURE ONLY (see gnn_oracle.py).

Follows (paths relative to the reference root):
  GNN/GNN_BaseClass.py:231-247  training_step: GradientTape around evaluate_single_graph(training=True), gradients of
                                 net_state divided by the iteration count when ``mean`` (:241), then apply_gradients
  GNN/GNN.py:180-199            evaluate_single_graph: loss_function(targets, out) * sample_weights, reduce_sum
  GNN/GNN.py:251-280            Loop with training=True: the SAME loop; the Keras layers switch behaviour:
  GNN/MLP.py:54-64              Dropout at dropout_pos (kept units scaled by 1/(1-rate)), BatchNormalization on batch statistics

Keras semantics not in the reference (parity unpinned, as for inference):
  Dropout(training=True): y = x * mask / (1 - rate), mask ~ Bernoulli(1 - rate) per element and per call -> masks are INJECTED;
  BatchNormalization(training=True) on [N, F]: batch mean, BIASED batch variance, y = gamma (x - mu) / sqrt(var + eps) + beta,
  moving <- moving * momentum + batch * (1 - momentum) at every call (so net_state's moving statistics move k times per step);
  categorical_crossentropy(from_logits=False): p = out / sum(out); p = clip(p, 1e-7, 1 - 1e-7); -sum(t log p);
  mean_squared_error: mean over the last axis of (out - t)^2.
"""
from __future__ import annotations

import numpy as np

from oracle import gnn_oracle as orc

EPS_K = 1e-7


# ---------------------------------------------------------------------------------------------------------------------
def act_forward(z, name):
    return orc.activation(z, name)


def act_backward(da, z, a, name):
    dt = z.dtype.type
    if name in ('linear', None):
        return da
    if name == 'relu':
        return da * (z > 0)
    if name == 'selu':
        return da * np.where(z > 0, dt(orc.SELU_SCALE), dt(orc.SELU_SCALE * orc.SELU_ALPHA) * np.exp(np.minimum(z, dt(0))))
    if name == 'elu':
        return da * np.where(z > 0, dt(1), np.exp(np.minimum(z, dt(0))))
    if name == 'tanh':
        return da * (dt(1) - a * a)
    if name == 'sigmoid':
        return da * a * (dt(1) - a)
    if name == 'softmax':
        return a * (da - np.sum(da * a, axis=-1, keepdims=True))
    raise ValueError(name)


def mlp_train_forward(x, net, masks, dtype=np.float64):
    """Training-mode forward of the Sequential of MLP.py:46-64.
    net: dict(weights, activations, batch_normalization, dropout={dense_index: rate}) - a Dropout sits in FRONT of Dense
    number dense_index; masks: {dense_index: 0/1 array shaped like that layer's input}.  Returns (y, cache)."""
    n = len(net['activations'])
    W = [np.asarray(net['weights'][2 * l], dtype) for l in range(n)]
    b = [np.asarray(net['weights'][2 * l + 1], dtype) for l in range(n)]
    h = np.asarray(x, dtype)
    cache = dict(h_in=[], z=[], a=[], scale={})
    for l in range(n):
        rate = net.get('dropout', {}).get(l)
        if rate and net.get('alphadropout'):
            # Keras AlphaDropout (MLP.py:59-61 with alphadropout=True): dropped units -> alpha' = -scale * alpha, then a x + b
            alpha_p = dtype(-1.0507009873554805 * 1.6732632423543772)
            a_ = dtype(1) / np.sqrt(dtype(1 - rate) * (dtype(1) + dtype(rate) * alpha_p ** 2))
            b_ = -a_ * alpha_p * dtype(rate)
            keep = np.asarray(masks[l], dtype)
            cache['scale'][l] = a_ * keep
            h = a_ * (h * keep + alpha_p * (dtype(1) - keep)) + b_
        elif rate:
            sc = np.asarray(masks[l], dtype) / dtype(1 - rate)
            cache['scale'][l] = sc
            h = h * sc
        z = h @ W[l] + b[l]
        a = act_forward(z, net['activations'][l])
        cache['h_in'].append(h); cache['z'].append(z); cache['a'].append(a)
        h = a
    if net['batch_normalization']:
        gamma, beta = (np.asarray(v, dtype) for v in net['weights'][2 * n:2 * n + 2])
        mu = h.mean(axis=0)
        var = ((h - mu) ** 2).mean(axis=0)
        inv = dtype(1) / np.sqrt(var + dtype(orc.BN_EPS))
        xhat = (h - mu) * inv
        cache.update(bn=(xhat, inv, gamma), batch_mean=mu, batch_var=var)
        h = gamma * xhat + beta
    return h, cache


def mlp_train_backward(dy, net, cache, dtype=np.float64):
    """Returns (dx, grads) with grads in get_weights() order: [dW1, db1, ..., (dgamma, dbeta)]."""
    n = len(net['activations'])
    W = [np.asarray(net['weights'][2 * l], dtype) for l in range(n)]
    grads_bn = []
    d = np.asarray(dy, dtype)
    if net['batch_normalization']:
        xhat, inv, gamma = cache['bn']
        m = xhat.shape[0]
        grads_bn = [np.sum(d * xhat, axis=0), np.sum(d, axis=0)]
        dxh = d * gamma
        d = inv / m * (m * dxh - dxh.sum(axis=0) - xhat * np.sum(dxh * xhat, axis=0))
    grads = [None] * (2 * n)
    for l in reversed(range(n)):
        dz = act_backward(d, cache['z'][l], cache['a'][l], net['activations'][l])
        grads[2 * l] = cache['h_in'][l].T @ dz
        grads[2 * l + 1] = dz.sum(axis=0)
        d = dz @ W[l].T
        if l in cache['scale']:
            d = d * cache['scale'][l]
    return d, grads + grads_bn


# ---------------------------------------------------------------------------------------------------------------------
def loss_forward_backward(kind, targets, out, weights, dtype=np.float64):
    """(sum_i w_i L(t_i, out_i), d/d out) for the Keras losses the starter can pass (starter.py:82)."""
    t, o, w = np.asarray(targets, dtype), np.asarray(out, dtype), np.asarray(weights, dtype)
    if kind == 'categorical_crossentropy':
        s = o.sum(axis=-1, keepdims=True)
        p = o / s
        pc = np.clip(p, dtype(EPS_K), dtype(1 - EPS_K))
        loss = -(t * np.log(pc)).sum(axis=-1)
        g = np.where((p >= EPS_K) & (p <= 1 - EPS_K), -t / pc, dtype(0))          # dL/dp (zero where clipped)
        do = (g - np.sum(g * p, axis=-1, keepdims=True)) / s
    elif kind == 'categorical_crossentropy_from_logits':
        z = o - o.max(axis=-1, keepdims=True)
        logp = z - np.log(np.exp(z).sum(axis=-1, keepdims=True))
        loss = -(t * logp).sum(axis=-1)
        do = np.exp(logp) * t.sum(axis=-1, keepdims=True) - t
    elif kind == 'mean_squared_error':
        loss = ((o - t) ** 2).mean(axis=-1)
        do = dtype(2) * (o - t) / o.shape[-1]
    else:
        raise ValueError(kind)
    return float(np.sum(loss * w)), do * w[:, None]


def train_forward(g, net_state, net_output, state_vect_dim, max_iteration, threshold, state0, masks_state, masks_output,
                  momentum=0.99, dtype=np.float64, edge_based=False):
    """Training-mode Loop (GNN.py:251-280 with training=True): returns the context the backward pass needs, with the
    node-level outputs ``out_nodes`` [M, T], the final ``state`` and the moving statistics after the k + 1 BN calls."""
    nodes = np.asarray(g['nodes'], dtype)
    n = nodes.shape[0]
    agg_arcs = orc.spmm_csr(g['arcT'], np.asarray(g['arcs'], dtype)[:, 2:], dtype)
    agg_nodes = np.zeros((n, 0), dtype)
    if state_vect_dim:
        state = np.asarray(state0, dtype)
        agg_nodes = orc.spmm_csr(g['adjT'], nodes, dtype)
    else:
        state = nodes.copy()
    state_old = np.ones_like(state)
    caches, k = [], 0
    n_st = len(net_state['activations'])
    mov_s = [np.asarray(v, dtype).copy() for v in net_state['weights'][2 * n_st + 2:2 * n_st + 4]] if net_state['batch_normalization'] else None
    while bool(np.any(orc.not_converged(state, state_old, threshold))) and k < max_iteration:
        comps = state if not state_vect_dim else np.concatenate([state, nodes], axis=1)
        inp = np.concatenate([comps, orc.spmm_csr(g['adjT'], state, dtype), agg_nodes, agg_arcs], axis=1)
        new, cache = mlp_train_forward(inp, net_state, masks_state[k], dtype)
        caches.append(cache)
        if mov_s is not None:
            mov_s[0] = mov_s[0] * momentum + cache['batch_mean'] * (1 - momentum)
            mov_s[1] = mov_s[1] * momentum + cache['batch_var'] * (1 - momentum)
        k, state, state_old = k + 1, new, state
    mask = np.logical_and(g['set_mask'], g['output_mask'])
    feats = state if not state_vect_dim else np.concatenate([state, nodes], axis=1)
    # GNNedgeBased.apply_filters (GNN.py:289-302): one row per masked arc, [F[dst] | F[src] | arc label]
    rows_in = orc.edge_features(g, state, state_vect_dim, dtype) if edge_based else feats[mask]
    out_nodes, cache_o = mlp_train_forward(rows_in, net_output, masks_output, dtype)
    n_ou = len(net_output['activations'])
    mov_o = None
    if net_output['batch_normalization']:
        mov_o = [np.asarray(v, dtype).copy() for v in net_output['weights'][2 * n_ou + 2:2 * n_ou + 4]]
        mov_o[0] = mov_o[0] * momentum + cache_o['batch_mean'] * (1 - momentum)
        mov_o[1] = mov_o[1] * momentum + cache_o['batch_var'] * (1 - momentum)
    return dict(g=g, net_state=net_state, net_output=net_output, D=state_vect_dim, k=k, caches=caches, cache_o=cache_o, mask=mask,
                state=state, out_nodes=out_nodes, moving_state=mov_s, moving_output=mov_o, dtype=dtype, edge_based=edge_based)


def train_backward(ctx, d_out_nodes, d_state_extra=None, want_d_arcs=False):
    """Back-propagation through net_output and the k executed bodies.  d_out_nodes: d loss / d out_nodes [M, T];
    d_state_extra: an additional gradient on the FINAL state [N, Ds] (LGNN: the next layer's labels contain it).
    Returns (grads_state summed over the iterations, grads_output, d_nodes [N, NL]): d_nodes is the gradient with
    respect to the node labels this layer saw (LGNN: they contain the previous layer's state / output).
    want_d_arcs: also return d loss / d arc labels [E, AL] in ORIGINAL arc order (edge-based LGNN: the arc labels contain the
    previous layer's output, LGNN.py:253-254): the label columns of the per-arc readout rows, and ArcNode^T . arc labels
    (GNN.py:259), a loop-invariant term that enters every body."""
    g, dtype, k = ctx['g'], ctx['dtype'], ctx['k']
    net_state, net_output, D = ctx['net_state'], ctx['net_output'], ctx['D']
    nodes = np.asarray(g['nodes'], dtype)
    n, nl = nodes.shape
    ds = ctx['state'].shape[1]
    mask = ctx['mask']
    d_feats, grads_o = mlp_train_backward(d_out_nodes, net_output, ctx['cache_o'], dtype)
    d_state = np.zeros((n, ds), dtype)
    d_nodes = np.zeros((n, nl), dtype)
    indptr, src, w = g['adjT']
    w = np.asarray(w, dtype)
    dst = np.repeat(np.arange(n), np.diff(indptr))
    if ctx.get('edge_based'):       # mask is over arcs; both endpoints of a masked arc receive gradient
        wn = ds + (nl if D else 0)
        np.add.at(d_state, dst[mask], d_feats[:, :ds])
        np.add.at(d_state, src[mask], d_feats[:, wn:wn + ds])
        if D:
            np.add.at(d_nodes, dst[mask], d_feats[:, ds:wn])
            np.add.at(d_nodes, src[mask], d_feats[:, wn + ds:2 * wn])
    else:
        d_state[mask] = d_feats[:, :ds]
        if D:
            d_nodes[mask] += d_feats[:, ds:]
    if d_state_extra is not None:
        d_state = d_state + np.asarray(d_state_extra, dtype)
    grads_s = None
    c_aggs = ds + (nl if D else 0)
    c_aggn = c_aggs + ds
    c_agga = c_aggn + (nl if D else 0)
    al = np.asarray(g['arcs']).shape[1] - 2
    d_arcs = np.zeros((np.asarray(g['arcs']).shape[0], al), dtype)
    d_agg_arcs = np.zeros((n, al), dtype)
    if want_d_arcs and ctx.get('edge_based'):
        wn_ = ds + (nl if D else 0)
        d_arcs[np.nonzero(mask)[0]] += d_feats[:, 2 * wn_:2 * wn_ + al]     # readout row m <-> arc position p (paired by position)
    for it in reversed(range(k)):
        d_inp, gk = mlp_train_backward(d_state, net_state, ctx['caches'][it], dtype)
        grads_s = gk if grads_s is None else [a + b for a, b in zip(grads_s, gk)]
        d_state = d_inp[:, :ds].copy()
        # aggregated_states = Adjacency^T . state  =>  d state[src] += w * d agg[dst]
        np.add.at(d_state, src, w[:, None] * d_inp[dst, c_aggs:c_aggs + ds])
        if D:       # node labels enter every iteration directly and through aggregated_nodes (GNN.py:228, :263)
            d_nodes += d_inp[:, ds:ds + nl]
            np.add.at(d_nodes, src, w[:, None] * d_inp[dst, c_aggn:c_aggn + nl])
        d_agg_arcs += d_inp[:, c_agga:c_agga + al]
    if not D:
        d_nodes = d_state           # state_0 = nodes (GNN.py:265)
    if want_d_arcs:
        ip_a, arc_id, w_a = g['arcT']
        dst_a = np.repeat(np.arange(n), np.diff(ip_a))
        d_arcs[arc_id] += np.asarray(w_a, dtype)[:, None] * d_agg_arcs[dst_a]
    if grads_s is None:
        n_st = len(net_state['activations'])
        grads_s = [np.zeros_like(np.asarray(v, dtype)) for v in net_state['weights'][:2 * n_st + (2 if net_state['batch_normalization'] else 0)]]
    if want_d_arcs:
        return grads_s, grads_o, d_nodes, d_arcs
    return grads_s, grads_o, d_nodes


def train_step(g, net_state, net_output, state_vect_dim, max_iteration, threshold, state0, masks_state, masks_output,
               targets, sample_weights, loss='categorical_crossentropy', mean=True, graph_based=False, momentum=0.99,
               dtype=np.float64, edge_based=False):
    """One training_step (GNN_BaseClass.py:231-247) without the optimizer.

    masks_state: list (one per possible iteration) of {dense_index: mask [N, width]}; masks_output: {dense_index: mask [M, width]}.
    Returns dict(k, loss, grads_state, grads_output, moving_state=(mean, var), moving_output=(mean, var), out)."""
    ctx = train_forward(g, net_state, net_output, state_vect_dim, max_iteration, threshold, state0, masks_state, masks_output,
                        momentum, dtype, edge_based)
    out = ctx['out_nodes']
    if graph_based:
        ng = np.asarray(g['NodeGraph'], dtype)
        out = ng.T @ out
    loss_value, d_out = loss_forward_backward(loss, targets, out, sample_weights, dtype)
    if graph_based:
        d_out = ng @ d_out
    grads_s, grads_o, _ = train_backward(ctx, d_out)
    k = ctx['k']
    if mean and k:
        grads_s = [gv / k for gv in grads_s]
    return dict(k=float(k), loss=loss_value, grads_state=grads_s, grads_output=grads_o, moving_state=ctx['moving_state'],
                moving_output=ctx['moving_output'], out=out, state=ctx['state'])


def lgnn_train_step(g, layers, get_state, get_output, training_mode, state0, masks_state, masks_output, targets, sample_weights,
                    loss='categorical_crossentropy', mean=True, graph_based=False, dtype=np.float64, edge_based=False):
    """Joint training step of an LGNN in 'parallel' or 'residual' mode (LGNN.py:201-224 inside GNN_BaseClass.py:231-247):
    the tape spans the whole stack, so layer i also receives gradient through the labels of layer i + 1
    (update_graph, LGNN.py:227-260: [nodes | state_i? | scatter(out_i)?]).

    layers: list of dict(net_state, net_output, state_vect_dim, max_iteration, threshold); state0 / masks_*: one entry per layer.
    Returns dict(k=[...], loss, grads_state=[...], grads_output=[...], outs=[...])."""
    assert training_mode in ('parallel', 'residual')
    L = len(layers)
    nodes0 = np.asarray(g['nodes'], dtype)
    arcs0 = np.asarray(g['arcs'], dtype)
    nlb, alb = nodes0.shape[1], arcs0.shape[1] - 2
    ctxs, outs = [], []
    cur = g
    ng = np.asarray(g['NodeGraph'], dtype) if graph_based else None
    for i, ly in enumerate(layers):
        ctx = train_forward(cur, ly['net_state'], ly['net_output'], ly['state_vect_dim'], ly['max_iteration'], ly['threshold'],
                            state0[i], masks_state[i], masks_output[i], dtype=dtype, edge_based=edge_based)
        ctxs.append(ctx)
        outs.append(ng.T @ ctx['out_nodes'] if graph_based else ctx['out_nodes'])
        if i < L - 1:
            extra = []
            cur = dict(g)
            if get_state: extra.append(ctx['state'])
            if get_output:
                sc = np.zeros((len(ctx['mask']), ctx['out_nodes'].shape[1]), dtype)
                sc[ctx['mask']] = ctx['out_nodes']
                if edge_based: cur['arcs'] = np.concatenate([arcs0, sc], axis=1)      # LGNN.py:253-254
                else: extra.append(sc)
            cur['nodes'] = np.concatenate([nodes0] + extra, axis=1)
    if training_mode == 'residual':
        loss_value, d = loss_forward_backward(loss, targets, np.mean(outs, axis=0), sample_weights, dtype)
        d_outs = [d / L] * L
    else:
        pairs = [loss_forward_backward(loss, targets, o, sample_weights, dtype) for o in outs]
        loss_value = float(np.mean([p[0] for p in pairs]))
        d_outs = [p[1] / L for p in pairs]
    grads_s, grads_o = [None] * L, [None] * L
    d_state_extra = d_out_extra = None
    for i in reversed(range(L)):
        ctx = ctxs[i]
        d_nodes_out = ng @ d_outs[i] if graph_based else d_outs[i]
        if d_out_extra is not None:
            d_nodes_out = d_nodes_out + d_out_extra
        if edge_based:
            gs, go, d_nodes, d_arcs = train_backward(ctx, d_nodes_out, d_state_extra, want_d_arcs=True)
        else:
            gs, go, d_nodes = train_backward(ctx, d_nodes_out, d_state_extra)
        if mean and ctx['k']:
            gs = [v / ctx['k'] for v in gs]
        grads_s[i], grads_o[i] = gs, go
        if i > 0:
            prev = ctxs[i - 1]
            c = nlb
            d_state_extra = d_out_extra = None
            if get_state:
                d_state_extra = d_nodes[:, c:c + prev['state'].shape[1]]
                c += prev['state'].shape[1]
            if get_output:
                if edge_based: d_out_extra = d_arcs[prev['mask'], alb:alb + prev['out_nodes'].shape[1]]
                else: d_out_extra = d_nodes[prev['mask'], c:c + prev['out_nodes'].shape[1]]
    return dict(k=[float(c['k']) for c in ctxs], loss=loss_value, grads_state=grads_s, grads_output=grads_o, outs=outs)


def adam_update(params, grads, m, v, step, lr=0.001, beta1=0.9, beta2=0.999, eps=1e-7):
    """Keras Adam (non-amsgrad): lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p -= lr_t m / (sqrt(v) + eps)."""
    lr_t = lr * np.sqrt(1 - beta2 ** step) / (1 - beta1 ** step)
    out = []
    for i, (p, gr) in enumerate(zip(params, grads)):
        m[i] = beta1 * m[i] + (1 - beta1) * gr
        v[i] = beta2 * v[i] + (1 - beta2) * gr * gr
        out.append(p - lr_t * m[i] / (np.sqrt(v[i]) + eps))
    return out
