"""Layered GNN with the reference's public surface (``GNN/LGNN.py``): a stack of GNNs in which layer i+1 sees the
ORIGINAL node labels widened by layer i's state and/or output (reference LGNN.py:227-290).

The whole stack stays on the MI355X: each layer is one ``gnn_loop_run``; the relabelling between layers
(``update_graph``: concat + scatter through the mask) is ``gnn_graph_update_labels`` on device-resident graphs that share
one CSR.
"""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

from GNN.GNN import GNNnodeBased, GNNgraphBased, GNNedgeBased
from GNN.GNN_BaseClass import BaseClass
from GNN.graph_class import GraphObject, GraphTensor


class LGNN(BaseClass):
    def __init__(self, gnns: list, get_state: bool, get_output: bool, optimizer, loss_function, loss_arguments: Optional[dict],
                 addressed_problem: str, extra_metrics: Optional[dict] = None, extra_metrics_arguments: Optional[dict[str, dict]] = None,
                 path_writer: str = 'writer/', namespace: str = 'LGNN') -> None:
        kinds = {type(i) for i in gnns}
        if len(kinds) != 1: raise TypeError('parameter <gnn> must contain gnns of the same type')
        super().__init__(optimizer, loss_function, loss_arguments, addressed_problem, extra_metrics, extra_metrics_arguments,
                         path_writer, namespace)
        self.get_state = get_state
        self.get_output = get_output
        self.gnns = gnns
        self.LAYERS = len(gnns)
        self.GNNS_TYPE = kinds.pop()
        self.namespace = [f'{namespace} - GNN{i}' for i in range(self.LAYERS)]
        self.training_mode = None
        for gnn, name in zip(self.gnns, self.namespace):
            gnn.namespace = [name]
            gnn.path_writer = f'{self.path_writer}{name}/'

    def copy(self, *, path_writer: str = '', namespace: str = '', copy_weights: bool = True) -> 'LGNN':
        optimizer = self.optimizer
        if hasattr(optimizer, 'get_config'):
            optimizer = optimizer.__class__(**optimizer.get_config())
        return self.__class__(gnns=[i.copy(copy_weights=copy_weights) for i in self.gnns], get_state=self.get_state,
                              get_output=self.get_output, optimizer=optimizer, loss_function=self.loss_function,
                              loss_arguments=self.loss_args, addressed_problem=self.addressed_problem,
                              extra_metrics=self.extra_metrics, extra_metrics_arguments=self.mt_args,
                              path_writer=path_writer or self.path_writer + '_copied/', namespace=namespace or 'LGNN')

    def save(self, path: str):
        """Save the stack to folder <path>: one sub-folder per GNN + config.json (reference LGNN.py:83-103)."""
        import json, os
        from GNN import losses, optimizers
        from GNN.GNN import GNNnodeBased, GNNedgeBased, GNNgraphBased
        if path[-1] != '/': path += '/'
        for i, gnn in enumerate(self.gnns): gnn.save(f'{path}GNN{i}/')
        os.makedirs(path, exist_ok=True)
        gnns_type = {GNNnodeBased: 'n', GNNedgeBased: 'a', GNNgraphBased: 'g'}[type(self.gnns[0])]
        loss_name = getattr(self.loss_function, '__name__', None)
        with open(f'{path}config.json', 'w') as f:
            json.dump({'get_state': self.get_state, 'get_output': self.get_output, 'loss_function': loss_name if hasattr(losses, str(loss_name)) else None,
                       'loss_arguments': self.loss_args, 'optimizer': optimizers.serialize(self.optimizer),
                       'addressed_problem': self.addressed_problem, 'gnns_type': gnns_type}, f)

    @classmethod
    def load(cls, path: str, path_writer=None, namespace: str = 'LGNN', extra_metrics=None, extra_metrics_arguments=None):
        """Load a stack saved by save() (reference LGNN.py:106-141, same arguments)."""
        import json, os
        from GNN import losses, optimizers
        from GNN.GNN import GNNnodeBased, GNNedgeBased, GNNgraphBased
        if path[-1] != '/': path += '/'
        if path_writer is None: path_writer = f'{path}writer'
        with open(f'{path}config.json') as f:
            config = json.load(f)
        optz = optimizers.deserialize(config.pop('optimizer', None))
        loss_name = config.pop('loss_function', None)
        loss = getattr(losses, loss_name) if loss_name else None
        gnn_cls = {'n': GNNnodeBased, 'a': GNNedgeBased, 'g': GNNgraphBased}[config.pop('gnns_type')]
        folders = sorted((d for d in os.listdir(path) if d.startswith('GNN') and os.path.isdir(f'{path}{d}')), key=lambda d: int(d[3:]))
        gnns = [gnn_cls.load(f'{path}{d}', path_writer=f'{path_writer}{namespace} - {d}/', namespace='GNN') for d in folders]
        return cls(gnns=gnns, optimizer=optz, loss_function=loss, extra_metrics=extra_metrics,
                   extra_metrics_arguments=extra_metrics_arguments, path_writer=path_writer, namespace=namespace, **config)

    # ---- weights ------------------------------------------------------------------------------------------------------
    def get_dense_layers(self):
        return [layer for gnn in self.gnns for layer in gnn.get_dense_layers()]

    def trainable_variables(self):
        return [g.net_state.trainable_variables for g in self.gnns], [g.net_output.trainable_variables for g in self.gnns]

    def get_weights(self):
        return [g.net_state.get_weights() for g in self.gnns], [g.net_output.get_weights() for g in self.gnns]

    def set_weights(self, weights_state, weights_output) -> None:
        assert len(weights_state) == len(weights_output) == self.LAYERS
        for gnn, ws, wo in zip(self.gnns, weights_state, weights_output):
            gnn.net_state.set_weights(ws)
            gnn.net_output.set_weights(wo)

    # ---- inference ------------------------------------------------------------------------------------------------------
    def __call__(self, g: Union[GraphObject, GraphTensor]):
        return self.Loop(g, training=False)[-1][-1]

    def predict(self, g: Union[GraphObject, GraphTensor], idx: Union[int, list[int], range, str] = -1):
        """Output(s) of the chosen layer(s) (reference LGNN.py:172-198)."""
        layers = range(self.LAYERS)
        if isinstance(idx, int):
            assert idx in layers
        elif isinstance(idx, (list, range)):
            assert all(i in layers for i in idx)
            idx = sorted(idx)
        elif idx == 'all':
            idx = layers
        else:
            raise ValueError('param <idx> must be 1.int; 2.list of ordered ints in range(self.LAYERS); 3. str "all"')
        out = self.Loop(g, training=False)[-1]
        return out[idx] if isinstance(idx, int) else [out[i] for i in idx]

    def evaluate_single_graph(self, g: Union[GraphObject, GraphTensor], training: bool) -> tuple:
        """Parallel mode: mean of the per-layer losses; residual (training only): loss of the mean output
        (reference LGNN.py:201-224)."""
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        targs = self.GNNS_TYPE.get_filtered_tensor(g, g.targets)
        loss_weights = self.GNNS_TYPE.get_filtered_tensor(g, g.sample_weights)
        it, _, out = self.Loop(g, training=training)
        if training and self.training_mode == 'residual':
            loss = self.loss_function(targs, np.mean(out, axis=0), **self.loss_args) * loss_weights
        else:
            loss = np.mean([self.loss_function(targs, o, **self.loss_args) * loss_weights for o in out], axis=0)
        return it, np.sum(loss), targs, out[-1]

    def update_graph(self, g: GraphTensor, state, output) -> GraphTensor:
        """Host form of the relabelling (reference LGNN.py:227-260); Loop uses the device form instead.  Node/graph-based
        layers put the scattered output on the node labels, edge-based ones on the arc labels (:253-256)."""
        g = g.copy()
        extra = []
        if self.get_state: extra.append(np.asarray(state, dtype=np.float32))
        if self.get_output:
            mask = g.loop_mask()
            scattered = np.zeros((len(mask), output.shape[1]), dtype=np.float32)
            scattered[np.nonzero(mask)[0]] = output
            if self.GNNS_TYPE == GNNedgeBased:
                g.arcs = np.concatenate([g.arcs, scattered], axis=1)
            else:
                extra.append(scattered)
        g.nodes = np.concatenate([g.nodes] + extra, axis=1)
        g._device_graph = None
        return g

    def Loop(self, g: Union[GraphObject, GraphTensor], *, training: bool = False, state0=None):
        """(K list, last state, outs list) as reference LGNN.py:263-290.  ``state0``: optional list of injected initial
        states, one per layer."""
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        graph_based = self.GNNS_TYPE == GNNgraphBased
        edge_based = self.GNNS_TYPE == GNNedgeBased
        if graph_based:
            if g.NodeGraph is None: raise ValueError('WRONG GNN. NodeGraph is None: GNN is graph-based, while problem is non graph-based.')
            if not g.loop_mask().all(): raise ValueError('graph-based GNN needs set_mask and output_mask all True')
        state0 = state0 or [None] * self.LAYERS
        base = g.device_graph(self.gnns[0].device)
        if edge_based and not g.__dict__.get('_arc_order_set'):
            base.set_arc_order(g.ArcNode[1], g.arcs[:, 2:])        # what the arc side of the relabelling needs
            g._arc_order_set = True
        current = base
        derived = g.__dict__.setdefault('_lgnn_graphs', {})
        K, outs = [], []
        loop = None
        for idx, gnn in enumerate(self.gnns):
            if edge_based:
                loop = gnn._device_loop(current)
                gnn._prepare_loop(g, loop, own_labels=current is not base)
                if gnn.state_vect_dim > 0: loop.set_state0(state0[idx], gnn.seed)
                k = gnn._train_forward(loop) if training else loop.run(False)
            else:
                k, loop = gnn._run(current, training, state0[idx])
            K.append(k)
            last = idx == self.LAYERS - 1
            outs.append(loop.readout(*g.nodegraph_csr()) if graph_based else loop.output())
            if not last:
                # relabel from the ORIGINAL graph (reference LGNN.py:287): nodes [nodes | state? | scattered output?], or, for
                # edge-based layers, nodes [nodes | state?] and arcs [arc labels | scattered output?]
                if edge_based:
                    key = ('a', self.get_state * loop.Ds, self.get_output * loop.T)
                    nxt = derived.get(key)
                    if nxt is None:
                        nxt = derived[key] = base.derive_edge(key[1], key[2])
                else:
                    key = self.get_state * loop.Ds + self.get_output * loop.T
                    nxt = derived.get(key)
                    if nxt is None:
                        nxt = derived[key] = base.derive(key)
                nxt.update_labels(base, loop, self.get_state, self.get_output)
                current = nxt
        return K, loop.state(), outs

    # ---- training ---------------------------------------------------------------------------------------------------------
    def train(self, gTr, epochs: int, gVa=None, update_freq: int = 10, max_fails: int = 10, observed_metric: str = 'Loss',
              policy='min', *, mean: bool = True, training_mode: str = 'parallel', verbose: int = 3) -> None:
        assert training_mode in ['parallel', 'serial', 'residual']
        if (self.training_mode is not None) and (self.training_mode != training_mode): raise ValueError
        self.training_mode = training_mode
        if training_mode == 'serial':
            # layers are trained one after the other on graphs relabelled by the previous layer (reference LGNN.py:325-340)
            gTr1, gVa1 = self.checktype(gTr), self.checktype(gVa)
            gTr0, gVa0 = gTr1, gVa1
            for idx, gnn in enumerate(self.gnns):
                if verbose in [1, 3]: print(f'\n\n------------------- GNN{idx} -------------------\n')
                gnn.train(gTr1, epochs, gVa1, update_freq, max_fails, observed_metric, policy, mean=mean, verbose=verbose)
                node_loop = (lambda g: gnn.Loop(g)) if self.GNNS_TYPE == GNNedgeBased else (lambda g: GNNnodeBased.Loop(gnn, g))
                gTr1 = [self.update_graph(g, *node_loop(gt)[1:]) for g, gt in zip(gTr0, gTr1)]
                if gVa0: gVa1 = [self.update_graph(g, *node_loop(gt)[1:]) for g, gt in zip(gVa0, gVa1)]
            return
        # 'parallel' / 'residual': one joint step per batch (reference LGNN.py:343-344 -> GNN_BaseClass.train)
        super().train(gTr, epochs, gVa, update_freq, max_fails, observed_metric, policy, mean=mean, verbose=verbose)

    def training_step(self, g: GraphTensor, mean: bool, *, state0=None, masks_state=None, masks_output=None) -> dict:
        """Joint step of the whole stack (reference GNN_BaseClass.py:231-247 around LGNN.py:201-224): training-mode forward
        of every layer (``gnn_loop_train_forward``), loss = mean of the per-layer losses ('parallel') or loss of the mean
        output ('residual'), then the backward passes from the last layer to the first (``gnn_loop_train_backward``); layer
        i also receives the gradient that reaches it through the labels of layer i + 1 (update_graph, LGNN.py:227-260).
        ``state0`` / ``masks_*``: optional per-layer lists (tests inject them)."""
        from GNN import losses, _engine
        edge_based = self.GNNS_TYPE == GNNedgeBased
        if self.training_mode not in ('parallel', 'residual'):
            raise ValueError("training_step is the joint step of training_mode 'parallel' / 'residual'")
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        if self.optimizer is None or not hasattr(self.optimizer, 'apply_gradients'):
            raise TypeError('train() needs an optimizer with apply_gradients, e.g. GNN.optimizers.Adam()')
        kind = losses.device_loss_kind(self.loss_function, self.loss_args)
        graph_based = self.GNNS_TYPE == GNNgraphBased
        if graph_based and not g.loop_mask().all():
            raise ValueError('graph-based GNN needs set_mask and output_mask all True')
        L = self.LAYERS
        state0 = state0 or [None] * L
        masks_state = masks_state or [None] * L
        masks_output = masks_output or [None] * L
        targets = self.GNNS_TYPE.get_filtered_tensor(g, g.targets)
        weights = self.GNNS_TYPE.get_filtered_tensor(g, g.sample_weights)
        base = g.device_graph(self.gnns[0].device)
        if edge_based and not g.__dict__.get('_arc_order_set'):
            base.set_arc_order(g.ArcNode[1], g.arcs[:, 2:])
            g._arc_order_set = True
        derived = g.__dict__.setdefault('_lgnn_graphs', {})
        mask = g.loop_mask()
        NLb, ALb = np.asarray(g.nodes).shape[1], np.asarray(g.arcs).shape[1] - 2
        # ---- forward ----
        current, loops, K, outs = base, [], [], []
        for idx, gnn in enumerate(self.gnns):
            loop = gnn._device_loop(current)
            if edge_based: gnn._prepare_loop(g, loop, own_labels=current is not base)
            if gnn.state_vect_dim > 0:
                gnn.seed += 1
                loop.set_state0(state0[idx], gnn.seed)
            gnn._train_calls = getattr(gnn, '_train_calls', 0) + 1
            k, out_nodes = loop.train_forward(gnn.net_state.device_mlp(gnn.device), gnn.net_output.device_mlp(gnn.device), None,
                                              dropout_state=gnn.net_state.dropout_rates(), dropout_output=gnn.net_output.dropout_rates(),
                                              masks_state=masks_state[idx], masks_output=masks_output[idx],
                                              seed=gnn.seed * 1000003 + gnn._train_calls, bn_state=None, bn_output=None)     # gamma / beta: the device copies
            loops.append(loop); K.append(k)
            outs.append(loop.readout(*g.nodegraph_csr()) if graph_based else out_nodes)
            if idx < L - 1:
                if edge_based:
                    key = ('a', self.get_state * loop.Ds, self.get_output * loop.T)
                    nxt = derived.get(key)
                    if nxt is None:
                        nxt = derived[key] = base.derive_edge(key[1], key[2])
                else:
                    key = self.get_state * loop.Ds + self.get_output * loop.T
                    nxt = derived.get(key)
                    if nxt is None:
                        nxt = derived[key] = base.derive(key)
                nxt.update_labels(base, loop, self.get_state, self.get_output)
                current = nxt
        # ---- loss (reference LGNN.py:219-222) ----
        if self.training_mode == 'residual':
            loss, d = _engine.loss_grad(kind, targets, np.mean(outs, axis=0, dtype=np.float32), weights)
            d_outs = [d / L] * L
        else:
            pairs = [_engine.loss_grad(kind, targets, o, weights) for o in outs]
            loss = float(np.mean([p[0] for p in pairs]))
            d_outs = [p[1] / L for p in pairs]
        # ---- backward, last layer first ----
        ng = np.asarray(g.NodeGraph, dtype=np.float32) if graph_based else None
        results = [None] * L
        d_state_extra = d_out_extra = None
        for idx in reversed(range(L)):
            d_nodes_out = ng @ d_outs[idx] if graph_based else d_outs[idx]
            if d_out_extra is not None: d_nodes_out = d_nodes_out + d_out_extra
            chain = idx > 0
            res = results[idx] = loops[idx].train_backward(d_nodes_out, d_state_extra,
                                                           want_d_nodes=chain and (self.get_state or (self.get_output and not edge_based)),
                                                           want_d_arcs=chain and edge_based and self.get_output)
            d_state_extra = d_out_extra = None
            if chain:
                prev, c = loops[idx - 1], NLb
                if self.get_state:
                    d_state_extra = res['d_nodes'][:, c:c + prev.Ds]
                    c += prev.Ds
                if self.get_output:     # the previous output sits on the arc labels of an edge-based layer, else on the node labels
                    d_out_extra = res['d_arcs'][mask, ALb:ALb + prev.T] if edge_based else res['d_nodes'][mask, c:c + prev.T]
        # ---- update: net_state gradients / k when mean (GNN_BaseClass.py:241); one optimizer over all layers (:244-247) ----
        from GNN import regularizers
        # Device-side update (gnn_loop_optimizer_step per layer: gradients, weights and optimizer slots stay in HBM) when nothing of
        # it lives on the host: an optimizer that knows the engine's update rules and no kernel / bias regularizers
        if (getattr(self, 'device_optimizer', True) and hasattr(self.optimizer, 'device_step_args')
                and not regularizers.any_regularizer(self.get_dense_layers())):
            kind_o, hyper = self.optimizer.device_step_args()           # one optimizer step over all layers (reference :244-247)
            for gnn, loop, k in zip(self.gnns, loops, K):
                gnn.net_state.bind_optimizer(self.optimizer)
                gnn.net_output.bind_optimizer(self.optimizer)
                loop.optimizer_step(kind_o, hyper, (1.0 / k) if (mean and k) else 1.0, getattr(gnn.net_state.layers[-1], 'momentum', 0.99),
                                    getattr(gnn.net_output.layers[-1], 'momentum', 0.99))
                gnn.net_state.mark_device_newer()
                gnn.net_output.mark_device_newer()
            self.optimizer.device_step_done()
            return dict(loss=loss, k=K, grads_state=[r['grads_state'] for r in results], grads_output=[r['grads_output'] for r in results], outs=outs)
        for gnn, r in zip(self.gnns, results):     # regularizer terms of the taped loss (reference GNN_BaseClass.py:223-235)
            for net, key in ((gnn.net_state, 'grads_state'), (gnn.net_output, 'grads_output')):
                pen, rg = regularizers.penalty_and_gradients(net.dense_layers)
                loss += pen
                for li, (gk, gb) in enumerate(rg):
                    if gk is not None: r[key][2 * li] = r[key][2 * li] + gk
                    if gb is not None: r[key][2 * li + 1] = r[key][2 * li + 1] + gb
        gs = [[a / k for a in r['grads_state']] if (mean and k) else r['grads_state'] for r, k in zip(results, K)]
        go = [r['grads_output'] for r in results]
        ws, wo = self.trainable_variables()
        dW = [a for layer in gs + go for a in layer]
        W = [a for layer in ws + wo for a in layer]
        assert len(dW) == len(W)
        new = self.optimizer.apply_gradients(zip(dW, W))
        pos = 0
        for net in [gnn.net_state for gnn in self.gnns] + [gnn.net_output for gnn in self.gnns]:
            n = len(net.trainable_variables)
            net.set_trainable(new[pos:pos + n])
            pos += n
        for gnn, r, loop in zip(self.gnns, results, loops):
            gnn.net_state.update_moving_statistics(r['bn_batch_state'])
            if loop.n_masked: gnn.net_output.update_moving_statistics(r['bn_batch_output'])
        return dict(loss=loss, k=K, grads_state=[r['grads_state'] for r in results], grads_output=go, outs=outs)
