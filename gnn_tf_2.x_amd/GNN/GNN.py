"""GNN models with the reference's public surface (``GNN/GNN.py``), running ``Loop`` on the MI355X.

``Loop(g)`` = reference GNN.py:251-280: loop-invariant aggregates, iterate ``state <- net_state([state | labels |
Adjacency^T.state | aggregated labels])`` while some node moved by more than ``threshold`` (relative L2, strict '>') and
``k < max_iteration``, then ``net_output`` on the rows selected by ``set_mask & output_mask``.  The whole loop is one
call into ``libgnn_hip.so`` (``gnn_loop_run``); Python only converts the graph once and copies results back.

Extensions over the reference signature (needed because TensorFlow's RNG stream cannot be reproduced):
``Loop(g, training=False, state0=None)`` takes an injected initial state for ``state_vect_dim > 0``; without it the
engine draws N(0, 0.1^2) from its own generator seeded by ``self.seed``.
"""
from __future__ import annotations

import weakref
from typing import Optional, Union

import numpy as np

from GNN import _engine
from GNN.GNN_BaseClass import BaseClass
from GNN.MLP import Sequential, clone_model
from GNN.graph_class import GraphObject, GraphTensor


class GNNnodeBased(BaseClass):
    """GNN for node-based problems."""

    def __init__(self, net_state: Sequential, net_output: Sequential, optimizer, loss_function, loss_arguments: Optional[dict],
                 state_vect_dim: int, max_iteration: int, threshold: float, addressed_problem: str,
                 extra_metrics: Optional[dict] = None, extra_metrics_arguments: Optional[dict[str, dict]] = None,
                 path_writer: str = 'writer/', namespace: str = 'GNN') -> None:
        if not isinstance(state_vect_dim, int) or state_vect_dim < 0: raise TypeError('param <state_vect_dim> must be int>=0')
        super().__init__(optimizer, loss_function, loss_arguments, addressed_problem, extra_metrics, extra_metrics_arguments,
                         path_writer, namespace)
        self.net_state = net_state
        self.net_output = net_output
        self.max_iteration = max_iteration
        self.state_threshold = threshold
        self.state_vect_dim = state_vect_dim
        self.seed = 0
        self.device = 0
        self.impl = 2           # 2: fused kernel, bf16-split MFMA (fastest, fp32-accurate); 1: fused, bit-exact f32 MFMA; 0: one kernel per TF op

    # ---- copies, weights ---------------------------------------------------------------------------------------------
    def copy(self, *, path_writer: str = '', namespace: str = '', copy_weights: bool = True):
        optimizer = self.optimizer
        if hasattr(optimizer, 'get_config'):
            optimizer = optimizer.__class__(**optimizer.get_config())
        return self.__class__(net_state=clone_model(self.net_state, copy_weights), net_output=clone_model(self.net_output, copy_weights),
                              optimizer=optimizer, loss_function=self.loss_function, loss_arguments=self.loss_args,
                              max_iteration=self.max_iteration, threshold=self.state_threshold,
                              addressed_problem=self.addressed_problem, extra_metrics=self.extra_metrics,
                              extra_metrics_arguments=self.mt_args, state_vect_dim=self.state_vect_dim,
                              path_writer=path_writer or self.path_writer + '_copied/', namespace=namespace or 'GNN')

    def save(self, path: str):
        """Save the model to folder <path>, without extra_metrics (reference GNN.py:93-111; the reference writes two Keras
        SavedModels, here: architecture + weights of each net as JSON + .npz)."""
        import json, os
        from GNN import losses, optimizers
        from GNN.MLP import sequential_config
        if path[-1] != '/': path += '/'
        os.makedirs(path, exist_ok=True)
        for name, net in (('net_state', self.net_state), ('net_output', self.net_output)):
            np.savez(f'{path}{name}.npz', *net.get_weights())
            with open(f'{path}{name}.json', 'w') as f:
                json.dump(sequential_config(net), f)
        loss_name = getattr(self.loss_function, '__name__', None)
        with open(f'{path}config.json', 'w') as f:
            json.dump({'loss_function': loss_name if hasattr(losses, str(loss_name)) else None, 'loss_arguments': self.loss_args,
                       'optimizer': optimizers.serialize(self.optimizer), 'max_iteration': self.max_iteration, 'threshold': self.state_threshold,
                       'addressed_problem': self.addressed_problem, 'state_vect_dim': self.state_vect_dim}, f)

    @classmethod
    def load(cls, path: str, path_writer=None, namespace: str = 'GNN', extra_metrics=None, extra_metrics_arguments=None):
        """Load a model saved by save() (reference GNN.py:114-149, same arguments): the GNN type is the calling class."""
        import json
        from GNN import losses, optimizers
        from GNN.MLP import sequential_from_config
        if path[-1] != '/': path += '/'
        if path_writer is None: path_writer = f'{path}writer'
        with open(f'{path}config.json') as f:
            config = json.load(f)
        optz = optimizers.deserialize(config.pop('optimizer', None))
        loss_name = config.pop('loss_function', None)
        loss = getattr(losses, loss_name) if loss_name else None
        nets = []
        for name in ('net_state', 'net_output'):
            with open(f'{path}{name}.json') as f:
                arch = json.load(f)
            with np.load(f'{path}{name}.npz') as z:
                weights = [z[f'arr_{i}'] for i in range(len(z.files))]
            nets.append(sequential_from_config(arch, weights))
        return cls(net_state=nets[0], net_output=nets[1], optimizer=optz, loss_function=loss, extra_metrics=extra_metrics,
                   extra_metrics_arguments=extra_metrics_arguments, path_writer=path_writer, namespace=namespace, **config)

    def get_dense_layers(self):
        return self.net_state.dense_layers + self.net_output.dense_layers

    def trainable_variables(self):
        return [self.net_state.trainable_variables], [self.net_output.trainable_variables]

    def get_weights(self):
        return [self.net_state.get_weights()], [self.net_output.get_weights()]

    def set_weights(self, weights_state, weights_output) -> None:
        assert len(weights_state) == len(weights_output) == 1
        self.net_state.set_weights(weights_state[0])
        self.net_output.set_weights(weights_output[0])

    # ---- inference -----------------------------------------------------------------------------------------------------
    def __call__(self, g: Union[GraphObject, GraphTensor]):
        return self.Loop(g, training=False)[-1]

    def evaluate_single_graph(self, g: Union[GraphObject, GraphTensor], training: bool) -> tuple:
        """(iterations, summed loss, targets, output) of one graph (reference GNN.py:180-199)."""
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        targs = self.get_filtered_tensor(g, g.targets)
        loss_weights = self.get_filtered_tensor(g, g.sample_weights)
        it, _, out = self.Loop(g, training=training, _want_state=False)      # the state (N x Ds floats) is not needed here: stays on the device
        loss = self.loss_function(targs, out, **self.loss_args) * loss_weights
        return it, np.sum(loss), targs, out

    def _device_loop(self, dev_graph: _engine.Graph) -> _engine.Loop:
        """One gnn_loop per (this model, device graph), created on first use."""
        cache = dev_graph.__dict__.setdefault('_loops', {})
        # the C loop is configured once: a changed max_iteration / state_threshold / state_vect_dim needs a new one
        key = (id(self), int(self.max_iteration), float(self.state_threshold), int(self.state_vect_dim))
        owner, loop = cache.get(key, (None, None))
        if owner is None or owner() is not self:
            for old in [k for k in cache if k[0] == id(self)]:
                del cache[old]
            loop = _engine.Loop(dev_graph, self.net_state.device_mlp(self.device), self.net_output.device_mlp(self.device),
                                self.state_vect_dim, self.max_iteration, self.state_threshold)
            cache[key] = (weakref.ref(self), loop)
        if getattr(loop, '_impl_set', None) != self.impl:      # (gnn_loop_set_impl checks the fused path, which re-packs the weight images when the weights
            loop.set_impl(self.impl)                          #  have changed - 0.17 ms after every optimizer step; the next inference Loop packs them anyway)
            loop._impl_set = self.impl
        return loop

    def _prerun(self, graphs) -> list:
        """evaluate() over several graphs (reference GNN_BaseClass.py:165-189 walks them one after the other): their device Loops in ONE
        gnn_loop_run_many call - the persistent launches of small graphs run side by side - and the Loop() calls that follow take these
        results instead of running again.  Returns the loops it marked (evaluate clears the marks when it is done)."""
        if len(graphs) < 2 or not all(isinstance(g, GraphTensor) for g in graphs):
            return []
        loops = [self._device_loop(g.device_graph(self.device)) for g in graphs]
        if len({id(lp) for lp in loops}) != len(loops):
            return []                                       # the same graph twice in the list: one after the other, as before
        if self.state_vect_dim > 0:
            for lp in loops: lp.set_state0(None, self.seed)
        for lp, k in zip(loops, _engine.Loop.run_many(loops)):
            lp._fresh = k
        return loops

    def _run(self, dev_graph: _engine.Graph, training: bool, state0) -> tuple[float, _engine.Loop]:
        loop = self._device_loop(dev_graph)
        if not training and state0 is None and getattr(loop, '_fresh', None) is not None:
            k, loop._fresh = loop._fresh, None              # run by _prerun a moment ago, with these weights and this initial state
            return k, loop
        if self.state_vect_dim > 0:
            loop.set_state0(state0, self.seed)
        if training:
            return self._train_forward(loop), loop
        return loop.run(False), loop

    def _train_forward(self, loop) -> float:
        """Loop(training=True) (reference GNN.py:251-280 with the Keras layers in training mode): Dropout with fresh masks,
        BatchNormalization on batch statistics (the moving statistics are updated as Keras does on every training-mode call).
        The loop then holds the training-mode state / outputs."""
        self._train_calls = getattr(self, '_train_calls', 0) + 1
        # gamma / beta are read from the device copies (bn_state = None) and the moving statistics are updated there as well
        k, _ = loop.train_forward(self.net_state.device_mlp(self.device), self.net_output.device_mlp(self.device), None,
                                  dropout_state=self.net_state.dropout_rates(), dropout_output=self.net_output.dropout_rates(),
                                  seed=self.seed * 1000003 + self._train_calls, bn_state=None, bn_output=None)
        loop.update_moving_statistics(getattr(self.net_state.layers[-1], 'momentum', 0.99), getattr(self.net_output.layers[-1], 'momentum', 0.99))
        if self.net_state.batch_normalization: self.net_state.mark_device_newer()
        if self.net_output.batch_normalization: self.net_output.mark_device_newer()
        return k

    # ---- training -------------------------------------------------------------------------------------------------------
    _graph_based = False

    def training_step(self, g: GraphTensor, mean: bool, *, state0=None, masks_state=None, masks_output=None) -> dict:
        """One batch: device gradients (gnn_loop_train_step), net_state gradients divided by the iteration count when
        ``mean`` (reference GNN_BaseClass.py:241), optimizer update, BatchNormalization moving statistics.  Returns the raw
        result of the device step (loss, k, gradients) for inspection."""
        from GNN import losses
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        if self.optimizer is None or not hasattr(self.optimizer, 'apply_gradients'):
            raise TypeError('train() needs an optimizer with apply_gradients, e.g. GNN.optimizers.Adam()')
        kind = losses.device_loss_kind(self.loss_function, self.loss_args)
        if self._graph_based and not g.loop_mask().all():
            raise ValueError('graph-based GNN needs set_mask and output_mask all True')
        loop = self._device_loop(g.device_graph(self.device))
        self._prepare_loop(g, loop)
        if self.state_vect_dim > 0:
            self.seed += 1
            loop.set_state0(state0, self.seed)
        targets = self.get_filtered_tensor(g, g.targets)
        weights = self.get_filtered_tensor(g, g.sample_weights)
        self._train_calls = getattr(self, '_train_calls', 0) + 1
        from GNN import regularizers
        dev_s, dev_o = self.net_state.device_mlp(self.device), self.net_output.device_mlp(self.device)
        # Device-side optimizer step (weights, slots and gradients stay in HBM) when nothing of the step lives on the host:
        # an optimizer that knows the engine's update rules and no kernel / bias regularizers (their gradients are host-side).
        on_device = (getattr(self, 'device_optimizer', True) and not regularizers.any_regularizer(self.get_dense_layers())
                     and hasattr(self.optimizer, 'device_step_args'))
        step_args = self.optimizer.device_step_args() if on_device else None
        if step_args is not None:
            self.net_state.bind_optimizer(self.optimizer)
            self.net_output.bind_optimizer(self.optimizer)
            bn_s, bn_o = self.net_state.layers[-1], self.net_output.layers[-1]
            loop.arm_optimizer(step_args[0], step_args[1], mean, getattr(bn_s, 'momentum', 0.99), getattr(bn_o, 'momentum', 0.99))
        res = loop.train_step(dev_s, dev_o, None, targets, weights,
                              kind, g.nodegraph_csr() if self._graph_based else None, dropout_state=self.net_state.dropout_rates(),
                              dropout_output=self.net_output.dropout_rates(), masks_state=masks_state, masks_output=masks_output,
                              seed=self.seed * 1000003 + self._train_calls,
                              bn_state=None if step_args is not None else self.net_state.bn_gamma_beta(),
                              bn_output=None if step_args is not None else self.net_output.bn_gamma_beta())
        if step_args is not None:
            self.optimizer.device_step_done()          # counted only now: a failed step (exception above) leaves t where it was
            self.net_state.mark_device_newer()
            self.net_output.mark_device_newer()
            return res
        k = res['k']
        # regularizer terms are part of the taped loss (reference GNN_BaseClass.py:223-235): their gradients join the device ones
        for net, key in ((self.net_state, 'grads_state'), (self.net_output, 'grads_output')):
            pen, rg = regularizers.penalty_and_gradients(net.dense_layers)
            res['loss'] += pen
            for li, (gk, gb) in enumerate(rg):
                if gk is not None: res[key][2 * li] = res[key][2 * li] + gk
                if gb is not None: res[key][2 * li + 1] = res[key][2 * li + 1] + gb
        gs = [a / k for a in res['grads_state']] if (mean and k) else res['grads_state']
        ws, wo = self.net_state.trainable_variables, self.net_output.trainable_variables
        new = self.optimizer.apply_gradients(zip(gs + res['grads_output'], ws + wo))
        self.net_state.set_trainable(new[:len(ws)])
        self.net_output.set_trainable(new[len(ws):])
        self.net_state.update_moving_statistics(res['bn_batch_state'])
        if loop.n_masked: self.net_output.update_moving_statistics(res['bn_batch_output'])
        return res

    def _prepare_loop(self, g: GraphTensor, loop) -> None:
        """Hook for subclasses that need more than the node mask on the device loop."""

    def Loop(self, g: Union[GraphObject, GraphTensor], *, training: bool = False, state0=None, _want_state: bool = True):
        """(k, state [N, Ds], out [M, T]) with k a float as in the reference (GNN.py:267, :280).  _want_state=False (internal:
        evaluate / test) leaves the state on the device and returns None in its place."""
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        k, loop = self._run(g.device_graph(self.device), training, state0)
        return k, (loop.state() if _want_state else None), loop.output()


class GNNedgeBased(GNNnodeBased):
    """GNN for edge-based problems: node-based state loop, then net_output on [state(i0) | state(i1) | arc label] of the
    arcs selected by set_mask & output_mask (reference GNN.py:286-302; the pairing of index pairs and arc labels by position
    is the reference's)."""

    def _prerun(self, graphs) -> list:
        return []                                           # (the per-arc readout is configured inside Loop: one graph after the other)

    def Loop(self, g: Union[GraphObject, GraphTensor], *, training: bool = False, state0=None, _want_state: bool = True):
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        dev = g.device_graph(self.device)
        loop = self._device_loop(dev)
        self._prepare_loop(g, loop)
        if self.state_vect_dim > 0:
            loop.set_state0(state0, self.seed)
        k = self._train_forward(loop) if training else loop.run(False)
        return k, (loop.state() if _want_state else None), loop.output()

    def _prepare_loop(self, g: GraphTensor, loop, own_labels: bool = False) -> None:
        """own_labels: the loop runs on an LGNN-derived graph, which carries its own (widened) arc labels on the device."""
        if not getattr(loop, '_edge_ready', False):
            entry_dst, arc_labels, mask = g.edge_readout_arrays()
            loop.set_edge_readout(entry_dst, None if own_labels else arc_labels, mask)
            loop._edge_ready = True


class GNNgraphBased(GNNnodeBased):
    """GNN for graph-based problems: node-based Loop followed by the NodeGraph readout (reference GNN.py:318-333)."""

    _graph_based = True

    @staticmethod
    def get_filtered_tensor(g: GraphTensor, inp):
        return np.asarray(inp, dtype=np.float32)      # targets are per graph: never filtered (reference GNN.py:313-315)

    def Loop(self, g: Union[GraphObject, GraphTensor], *, training: bool = False, state0=None, _want_state: bool = True):
        if g.NodeGraph is None: raise ValueError('WRONG GNN. NodeGraph is None: GNN is graph-based, while problem is non graph-based.')
        if isinstance(g, GraphObject): g = GraphTensor.fromGraphObject(g)
        if not g.loop_mask().all():
            # the reference multiplies NodeGraph [N, G] with the masked node outputs [M, T]: a shape error unless M == N
            raise ValueError('graph-based GNN needs set_mask and output_mask all True (NodeGraph rows must match node outputs)')
        k, loop = self._run(g.device_graph(self.device), training, state0)
        return k, (loop.state() if _want_state else None), loop.readout(*g.nodegraph_csr())
