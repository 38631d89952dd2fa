"""MLP factory and dimension helper with the reference's signatures (``GNN/MLP.py``), without Keras.

``MLP(...)`` returns a small ``Sequential`` stand-in that keeps the Keras layer order of reference MLP.py:46-64
(``[Dropout?] Dense ... [BatchNormalization]``), the Keras ``get_weights()`` / ``set_weights()`` layout
``[W1, b1, ..., Wn, bn, gamma, beta, moving_mean, moving_variance]`` (reference GNN.py:163-172) and evaluates on the MI355X
through the C ABI (``gnn_mlp_forward``).  Weight initialisers follow the Keras definitions (VarianceScaling with a
truncated normal); their random streams are NumPy's, not TensorFlow's.
"""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

_TRUNC_STD = 0.87962566103423978   # std of a unit normal truncated to [-2, 2]


def _fans(shape):
    if len(shape) == 1:
        return shape[0], shape[0]
    return shape[0], shape[1]


def _variance_scaling(shape, scale, mode, distribution, rng):
    fan_in, fan_out = _fans(shape)
    n = {'fan_in': fan_in, 'fan_out': fan_out, 'fan_avg': (fan_in + fan_out) / 2}[mode]
    if distribution == 'uniform':
        lim = np.sqrt(3 * scale / n)
        return rng.uniform(-lim, lim, shape)
    std = np.sqrt(scale / n) / _TRUNC_STD
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return x * std


_INITIALIZERS = {
    'zeros': lambda s, r: np.zeros(s), 'ones': lambda s, r: np.ones(s),
    'lecun_normal': lambda s, r: _variance_scaling(s, 1.0, 'fan_in', 'normal', r),
    'lecun_uniform': lambda s, r: _variance_scaling(s, 1.0, 'fan_in', 'uniform', r),
    'glorot_normal': lambda s, r: _variance_scaling(s, 1.0, 'fan_avg', 'normal', r),
    'glorot_uniform': lambda s, r: _variance_scaling(s, 1.0, 'fan_avg', 'uniform', r),
    'he_normal': lambda s, r: _variance_scaling(s, 2.0, 'fan_in', 'normal', r),
    'he_uniform': lambda s, r: _variance_scaling(s, 2.0, 'fan_in', 'uniform', r),
    'random_normal': lambda s, r: 0.05 * r.standard_normal(s),
}

_rng = np.random.default_rng()


def set_seed(seed) -> None:
    """Seed the initialisers (the reference relies on TF's global generator)."""
    global _rng
    _rng = np.random.default_rng(seed)


def _initialize(spec, shape):
    if callable(spec):
        return np.asarray(spec(shape), dtype=np.float32)
    if spec not in _INITIALIZERS:
        raise ValueError(f'unknown initializer {spec!r}')
    return _INITIALIZERS[spec](shape, _rng).astype(np.float32)


def _device_backed(name):
    """Array attribute of a layer whose newest value may live on the device (after a device-side optimizer step, reference
    GNN_BaseClass.py:243-247): reading it first pulls the owner Sequential's arrays back if they are stale."""
    priv = '_' + name

    def get(self):
        owner = getattr(self, '_owner', None)
        if owner is not None and owner._host_stale:
            owner._pull_from_device()
        return getattr(self, priv)

    def put(self, value):
        owner = getattr(self, '_owner', None)
        if owner is not None and owner._host_stale:
            owner._pull_from_device()
        setattr(self, priv, value)

    return property(get, put)


class Dense:
    kernel = _device_backed('kernel')
    bias = _device_backed('bias')

    def __init__(self, units, activation=None, kernel_initializer='glorot_uniform', bias_initializer='zeros',
                 kernel_regularizer=None, bias_regularizer=None, input_shape=None):
        self.units, self.activation = int(units), (activation if activation is not None else 'linear')
        self.kernel_initializer, self.bias_initializer = kernel_initializer, bias_initializer
        self.kernel_regularizer, self.bias_regularizer = kernel_regularizer, bias_regularizer
        self.input_shape = input_shape
        self.kernel = self.bias = None

    def build(self, n_in):
        self.kernel = _initialize(self.kernel_initializer, (n_in, self.units))
        self.bias = _initialize(self.bias_initializer, (self.units,))


class Dropout:
    def __init__(self, rate):
        self.rate = rate


class AlphaDropout(Dropout):
    pass


class BatchNormalization:
    gamma = _device_backed('gamma')
    beta = _device_backed('beta')
    moving_mean = _device_backed('moving_mean')
    moving_variance = _device_backed('moving_variance')

    def __init__(self, epsilon=1e-3, momentum=0.99):
        self.epsilon, self.momentum = epsilon, momentum
        self.gamma = self.beta = self.moving_mean = self.moving_variance = None

    def build(self, n):
        self.gamma, self.beta = np.ones(n, np.float32), np.zeros(n, np.float32)
        self.moving_mean, self.moving_variance = np.zeros(n, np.float32), np.ones(n, np.float32)


class Sequential:
    """Ordered layer list with the Keras weight layout; inference runs on the device."""

    def __init__(self, layers):
        self.layers = list(layers)
        dense = [l for l in self.layers if isinstance(l, Dense)]
        if not dense or dense[0].input_shape is None:
            raise ValueError('the first Dense layer needs input_shape')
        width = int(dense[0].input_shape[0])
        self.input_dim = width
        for layer in self.layers:
            if isinstance(layer, Dense):
                layer.build(width)
                width = layer.units
            elif isinstance(layer, BatchNormalization):
                layer.build(width)
        self.output_dim = width
        bns = [l for l in self.layers if isinstance(l, BatchNormalization)]
        if len(bns) > 1 or (bns and self.layers[-1] is not bns[0]):
            raise NotImplementedError('only one trailing BatchNormalization is supported (what GNN.MLP builds)')
        self._device = None
        self._host_stale = False          # True: the device holds newer arrays than the layers' host copies
        for layer in self.layers:
            if isinstance(layer, (Dense, BatchNormalization)):
                layer._owner = self

    def bind_optimizer(self, optimizer):
        """The device-side slots (Adam moments, SGD velocity) live with the device MLP; a different optimizer object starts from
        zero slots, as a new tf.keras optimizer would."""
        key = (optimizer, getattr(optimizer, '_slot_token', None))      # the token changes when the optimizer restarts (path switch)
        old = getattr(self, '_slots_of', None)
        if old is None or old[0] is not key[0] or old[1] is not key[1]:
            if old is not None and self._device is not None:
                self._device.reset_optimizer()
            self._slots_of = key

    def mark_device_newer(self):
        """A device-side optimizer step has updated the gnn_mlp's arrays; the host copies are refreshed on their next read."""
        self._host_stale = True

    def _pull_from_device(self):
        self._host_stale = False
        w = self._device.get_weights()
        dense = self.dense_layers
        for i, l in enumerate(dense):
            l._kernel, l._bias = w[2 * i], w[2 * i + 1]
        if self.batch_normalization:
            bn = self.layers[-1]
            bn._gamma, bn._beta, bn._moving_mean, bn._moving_variance = w[2 * len(dense):]

    # Keras-compatible views
    @property
    def dense_layers(self):
        return [l for l in self.layers if isinstance(l, Dense)]

    @property
    def batch_normalization(self):
        return isinstance(self.layers[-1], BatchNormalization)

    @property
    def activations(self):
        return [l.activation for l in self.dense_layers]

    def get_weights(self):
        out = []
        for l in self.dense_layers:
            out += [l.kernel.copy(), l.bias.copy()]
        if self.batch_normalization:
            bn = self.layers[-1]
            out += [bn.gamma.copy(), bn.beta.copy(), bn.moving_mean.copy(), bn.moving_variance.copy()]
        return out

    def set_weights(self, weights):
        dense = self.dense_layers
        expect = 2 * len(dense) + (4 if self.batch_normalization else 0)
        if len(weights) != expect:
            raise ValueError(f'expected {expect} arrays, got {len(weights)}')
        for i, l in enumerate(dense):
            k, b = np.asarray(weights[2 * i], np.float32), np.asarray(weights[2 * i + 1], np.float32)
            if k.shape != l.kernel.shape or b.shape != l.bias.shape:
                raise ValueError(f'layer {i}: shape mismatch')
            l.kernel, l.bias = k.copy(), b.copy()
        if self.batch_normalization:
            bn = self.layers[-1]
            bn.gamma, bn.beta, bn.moving_mean, bn.moving_variance = (np.asarray(a, np.float32).copy() for a in weights[2 * len(dense):])
        if self._device is not None:
            self._device.set_weights(self.get_weights())

    @property
    def trainable_variables(self):
        out = []
        for l in self.dense_layers:
            out += [l.kernel, l.bias]
        if self.batch_normalization:
            out += [self.layers[-1].gamma, self.layers[-1].beta]
        return out

    # ---- training-side views (used by BaseClass.train through gnn_loop_train_step) ------------------------------------
    def dropout_rates(self):
        """[n_dense + 1] Dropout rate in front of Dense l (last entry: in front of BatchNormalization); 0 = none."""
        rates, idx = [0.0] * (len(self.dense_layers) + 1), 0
        for layer in self.layers:
            if isinstance(layer, Dense):
                idx += 1
            elif isinstance(layer, AlphaDropout):
                rates[idx] = -float(layer.rate)          # the engine's code for AlphaDropout (include/gnn_hip.h)
            elif isinstance(layer, Dropout):
                rates[idx] = float(layer.rate)
        return rates

    def bn_gamma_beta(self):
        if not self.batch_normalization:
            return None
        return np.concatenate([self.layers[-1].gamma, self.layers[-1].beta]).astype(np.float32)

    def set_trainable(self, arrays):
        """Write back [W1, b1, ..., gamma, beta] (the order of ``trainable_variables``)."""
        w = self.get_weights()
        n = 2 * len(self.dense_layers)
        w[:n] = [np.asarray(a, np.float32) for a in arrays[:n]]
        if self.batch_normalization:
            w[n:n + 2] = [np.asarray(a, np.float32) for a in arrays[n:n + 2]]
        self.set_weights(w)

    def update_moving_statistics(self, batch_stats):
        """Keras: moving <- moving * momentum + batch * (1 - momentum), once per BatchNormalization call (``batch_stats`` has
        one [2, F] row per call, in call order)."""
        if not self.batch_normalization:
            return
        bn = self.layers[-1]
        mean, var = bn.moving_mean.astype(np.float64), bn.moving_variance.astype(np.float64)
        for mu, v in np.asarray(batch_stats, np.float64).reshape(-1, 2, mean.shape[0]):
            mean = mean * bn.momentum + mu * (1 - bn.momentum)
            var = var * bn.momentum + v * (1 - bn.momentum)
        w = self.get_weights()
        w[-2], w[-1] = mean.astype(np.float32), var.astype(np.float32)
        self.set_weights(w)

    def device_mlp(self, device: int = 0):
        from GNN import _engine
        if self._device is None:
            eps = self.layers[-1].epsilon if self.batch_normalization else 1e-3
            self._device = _engine.Mlp(self.get_weights(), self.activations, self.batch_normalization, eps, device)
        return self._device

    def __call__(self, x, training=False):
        if training:
            raise NotImplementedError('training-mode forward (Dropout masks, BatchNormalization batch statistics) is not implemented on the device yet')
        return self.device_mlp().forward(np.asarray(x, dtype=np.float32))


def sequential_config(model: Sequential) -> list:
    """JSON-able architecture of a Sequential (stands in for the SavedModel of reference GNN.py:100-101); initialisers that are
    callables are stored as 'zeros' (the weights are saved beside the architecture anyway)."""
    name = lambda spec: spec if isinstance(spec, str) else 'zeros'
    out = []
    for l in model.layers:
        if isinstance(l, Dense):
            out.append({'type': 'Dense', 'units': l.units, 'activation': l.activation, 'kernel_initializer': name(l.kernel_initializer),
                        'bias_initializer': name(l.bias_initializer), 'input_shape': list(l.input_shape) if l.input_shape is not None else None})
        elif isinstance(l, BatchNormalization):
            out.append({'type': 'BatchNormalization', 'epsilon': l.epsilon, 'momentum': l.momentum})
        else:
            out.append({'type': type(l).__name__, 'rate': float(l.rate)})
    return out


def sequential_from_config(config: list, weights=None) -> Sequential:
    layers = []
    for c in config:
        if c['type'] == 'Dense':
            layers.append(Dense(c['units'], c['activation'], c['kernel_initializer'], c['bias_initializer'],
                                input_shape=tuple(c['input_shape']) if c['input_shape'] is not None else None))
        elif c['type'] == 'BatchNormalization':
            layers.append(BatchNormalization(c['epsilon'], c['momentum']))
        elif c['type'] in ('Dropout', 'AlphaDropout'):
            layers.append({'Dropout': Dropout, 'AlphaDropout': AlphaDropout}[c['type']](c['rate']))
        else:
            raise ValueError(f"unknown layer type {c['type']!r}")
    model = Sequential(layers)
    if weights is not None:
        model.set_weights(weights)
    return model


def clone_model(model: Sequential, copy_weights: bool = False) -> Sequential:
    """Fresh Sequential with the same architecture (stands in for tf.keras.models.clone_model, reference GNN.py:80-81)."""
    layers = []
    for l in model.layers:
        if isinstance(l, Dense):
            layers.append(Dense(l.units, l.activation, l.kernel_initializer, l.bias_initializer, l.kernel_regularizer,
                                l.bias_regularizer, l.input_shape))
        elif isinstance(l, BatchNormalization):
            layers.append(BatchNormalization(l.epsilon, l.momentum))
        else:
            layers.append(type(l)(l.rate))
    new = Sequential(layers)
    if copy_weights:
        new.set_weights(model.get_weights())
    return new


# ---------------------------------------------------------------------------------------------------------------------
def MLP(input_dim: int, layers: list[int], activations, kernel_initializer, bias_initializer,
        kernel_regularizer=None, bias_regularizer=None, dropout_rate: Union[list[float], float, None] = None,
        dropout_pos: Optional[Union[list[int], int]] = None, alphadropout: bool = False, batch_normalization: bool = True):
    """Same arguments as reference MLP.py:11-13.  Dropout layers land at ``dropout_pos`` shifted by the number of dropout
    layers already inserted (MLP.py:54-55; position 0 is in front of the first Dense); a BatchNormalization closes the
    stack when ``batch_normalization`` (default True, MLP.py:13,63)."""
    if dropout_rate is None or dropout_pos is None:
        dropout_rate, dropout_pos = [], []
    broadcast = lambda v: v if type(v) == list else [v for _ in layers]
    activations, kernel_initializer, bias_initializer = broadcast(activations), broadcast(kernel_initializer), broadcast(bias_initializer)
    kernel_regularizer, bias_regularizer = broadcast(kernel_regularizer), broadcast(bias_regularizer)
    if type(dropout_pos) == int: dropout_pos = [dropout_pos]
    if type(dropout_rate) == float: dropout_rate = [dropout_rate for _ in dropout_pos]
    if len({len(x) for x in (activations, kernel_initializer, bias_initializer, kernel_regularizer, bias_regularizer, layers)}) > 1:
        raise ValueError('Dense parameters must have the same length to be correctly processed')
    if len(dropout_rate) != len(dropout_pos):
        raise ValueError('Dropout parameters must have the same length to be correctly processed')

    stack = [Dense(u, a, ki, bi, kr, br) for u, a, ki, bi, kr, br in
             zip(layers, activations, kernel_initializer, bias_initializer, kernel_regularizer, bias_regularizer)]
    drop = AlphaDropout if alphadropout else Dropout
    for n_inserted, (pos, rate) in enumerate(zip(dropout_pos, dropout_rate)):
        stack.insert(pos + n_inserted, drop(rate))
    # the reference sets input_shape on params[0], which may be a Dropout (MLP.py:58): the width is the model input either way
    next(l for l in stack if isinstance(l, Dense)).input_shape = (input_dim,)
    if batch_normalization:
        stack.append(BatchNormalization())
    return Sequential(stack)


# ---------------------------------------------------------------------------------------------------------------------
def get_inout_dims(net_name: str, dim_node_label: int, dim_arc_label: int, dim_target: int, problem_based: str, dim_state: int,
                   hidden_units: Union[None, int, list[int]],
                   *, layer: int = 0, get_state: bool = False, get_output: bool = False) -> tuple[int, list[int]]:
    """Input width and layer widths of net_state / net_output (reference MLP.py:68-122, including the LGNN layer > 0
    label-widening rules :93-100)."""
    assert layer >= 0
    assert problem_based in ['a', 'n', 'g']
    assert dim_state >= 0
    nl, al, t, ds = dim_node_label, dim_arc_label, dim_target, dim_state
    on_arcs = problem_based == 'a'
    if layer > 0:
        out_on_nodes = t * (not on_arcs) * get_output
        if ds != 0:
            nl += ds * get_state + out_on_nodes
        else:
            nl += layer * nl * get_state + ((layer - 1) * get_state + 1) * out_on_nodes
        al += t * on_arcs * get_output
    if net_name == 'state':
        input_shape, output_shape = al + 2 * (nl + ds), (ds if ds else nl)
    elif net_name == 'output':
        input_shape, output_shape = on_arcs * (nl + al + ds) + nl + dim_state, t
    else:
        raise ValueError(':param net_name: not in [\'state\', \'output\']')
    if hidden_units is None or (type(hidden_units) == int and hidden_units <= 0):
        hidden_units = []
    widths = hidden_units + [output_shape] if type(hidden_units) == list else [hidden_units, output_shape]
    return input_shape, widths
