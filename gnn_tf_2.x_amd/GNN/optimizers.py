"""Host-side optimizers standing in for tf.keras.optimizers (reference starter.py:81).  The trainable arrays of the two MLPs
are a few hundred KB, so the update runs in NumPy; gradients come from the device (gnn_loop_train_step)."""
import numpy as np


class Optimizer:
    """Slots (Adam moments, SGD velocity) live either on the device (with the gnn_mlp: gnn_loop_optimizer_step) or on the host
    (apply_gradients), the step counter here.  A model that changes path (a regularizer appears, device_optimizer is toggled) restarts
    BOTH together - slots and counter - like a new Keras optimizer would; `_slot_token` changes then, which makes Sequential.bind_optimizer
    zero the device slots."""
    _path = None
    _slot_token = None

    def get_config(self):
        return dict(self._config)

    def reset(self):
        self._slot_token = object()

    def _enter(self, path):
        if self._path is not None and self._path != path:
            self.reset()
        self._path = path

    def device_step_args(self):
        """(kind, hyper[<= 4]) of include/gnn_hip.h:gnn_loop_arm_optimizer for the NEXT step; None: host only.  The step is counted by
        device_step_done() once it has succeeded (a failed gnn_loop_train_step must not advance the bias correction)."""
        return None

    def device_step_done(self):
        pass

    def apply_gradients(self, grads_and_vars):
        """[(grad, array)] -> list of updated arrays, in order (arrays are identified by position across calls)."""
        raise NotImplementedError


class Adam(Optimizer):
    """Keras Adam: lr_t = lr sqrt(1 - b2^t) / (1 - b1^t); p <- p - lr_t m / (sqrt(v) + epsilon)."""

    def __init__(self, learning_rate=0.001, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        self._config = dict(learning_rate=learning_rate, beta_1=beta_1, beta_2=beta_2, epsilon=epsilon)
        self.learning_rate, self.beta_1, self.beta_2, self.epsilon = learning_rate, beta_1, beta_2, epsilon
        self.iterations, self._m, self._v = 0, None, None

    def reset(self):
        super().reset()
        self.iterations, self._m, self._v = 0, None, None

    def device_step_args(self):
        self._enter('device')
        t = self.iterations + 1
        lr_t = self.learning_rate * np.sqrt(1 - self.beta_2 ** t) / (1 - self.beta_1 ** t)
        return 1, [lr_t, self.beta_1, self.beta_2, self.epsilon]

    def device_step_done(self):
        self.iterations += 1

    def apply_gradients(self, grads_and_vars):
        self._enter('host')
        grads_and_vars = list(grads_and_vars)
        if self._m is None:
            self._m = [np.zeros_like(p, dtype=np.float64) for _, p in grads_and_vars]
            self._v = [np.zeros_like(p, dtype=np.float64) for _, p in grads_and_vars]
        self.iterations += 1
        t = self.iterations
        lr_t = self.learning_rate * np.sqrt(1 - self.beta_2 ** t) / (1 - self.beta_1 ** t)
        out = []
        for i, (g, p) in enumerate(grads_and_vars):
            g = np.asarray(g, np.float64)
            self._m[i] = self.beta_1 * self._m[i] + (1 - self.beta_1) * g
            self._v[i] = self.beta_2 * self._v[i] + (1 - self.beta_2) * g * g
            out.append((np.asarray(p, np.float64) - lr_t * self._m[i] / (np.sqrt(self._v[i]) + self.epsilon)).astype(np.float32))
        return out


class SGD(Optimizer):
    def __init__(self, learning_rate=0.01, momentum=0.0):
        self._config = dict(learning_rate=learning_rate, momentum=momentum)
        self.learning_rate, self.momentum, self._vel = learning_rate, momentum, None

    def reset(self):
        super().reset()
        self._vel = None

    def device_step_args(self):
        self._enter('device')
        return 0, [self.learning_rate, self.momentum]

    def apply_gradients(self, grads_and_vars):
        self._enter('host')
        grads_and_vars = list(grads_and_vars)
        if self._vel is None:
            self._vel = [np.zeros_like(p, dtype=np.float64) for _, p in grads_and_vars]
        out = []
        for i, (g, p) in enumerate(grads_and_vars):
            self._vel[i] = self.momentum * self._vel[i] - self.learning_rate * np.asarray(g, np.float64)
            out.append((np.asarray(p, np.float64) + self._vel[i]).astype(np.float32))
        return out


def serialize(opt) -> dict:
    """{'class_name', 'config'} (stands in for tf.keras.optimizers.serialize, reference GNN.py:105)."""
    return {'class_name': type(opt).__name__, 'config': opt.get_config()} if isinstance(opt, Optimizer) else None


def deserialize(d):
    if d is None:
        return None
    return {'Adam': Adam, 'SGD': SGD}[d['class_name']](**d['config'])
