"""NumPy stand-ins for the tf.keras.losses callables the starter passes as ``loss_function`` (reference starter.py:82).
They act on host arrays AFTER the device loop; they are not part of the hot path."""
import numpy as np

_EPS = 1e-7   # Keras backend epsilon


def categorical_crossentropy(y_true, y_pred, from_logits: bool = False, **_):
    y_true, y_pred = np.asarray(y_true, np.float32), np.asarray(y_pred, np.float32)
    if from_logits:
        z = y_pred - y_pred.max(axis=-1, keepdims=True)
        logp = z - np.log(np.exp(z).sum(axis=-1, keepdims=True))
    else:
        p = y_pred / y_pred.sum(axis=-1, keepdims=True)
        logp = np.log(np.clip(p, _EPS, 1 - _EPS))
    return -(y_true * logp).sum(axis=-1)


def binary_crossentropy(y_true, y_pred, **_):
    p = np.clip(np.asarray(y_pred, np.float32), _EPS, 1 - _EPS)
    y = np.asarray(y_true, np.float32)
    return -(y * np.log(p) + (1 - y) * np.log(1 - p)).mean(axis=-1)


def mean_squared_error(y_true, y_pred, **_):
    return np.square(np.asarray(y_pred, np.float32) - np.asarray(y_true, np.float32)).mean(axis=-1)


mse = mean_squared_error


def device_loss_kind(fn, loss_args) -> int:
    """Loss code of gnn_loop_train_step for the callables above (0 categorical_crossentropy, 1 mean_squared_error,
    2 categorical_crossentropy(from_logits=True))."""
    if fn is categorical_crossentropy:
        return 2 if loss_args.get('from_logits', False) else 0
    if fn is mean_squared_error:
        return 1
    raise NotImplementedError(f'training with loss {getattr(fn, "__name__", fn)!r} is not implemented on the MI355X engine '
                              f'(available: GNN.losses.categorical_crossentropy, GNN.losses.mean_squared_error)')
