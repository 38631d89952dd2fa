"""Weight regularizers with the call convention of ``tf.keras.regularizers`` (what the reference hands to ``MLP(...,
kernel_regularizer=, bias_regularizer=)``, reference MLP.py:11-13, and sums into the loss in GNN_BaseClass.py:223-228).
Host-side: the penalties and their gradients are a few KB of arithmetic next to the device gradients."""
from __future__ import annotations

import numpy as np


class Regularizer:
    def __call__(self, w) -> float:
        raise NotImplementedError

    def gradient(self, w) -> np.ndarray:
        raise NotImplementedError


class L1L2(Regularizer):
    def __init__(self, l1: float = 0.0, l2: float = 0.0):
        self.l1, self.l2 = float(l1), float(l2)

    def __call__(self, w) -> float:
        w = np.asarray(w, dtype=np.float64)
        return float(self.l1 * np.sum(np.abs(w)) + self.l2 * np.sum(w * w))

    def gradient(self, w) -> np.ndarray:
        w = np.asarray(w, dtype=np.float64)
        return (self.l1 * np.sign(w) + 2.0 * self.l2 * w).astype(np.float32)

    def get_config(self):
        return dict(l1=self.l1, l2=self.l2)


def l1(l: float = 0.01) -> L1L2:
    return L1L2(l1=l)


def l2(l: float = 0.01) -> L1L2:
    return L1L2(l2=l)


def l1_l2(l1: float = 0.01, l2: float = 0.01) -> L1L2:
    return L1L2(l1=l1, l2=l2)


def penalty_and_gradients(dense_layers) -> tuple[float, list]:
    """Sum of the penalties of ``dense_layers`` (reference GNN_BaseClass.py:223-228) and, per layer, the pair of gradients
    (d / d kernel, d / d bias), ``None`` where a layer has no regularizer."""
    total, grads = 0.0, []
    for layer in dense_layers:
        gk = gb = None
        for reg, attr in ((layer.kernel_regularizer, 'kernel'), (layer.bias_regularizer, 'bias')):
            if reg is None:
                continue
            if not (callable(reg) and hasattr(reg, 'gradient')):
                raise TypeError('regularizers must come from GNN.regularizers (a callable with a .gradient(w) method)')
            w = getattr(layer, attr)
            total += reg(w)
            if attr == 'kernel': gk = reg.gradient(w)
            else: gb = reg.gradient(w)
        grads.append((gk, gb))
    return total, grads


def any_regularizer(dense_layers) -> bool:
    return any(l.kernel_regularizer is not None or l.bias_regularizer is not None for l in dense_layers)
