"""Shared helpers for the tests: seeded MLP weights and small random graphs (inputs only, no reference code)."""
import numpy as np


def make_mlp(rng, n_in, layers, activation, batch_normalization=True, out_activation=None, bn_random=False, gain=1.0):
    """Keras-layout weight list [W1,b1,...,(gamma,beta,mean,var)], lecun-normal-like scale times ``gain``
    (gain < 1 makes the state map a contraction, so that rounding differences do not grow across iterations)."""
    weights, acts, d = [], [], n_in
    for i, u in enumerate(layers):
        weights.append((gain * rng.standard_normal((d, u)) / np.sqrt(d)).astype(np.float32))
        weights.append((rng.standard_normal(u) / np.sqrt(d)).astype(np.float32))
        acts.append(out_activation if (out_activation is not None and i == len(layers) - 1) else activation)
        d = u
    if batch_normalization:
        if bn_random:
            weights += [rng.uniform(0.5, 1.5, d).astype(np.float32), rng.uniform(-0.2, 0.2, d).astype(np.float32),
                        rng.uniform(-0.2, 0.2, d).astype(np.float32), rng.uniform(0.5, 1.5, d).astype(np.float32)]
        else:
            weights += [np.ones(d, np.float32), np.zeros(d, np.float32), np.zeros(d, np.float32), np.ones(d, np.float32)]
    return dict(weights=weights, activations=acts, batch_normalization=batch_normalization)


def random_arcs(rng, n, n_und, dim_arc_label, symmetric=True, sort=True):
    """Duplicate-free, self-loop-free arc list [src, dst, labels...] in the spirit of GNN_utils.randomGraph."""
    src = rng.integers(0, n - 1, n_und)
    dst = src + np.ceil((n - 1 - src) * rng.random(n_und)).astype(np.int64)
    und = np.unique(np.stack([src, dst], 1), axis=0)
    lab = (2 * rng.random((len(und), dim_arc_label)) - 1)
    if symmetric:
        ids = np.concatenate([und, und[:, ::-1]])
        lab = np.concatenate([lab, lab])
    else:
        ids = und
    arcs = np.concatenate([ids.astype(np.float64), lab], axis=1)
    if sort:
        arcs = arcs[np.lexsort((arcs[:, 1], arcs[:, 0]))]
    else:
        arcs = arcs[rng.permutation(len(arcs))]
    return arcs.astype(np.float32)
