"""ctypes front-end of the plain-C oracle (oracle/gnn_oracle.c).  TEST INFRASTRUCTURE ONLY (see gnn_oracle.py)."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, 'libgnn_oracle.so')
_lib = None

ACT_CODES = {'linear': 0, None: 0, 'relu': 1, 'selu': 2, 'elu': 3, 'tanh': 4, 'sigmoid': 5, 'softmax': 6}
BN_EPS = 1e-3


def default_threads() -> int:
    """Threads the C oracle may use: the CPU share of this process (affinity, cgroup quota), capped at 64."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    env = os.environ.get('GNN_ORACLE_THREADS')
    return int(env) if env else max(1, min(n, 64))


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ('gnn_oracle.c', 'gnn_oracle_f64.c')]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(['make', '-C', _HERE, '-B', 'libgnn_oracle.so'], stdout=subprocess.DEVNULL)
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        try:
            _lib = C.CDLL(_SO)
        except OSError:           # stale binary from another machine: rebuild once
            build(force=True)
            _lib = C.CDLL(_SO)
        _lib.orc_expf.restype = C.c_float
        _lib.orc_expf.argtypes = [C.c_float]
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_loop.restype = C.c_int
        _lib.orc_loop_f64.restype = C.c_int
        _lib.orc_set_threads(C.c_int(default_threads()))
    return _lib


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32))


class _Mlp:
    """Flattens a Keras-style weight list [W1,b1,...,(gamma,beta,mean,var)] for the C entry points."""

    def __init__(self, weights, activations, batch_normalization):
        n = len(activations)
        self.n = n
        self.W = [np.ascontiguousarray(weights[2 * l], dtype=np.float32) for l in range(n)]
        self.b = [np.ascontiguousarray(weights[2 * l + 1], dtype=np.float32) for l in range(n)]
        self.dims = np.array([self.W[0].shape[0]] + [w.shape[1] for w in self.W], dtype=np.int32)
        self.acts = np.array([ACT_CODES[a] for a in activations], dtype=np.int32)
        self.Wp = (C.POINTER(C.c_float) * n)(*[_fp(w) for w in self.W])
        self.bp = (C.POINTER(C.c_float) * n)(*[_fp(b) for b in self.b])
        self.bn = None
        if batch_normalization:
            self.bn = np.ascontiguousarray(np.concatenate([np.asarray(a, dtype=np.float32).ravel()
                                                           for a in weights[2 * n:2 * n + 4]]))


def expf(x: np.ndarray) -> np.ndarray:
    l = lib()
    x = np.asarray(x, dtype=np.float32)
    return np.array([l.orc_expf(C.c_float(float(v))) for v in x.ravel()], dtype=np.float32).reshape(x.shape)


def mlp_forward(x, weights, activations, batch_normalization):
    l = lib()
    m = _Mlp(weights, activations, batch_normalization)
    x = np.ascontiguousarray(x, dtype=np.float32)
    y = np.empty((x.shape[0], int(m.dims[-1])), dtype=np.float32)
    l.orc_mlp(C.c_int64(x.shape[0]), C.c_int(m.n), _ip(m.dims), _ip(m.acts), m.Wp, m.bp, _fp(m.bn),
              C.c_float(BN_EPS), _fp(x), C.c_int64(x.shape[1]), _fp(y), C.c_int64(y.shape[1]))
    return y


def spmm(csr, dense):
    l = lib()
    indptr, inner, val = csr
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    inner = np.ascontiguousarray(inner, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float32)
    dense = np.ascontiguousarray(dense, dtype=np.float32)
    out = np.empty((len(indptr) - 1, dense.shape[1]), dtype=np.float32)
    l.orc_spmm(C.c_int64(len(indptr) - 1), _ip(indptr), _ip(inner), _fp(val), _fp(dense), C.c_int(dense.shape[1]),
               _fp(out), C.c_int64(dense.shape[1]))
    return out


def readout(nodegraph, out_nodes):
    l = lib()
    ng = np.ascontiguousarray(nodegraph, dtype=np.float32)
    on = np.ascontiguousarray(out_nodes, dtype=np.float32)
    res = np.empty((ng.shape[1], on.shape[1]), dtype=np.float32)
    l.orc_readout(C.c_int64(ng.shape[0]), C.c_int(ng.shape[1]), C.c_int(on.shape[1]), _fp(ng), _fp(on), _fp(res))
    return res


def loop_node(g: dict, net_state: dict, net_output: dict, state_vect_dim: int, max_iteration: int, threshold: float,
              state0=None, n_threads: int = 0, want_out: bool = True):
    """Same contract as gnn_oracle.loop_node, evaluated by the C restatement (fixed fp evaluation order)."""
    n_threads = n_threads or default_threads()
    l = lib()
    nodes = np.ascontiguousarray(g['nodes'], dtype=np.float32)
    arcl = np.ascontiguousarray(np.asarray(g['arcs'], dtype=np.float32)[:, 2:])
    n, nl, al = nodes.shape[0], nodes.shape[1], arcl.shape[1]
    indptr = np.ascontiguousarray(g['adjT'][0], dtype=np.int32)
    assert np.array_equal(indptr, g['arcT'][0])
    adj_src = np.ascontiguousarray(g['adjT'][1], dtype=np.int32)
    adj_w = np.ascontiguousarray(g['adjT'][2], dtype=np.float32)
    arc_id = np.ascontiguousarray(g['arcT'][1], dtype=np.int32)
    arc_w = np.ascontiguousarray(g['arcT'][2], dtype=np.float32)
    mask = np.ascontiguousarray(np.logical_and(g['set_mask'], g['output_mask']), dtype=np.uint8)
    st = _Mlp(net_state['weights'], net_state['activations'], net_state['batch_normalization'])
    ou = _Mlp(net_output['weights'], net_output['activations'], net_output['batch_normalization'])
    ds = state_vect_dim if state_vect_dim else nl
    s0 = None
    if state_vect_dim:
        s0 = np.ascontiguousarray(state0, dtype=np.float32)
        assert s0.shape == (n, ds)
    k = C.c_float(0)
    m = C.c_int64(0)
    state = np.empty((n, ds), dtype=np.float32)
    out = np.empty((int(mask.sum()), int(ou.dims[-1])), dtype=np.float32) if want_out else None
    rc = l.orc_loop(C.c_int64(n), _ip(indptr), _ip(adj_src), _fp(adj_w), _ip(arc_id), _fp(arc_w), _fp(nodes),
                    C.c_int(nl), _fp(arcl), C.c_int(al), mask.ctypes.data_as(C.POINTER(C.c_uint8)),
                    C.c_int(state_vect_dim), C.c_int(st.n), _ip(st.dims), _ip(st.acts), st.Wp, st.bp, _fp(st.bn),
                    C.c_int(ou.n), _ip(ou.dims), _ip(ou.acts), ou.Wp, ou.bp, _fp(ou.bn), C.c_float(BN_EPS),
                    C.c_int(max_iteration), C.c_float(threshold), _fp(s0), C.byref(k), _fp(state), _fp(out),
                    C.byref(m), C.c_int(n_threads))
    if rc != 0:
        raise RuntimeError(f'orc_loop failed: {rc}')
    return float(k.value), state, out


def loop_node_f64(g: dict, net_state: dict, net_output: dict, state_vect_dim: int, max_iteration: int, threshold: float,
                  state0=None, n_threads: int = 0, want_out: bool = True):
    """The float64 shadow (oracle/gnn_oracle_f64.c): same contract as loop_node, state / out returned as float64.  The arbiter for
    rounding questions at sizes where gnn_oracle.loop_node(dtype=np.float64) is too slow."""
    n_threads = n_threads or default_threads()
    l = lib()
    nodes = np.ascontiguousarray(g['nodes'], dtype=np.float32)
    arcl = np.ascontiguousarray(np.asarray(g['arcs'], dtype=np.float32)[:, 2:])
    n, nl, al = nodes.shape[0], nodes.shape[1], arcl.shape[1]
    indptr = np.ascontiguousarray(g['adjT'][0], dtype=np.int32)
    adj_src = np.ascontiguousarray(g['adjT'][1], dtype=np.int32)
    adj_w = np.ascontiguousarray(g['adjT'][2], dtype=np.float32)
    arc_id = np.ascontiguousarray(g['arcT'][1], dtype=np.int32)
    arc_w = np.ascontiguousarray(g['arcT'][2], dtype=np.float32)
    mask = np.ascontiguousarray(np.logical_and(g['set_mask'], g['output_mask']), dtype=np.uint8)
    st = _Mlp(net_state['weights'], net_state['activations'], net_state['batch_normalization'])
    ou = _Mlp(net_output['weights'], net_output['activations'], net_output['batch_normalization'])
    ds = state_vect_dim if state_vect_dim else nl
    s0 = np.ascontiguousarray(state0, dtype=np.float32) if state_vect_dim else None
    k, m = C.c_float(0), C.c_int64(0)
    state = np.empty((n, ds), dtype=np.float64)
    out = np.empty((int(mask.sum()), int(ou.dims[-1])), dtype=np.float64) if want_out else None
    dp = lambda a: a.ctypes.data_as(C.POINTER(C.c_double)) if a is not None else None
    rc = l.orc_loop_f64(C.c_int64(n), _ip(indptr), _ip(adj_src), _fp(adj_w), _ip(arc_id), _fp(arc_w), _fp(nodes),
                        C.c_int(nl), _fp(arcl), C.c_int(al), mask.ctypes.data_as(C.POINTER(C.c_uint8)),
                        C.c_int(state_vect_dim), C.c_int(st.n), _ip(st.dims), _ip(st.acts), st.Wp, st.bp, _fp(st.bn),
                        C.c_int(ou.n), _ip(ou.dims), _ip(ou.acts), ou.Wp, ou.bp, _fp(ou.bn), C.c_double(BN_EPS),
                        C.c_int(max_iteration), C.c_float(threshold), _fp(s0), C.byref(k), dp(state), dp(out),
                        C.byref(m), C.c_int(n_threads))
    if rc != 0:
        raise RuntimeError(f'orc_loop_f64 failed: {rc}')
    return float(k.value), state, out


def num_threads() -> int:
    return int(lib().orc_num_threads())


def set_spmm_single_thread(on: bool) -> None:
    """Timing knob of the CPU baseline: sparse products on one thread (see gnn_oracle.c)."""
    lib().orc_set_spmm_single_thread(C.c_int(bool(on)))
