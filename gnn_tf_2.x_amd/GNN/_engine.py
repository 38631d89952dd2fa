"""ctypes binding of include/gnn_hip.h (libgnn_hip.so).  This is the only door from Python into the product.

There is no CPU fallback: if the library is missing or no MI355X is visible the calls raise.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# GNN_HIP_LIBRARY: another build of the same library (the diagnostic build `make DIAG=1` -> libgnn_hip_diag.so); never a fallback
LIB_PATH = os.environ.get('GNN_HIP_LIBRARY') or os.path.join(_HERE, 'libgnn_hip.so')

ACT_CODES = {'linear': 0, None: 0, 'relu': 1, 'selu': 2, 'elu': 3, 'tanh': 4, 'sigmoid': 5, 'softmax': 6}

EXPORTS = ['gnn_last_error', 'gnn_version', 'gnn_device_count', 'gnn_device_synchronize', 'gnn_graph_create', 'gnn_graph_create_from_arcs',
           'gnn_graph_derive', 'gnn_graph_derive_edge', 'gnn_graph_set_arc_order', 'gnn_graph_update_labels', 'gnn_graph_get_nodes', 'gnn_graph_dims', 'gnn_graph_destroy',
           'gnn_mlp_create', 'gnn_mlp_set_weights', 'gnn_mlp_get_weights', 'gnn_mlp_reset_optimizer', 'gnn_mlp_forward', 'gnn_mlp_destroy', 'gnn_loop_create',
           'gnn_loop_set_state0', 'gnn_loop_run', 'gnn_loop_get_state', 'gnn_loop_get_output', 'gnn_loop_readout', 'gnn_loop_set_edge_readout', 'gnn_loop_train_step',
           'gnn_loop_train_forward', 'gnn_loop_train_backward', 'gnn_loop_arm_optimizer', 'gnn_loop_optimizer_step', 'gnn_loop_update_moving_statistics', 'gnn_loss_grad',
           'gnn_counters_get', 'gnn_lgnn_run', 'gnn_loop_run_many', 'gnn_loop_set_impl', 'gnn_loop_gate_info', 'gnn_loop_set_persistent', 'gnn_loop_set_tile_form', 'gnn_loop_drop_cached_aggregates', 'gnn_loop_set_profiling', 'gnn_loop_get_timing', 'gnn_loop_get_exchange_timing', 'gnn_loop_destroy', 'gnn_shard_range',
           'gnn_comm_unique_id', 'gnn_comm_create', 'gnn_comm_allreduce_max', 'gnn_comm_destroy', 'gnn_halo_plan', 'gnn_graph_create_halo',
           'gnn_comm_create_loopback', 'gnn_graph_set_full_adjacency', 'gnn_loop_set_slice_exchange', 'gnn_loop_run_group', 'gnn_loop_readout_group', 'gnn_graph_update_labels_group']

_lib = None


class EngineError(RuntimeError):
    pass


def lib():
    """Load libgnn_hip.so once.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise EngineError(f'{LIB_PATH} is missing: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                              f'or `make -C gnn_tf_2.x_amd/csrc`')
        _lib = C.CDLL(LIB_PATH)
        _lib.gnn_last_error.restype = C.c_char_p
        for name in EXPORTS:
            if name != 'gnn_last_error':
                getattr(_lib, name).restype = C.c_int
    return _lib


def loss_grad(loss_kind: int, targets, out, sample_weights):
    """gnn_loss_grad: (sum_i w_i L(t_i, out_i), d / d out) for loss_kind 0 categorical_crossentropy / 1 mean_squared_error."""
    t, o, w = _f32(targets), _f32(out), _f32(sample_weights)
    if t.shape != o.shape or w.shape != (o.shape[0],): raise ValueError(f'targets {t.shape}, outputs {o.shape}, weights {w.shape} do not match')
    d = np.zeros_like(o)
    loss = C.c_double()
    _check(lib().gnn_loss_grad(C.c_int(loss_kind), C.c_int64(o.shape[0]), C.c_int(o.shape[1]), _fp(t), _fp(o), _fp(w), C.byref(loss), _fp(d)))
    return float(loss.value), d


def _check(rc: int):
    if rc == 0:
        return
    msg = lib().gnn_last_error().decode(errors='replace')
    if rc == -1:
        raise ValueError(msg)
    if rc == -5:
        raise NotImplementedError(msg)
    raise EngineError(f'[gnn_status {rc}] {msg}')


def _f32(a, shape=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if shape is not None and a.shape != tuple(shape):
        raise ValueError(f'expected shape {tuple(shape)}, got {a.shape}')
    return a


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(C.POINTER(C.c_int32)) if a is not None else None


def device_count() -> int:
    n = C.c_int(0)
    _check(lib().gnn_device_count(C.byref(n)))
    return n.value


def require_device(device: int = 0) -> None:
    n = device_count()
    if n <= device:
        raise EngineError(f'no HIP device {device} visible ({n} found): this engine only runs on MI355X (gfx950)')


def shard_range(n_nodes: int, rank: int, world: int) -> tuple[int, int]:
    b, n = C.c_int64(0), C.c_int64(0)
    _check(lib().gnn_shard_range(C.c_int64(n_nodes), C.c_int(rank), C.c_int(world), C.byref(b), C.byref(n)))
    return b.value, n.value


def halo_plan(n_nodes: int, world: int, indptr, adj_src):
    """gnn_halo_plan: (slot [n_nodes] int32, counts [world], block) of the boundary exchange for a whole CSR-by-destination graph."""
    indptr = np.ascontiguousarray(indptr, dtype=np.int32)
    adj_src = np.ascontiguousarray(adj_src, dtype=np.int32)
    slot = np.empty(n_nodes, np.int32)
    counts = np.zeros(world, np.int64)
    block = C.c_int64(0)
    _check(lib().gnn_halo_plan(C.c_int64(n_nodes), C.c_int(world), _ip(indptr), _ip(adj_src), _ip(slot),
                               counts.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(block)))
    return slot, counts, int(block.value)


def shard_halo(n_nodes: int, rank: int, world: int, indptr, adj_src, nodes, plan=None):
    """Arguments of Graph.halo for one rank: sources renumbered into the compact index space
    [own rows | block of boundary rows per rank] and the node labels in the same space."""
    slot, counts, block = plan if plan is not None else halo_plan(n_nodes, world, indptr, adj_src)
    rb, nr = shard_range(n_nodes, rank, world)
    shard = ((n_nodes + world - 1) // world + 31) // 32 * 32
    indptr = np.asarray(indptr)
    e0, e1 = int(indptr[rb]), int(indptr[rb + nr])
    src = np.asarray(adj_src)[e0:e1].astype(np.int64)
    owner = src // shard
    mine = owner == rank
    if np.any(~mine & (slot[src] < 0)):
        raise ValueError('halo plan does not cover a remote source')
    remapped = np.where(mine, src - rb, shard + owner * block + slot[src]).astype(np.int32)
    nodes = np.asarray(nodes, np.float32)
    rep = np.zeros((shard + world * block, nodes.shape[1]), np.float32)
    rep[:nr] = nodes[rb:rb + nr]
    for q in range(world):
        qb, qn = shard_range(n_nodes, q, world)
        ids = qb + np.nonzero(slot[qb:qb + qn] >= 0)[0]
        rep[shard + q * block: shard + q * block + len(ids)] = nodes[ids]
    send = np.nonzero(slot[rb:rb + nr] >= 0)[0].astype(np.int32)
    return dict(block=block, send_rows=send, adj_src=remapped, nodes=rep, row_begin=rb, n_rows=nr, e0=e0, e1=e1)


def shard_csr(n_nodes: int, rank: int, world: int, indptr, *per_entry_arrays):
    """Node-range shard of a CSR-by-destination graph: (row_begin, n_rows, local indptr, sliced per-entry arrays...).
    Source ids inside the slices stay GLOBAL (every rank keeps a full replica of the state)."""
    rb, nr = shard_range(n_nodes, rank, world)
    indptr = np.asarray(indptr)
    e0, e1 = int(indptr[rb]), int(indptr[rb + nr])
    return (rb, nr, (indptr[rb:rb + nr + 1] - e0).astype(np.int32)) + tuple(np.asarray(a)[e0:e1] for a in per_entry_arrays)


class Graph:
    """Device-resident graph (gnn_graph).  CSR "by destination" as produced by graph_class.GraphTensor."""

    def __init__(self, n_nodes, indptr, adj_src, adj_w, arc_w, arc_labels, nodes, mask, row_begin=0, device=0, _handle=None):
        self._h = C.c_void_p()
        if _handle is not None:
            self._h = _handle
            return
        require_device(device)
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        adj_src = np.ascontiguousarray(adj_src, dtype=np.int32)
        adj_w, arc_w = _f32(adj_w), _f32(arc_w)
        arc_labels, nodes = _f32(arc_labels), _f32(nodes)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        n_rows, n_arcs = len(indptr) - 1, len(adj_src)
        if arc_labels.ndim != 2 or arc_labels.shape[0] != n_arcs or nodes.ndim != 2 or nodes.shape[0] != n_nodes:
            raise ValueError('arc_labels must be [n_arcs, AL] and nodes [n_nodes, NL]')
        if len(adj_w) != n_arcs or len(arc_w) != n_arcs or len(mask) != n_rows:
            raise ValueError('inconsistent array lengths')
        _check(lib().gnn_graph_create(C.c_int64(n_nodes), C.c_int64(row_begin), C.c_int64(n_rows), C.c_int64(n_arcs),
                                      _ip(indptr), _ip(adj_src), _fp(adj_w), _fp(arc_w), _fp(arc_labels),
                                      C.c_int(arc_labels.shape[1]), _fp(nodes), C.c_int(nodes.shape[1]),
                                      mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(device), C.byref(self._h)))

    @classmethod
    def halo(cls, n_nodes_global, rank, world, block, send_rows, indptr, adj_src_replica, adj_w, arc_w, arc_labels, nodes_replica, mask, device=0):
        """gnn_graph_create_halo: shard whose per-iteration exchange moves only boundary rows (see shard_halo)."""
        require_device(device)
        indptr = np.ascontiguousarray(indptr, dtype=np.int32)
        adj = np.ascontiguousarray(adj_src_replica, dtype=np.int32)
        send = np.ascontiguousarray(send_rows, dtype=np.int32)
        adj_w, arc_w, arc_labels, nodes_replica = _f32(adj_w), _f32(arc_w), _f32(arc_labels), _f32(nodes_replica)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        if len(adj_w) != len(adj) or len(arc_w) != len(adj) or arc_labels.shape[0] != len(adj) or len(mask) != len(indptr) - 1:
            raise ValueError('inconsistent array lengths')
        h = C.c_void_p()
        _check(lib().gnn_graph_create_halo(C.c_int64(n_nodes_global), C.c_int(rank), C.c_int(world), C.c_int64(block), C.c_int64(len(send)), _ip(send),
                                           C.c_int64(len(adj)), _ip(indptr), _ip(adj), _fp(adj_w), _fp(arc_w), _fp(arc_labels), C.c_int(arc_labels.shape[1]),
                                           _fp(nodes_replica), C.c_int(nodes_replica.shape[1]), mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(device), C.byref(h)))
        return cls(None, None, None, None, None, None, None, None, _handle=h)

    @classmethod
    def from_arcs(cls, n_nodes, arc_src, arc_dst, arc_labels, aggregation_mode, nodes, mask, device=0):
        """gnn_graph_create_from_arcs: device-side COO -> CSR build.  Returns (Graph, indptr, adj_src, adj_w, arc_id, arc_w):
        the handle and the host mirrors of Adjacency^T / ArcNode^T."""
        require_device(device)
        modes = {'sum': 0, 'normalized': 1, 'average': 2}
        if aggregation_mode not in modes: raise ValueError('ERROR: Unknown aggregation mode')
        arc_src = np.ascontiguousarray(arc_src, dtype=np.int32)
        arc_dst = np.ascontiguousarray(arc_dst, dtype=np.int32)
        arc_labels, nodes = _f32(arc_labels), _f32(nodes)
        mask = np.ascontiguousarray(mask, dtype=np.uint8)
        e = len(arc_src)
        if len(arc_dst) != e or arc_labels.shape[0] != e or nodes.shape[0] != n_nodes or len(mask) != n_nodes:
            raise ValueError('inconsistent array lengths')
        indptr, adj_src, arc_id = np.zeros(n_nodes + 1, np.int32), np.zeros(e, np.int32), np.zeros(e, np.int32)
        adj_w, arc_w = np.zeros(e, np.float32), np.zeros(e, np.float32)
        h = C.c_void_p()
        _check(lib().gnn_graph_create_from_arcs(C.c_int64(n_nodes), C.c_int64(e), _ip(arc_src), _ip(arc_dst), _fp(arc_labels),
                                                C.c_int(arc_labels.shape[1]), C.c_int(modes[aggregation_mode]), _fp(nodes), C.c_int(nodes.shape[1]),
                                                mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.c_int(device), C.byref(h), _ip(indptr), _ip(adj_src),
                                                _fp(adj_w), _ip(arc_id), _fp(arc_w)))
        return cls(None, None, None, None, None, None, None, None, _handle=h), indptr, adj_src, adj_w, arc_id, arc_w

    def dims(self):
        n, r, e, m = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int64()
        nl, al = C.c_int(), C.c_int()
        _check(lib().gnn_graph_dims(self._h, C.byref(n), C.byref(r), C.byref(e), C.byref(nl), C.byref(al), C.byref(m)))
        return dict(n_nodes=n.value, n_rows=r.value, n_arcs=e.value, NL=nl.value, AL=al.value, n_masked=m.value)

    def derive(self, extra: int) -> 'Graph':
        h = C.c_void_p()
        _check(lib().gnn_graph_derive(self._h, C.c_int(extra), C.byref(h)))
        d = Graph(None, None, None, None, None, None, None, None, _handle=h)
        d._base = self            # a derived graph uses arrays of its base (CSR, boundary rows): keep it alive
        return d

    def derive_edge(self, extra_nodes: int, extra_arcs: int) -> 'Graph':
        h = C.c_void_p()
        _check(lib().gnn_graph_derive_edge(self._h, C.c_int(extra_nodes), C.c_int(extra_arcs), C.byref(h)))
        d = Graph(None, None, None, None, None, None, None, None, _handle=h)
        d._base = self
        return d

    def set_arc_order(self, arc_id, arc_labels_orig) -> None:
        arc_id = np.ascontiguousarray(arc_id, dtype=np.int32)
        _check(lib().gnn_graph_set_arc_order(self._h, _ip(arc_id), _fp(_f32(arc_labels_orig))))

    def update_labels(self, base: 'Graph', loop: 'Loop', get_state: bool, get_output: bool) -> None:
        _check(lib().gnn_graph_update_labels(self._h, base._h, loop._h, C.c_int(bool(get_state)), C.c_int(bool(get_output))))

    def set_full_adjacency(self, n_global: int, indptr, adj_src, adj_w):
        """gnn_graph_set_full_adjacency: the whole graph's CSR by destination (global ids) for the feature-sliced exchange."""
        ip, src, w = np.ascontiguousarray(indptr, np.int32), np.ascontiguousarray(adj_src, np.int32), _f32(adj_w)
        _check(lib().gnn_graph_set_full_adjacency(self._h, C.c_int64(n_global), _ip(ip), _ip(src), _fp(w)))

    @staticmethod
    def update_labels_group(dsts, bases, loops, get_state: bool, get_output: bool) -> None:
        """gnn_graph_update_labels_group: the relabelling step for all ranks of a loopback group."""
        n = len(loops)
        arr = lambda xs: (C.c_void_p * n)(*[x._h for x in xs])
        _check(lib().gnn_graph_update_labels_group(arr(dsts), arr(bases), arr(loops), C.c_int(n), C.c_int(bool(get_state)), C.c_int(bool(get_output))))

    def nodes(self) -> np.ndarray:
        d = self.dims()
        out = np.empty((d['n_nodes'], d['NL']), dtype=np.float32)
        _check(lib().gnn_graph_get_nodes(self._h, _fp(out)))
        return out

    def close(self):
        if self._h:
            lib().gnn_graph_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Mlp:
    """Device-resident Sequential (gnn_mlp): Keras get_weights() list + activation names."""

    def __init__(self, weights, activations, batch_normalization, bn_eps=1e-3, device=0):
        require_device(device)
        self._h = C.c_void_p()
        self.n = len(activations)
        self.activations = list(activations)
        self.batch_normalization = bool(batch_normalization)
        for a in activations:
            if a not in ACT_CODES:
                raise ValueError(f'unsupported activation {a!r}')
        w, b, bn = self._pack(weights)
        self.dims = np.array([w[0].shape[0]] + [x.shape[1] for x in w], dtype=np.int32)
        acts = np.array([ACT_CODES[a] for a in activations], dtype=np.int32)
        wp = (C.POINTER(C.c_float) * self.n)(*[_fp(x) for x in w])
        bp = (C.POINTER(C.c_float) * self.n)(*[_fp(x) for x in b])
        _check(lib().gnn_mlp_create(C.c_int(self.n), _ip(self.dims), _ip(acts), wp, bp, _fp(bn), C.c_float(bn_eps),
                                    C.c_int(device), C.byref(self._h)))

    def _pack(self, weights):
        n = self.n
        if len(weights) != 2 * n + (4 if self.batch_normalization else 0):
            raise ValueError(f'expected {2 * n + (4 if self.batch_normalization else 0)} weight arrays, got {len(weights)}')
        w = [_f32(weights[2 * l]) for l in range(n)]
        b = [_f32(weights[2 * l + 1]) for l in range(n)]
        for l in range(n):
            if w[l].ndim != 2 or b[l].shape != (w[l].shape[1],) or (l and w[l].shape[0] != w[l - 1].shape[1]):
                raise ValueError(f'layer {l}: inconsistent weight shapes')
        bn = None
        if self.batch_normalization:
            bn = np.ascontiguousarray(np.concatenate([_f32(a).ravel() for a in weights[2 * n:]]))
            if bn.size != 4 * w[-1].shape[1]:
                raise ValueError('BatchNormalization arrays must each have the output width')
        return w, b, bn

    def set_weights(self, weights):
        w, b, bn = self._pack(weights)
        if [x.shape for x in w] != [(int(self.dims[l]), int(self.dims[l + 1])) for l in range(self.n)]:
            raise ValueError('weight shapes differ from the ones this MLP was created with')
        wp = (C.POINTER(C.c_float) * self.n)(*[_fp(x) for x in w])
        bp = (C.POINTER(C.c_float) * self.n)(*[_fp(x) for x in b])
        _check(lib().gnn_mlp_set_weights(self._h, wp, bp, _fp(bn)))

    def get_weights(self):
        """gnn_mlp_get_weights: the Keras get_weights() list as it is on the device now."""
        w = [np.empty((int(self.dims[l]), int(self.dims[l + 1])), np.float32) for l in range(self.n)]
        b = [np.empty((int(self.dims[l + 1]),), np.float32) for l in range(self.n)]
        f = int(self.dims[-1])
        bn = np.empty(4 * f, np.float32) if self.batch_normalization else None
        wp = (C.POINTER(C.c_float) * self.n)(*[_fp(x) for x in w])
        bp = (C.POINTER(C.c_float) * self.n)(*[_fp(x) for x in b])
        _check(lib().gnn_mlp_get_weights(self._h, wp, bp, _fp(bn)))
        out = []
        for l in range(self.n):
            out += [w[l], b[l]]
        if bn is not None:
            out += [bn[i * f:(i + 1) * f].copy() for i in range(4)]
        return out

    def reset_optimizer(self):
        """gnn_mlp_reset_optimizer: zero optimizer slots for the next device-side step (a new optimizer object took over)."""
        _check(lib().gnn_mlp_reset_optimizer(self._h))

    def forward(self, x):
        x = _f32(x)
        if x.ndim != 2 or x.shape[1] != self.dims[0]:
            raise ValueError(f'expected [n, {self.dims[0]}] input')
        y = np.empty((x.shape[0], int(self.dims[-1])), dtype=np.float32)
        _check(lib().gnn_mlp_forward(self._h, C.c_int64(x.shape[0]), _fp(x), _fp(y)))
        return y

    def close(self):
        if self._h:
            lib().gnn_mlp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Comm:
    """RCCL communicator (one process per GPU), or one member of an in-process loopback group (Comm.loopback)."""

    def __init__(self, unique_id: bytes, rank: int, world: int, device: int, _handle=None):
        self._h = C.c_void_p()
        self.rank, self.world = rank, world
        if _handle is not None:
            self._h = _handle
            return
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        _check(lib().gnn_comm_create(buf, C.c_int(rank), C.c_int(world), C.c_int(device), C.byref(self._h)))

    @classmethod
    def loopback(cls, world: int, device: int = 0) -> list:
        """gnn_comm_create_loopback: `world` communicators on ONE device (the sharded path on a single GPU; tests only)."""
        require_device(device)
        hs = (C.c_void_p * world)()
        _check(lib().gnn_comm_create_loopback(C.c_int(world), C.c_int(device), hs))
        return [cls(None, r, world, device, _handle=C.c_void_p(hs[r])) for r in range(world)]

    @staticmethod
    def unique_id() -> bytes:
        buf = (C.c_uint8 * 128)()
        _check(lib().gnn_comm_unique_id(buf))
        return bytes(buf)

    def allreduce_max(self, value: float) -> float:
        v = C.c_double(value)
        _check(lib().gnn_comm_allreduce_max(self._h, C.byref(v)))
        return v.value

    def close(self):
        if self._h:
            lib().gnn_comm_destroy(self._h)
            self._h = C.c_void_p()


class Loop:
    """One configured GNN.Loop on the device (gnn_loop)."""

    def __init__(self, graph: Graph, net_state: Mlp, net_output: Mlp, state_dim: int, max_iter: int, threshold: float,
                 comm: Comm | None = None):
        self._h = C.c_void_p()
        self._keep = (graph, net_state, net_output, comm)
        self.graph = graph
        d = graph.dims()
        self.n_rows, self.n_masked = d['n_rows'], d['n_masked']
        self.Ds = state_dim if state_dim else d['NL']
        self.T = int(net_output.dims[-1])
        self.state_dim = state_dim
        self.max_iter = int(max_iter)
        _check(lib().gnn_loop_create(graph._h, net_state._h, net_output._h, C.c_int(state_dim), C.c_int(max_iter),
                                     C.c_float(threshold), comm._h if comm else None, C.byref(self._h)))

    def set_state0(self, state0=None, seed: int = 0):
        if state0 is not None:
            state0 = _f32(state0, (self.n_rows, self.Ds))
        _check(lib().gnn_loop_set_state0(self._h, _fp(state0), C.c_uint64(seed)))

    def set_edge_readout(self, entry_dst, arc_labels, arc_mask):
        """Switch to the per-arc readout of GNNedgeBased (reference GNN.py:289-302)."""
        entry_dst = np.ascontiguousarray(entry_dst, dtype=np.int32)
        arc_labels = _f32(arc_labels) if arc_labels is not None else None      # None: the (derived) graph owns its arc labels
        arc_mask = np.ascontiguousarray(arc_mask, dtype=np.uint8)
        if not (len(entry_dst) == len(arc_mask)) or (arc_labels is not None and arc_labels.shape[0] != len(arc_mask)):
            raise ValueError('entry_dst, arc_labels and arc_mask must have one row per arc')
        _check(lib().gnn_loop_set_edge_readout(self._h, _ip(entry_dst), _fp(arc_labels) if arc_labels is not None else None,
                                               arc_mask.ctypes.data_as(C.POINTER(C.c_uint8))))
        self.n_masked = int(arc_mask.sum())

    def train_step(self, net_state: 'Mlp', net_output: 'Mlp', src_csr, targets, sample_weights, loss_kind: int, ng_csr=None,
                   dropout_state=None, dropout_output=None, masks_state=None, masks_output=None, seed: int = 0,
                   bn_state=None, bn_output=None):
        """gnn_loop_train_step: loss, iteration count, raw gradients (lists shaped like the trainable arrays) and the
        BatchNormalization batch statistics of every call."""
        sip, sdst, sw = (np.ascontiguousarray(src_csr[0], np.int32), np.ascontiguousarray(src_csr[1], np.int32), _f32(src_csr[2])) if src_csr is not None else (None, None, None)
        targets, sample_weights = _f32(targets), _f32(sample_weights)
        ls, lo = net_state.n, net_output.n
        ds_ = _f32(dropout_state if dropout_state is not None else np.zeros(ls + 1))
        do_ = _f32(dropout_output if dropout_output is not None else np.zeros(lo + 1))

        def shapes(net):
            out = []
            for l in range(net.n):
                out += [(int(net.dims[l]), int(net.dims[l + 1])), (int(net.dims[l + 1]),)]
            if net.batch_normalization:
                out += [(int(net.dims[-1]),), (int(net.dims[-1]),)]
            return out

        shp_s, shp_o = shapes(net_state), shapes(net_output)
        gs = np.zeros(sum(int(np.prod(x)) for x in shp_s), np.float32)
        go = np.zeros(sum(int(np.prod(x)) for x in shp_o), np.float32)
        fs, fo = int(net_state.dims[-1]), int(net_output.dims[-1])
        bns = np.zeros((max(1, self.max_iter), 2, fs), np.float32)      # the C side writes k <= max_iteration rows
        bno = np.zeros((2, fo), np.float32)
        ms = np.ascontiguousarray(masks_state, np.uint8) if masks_state is not None else None
        mo = np.ascontiguousarray(masks_output, np.uint8) if masks_output is not None else None
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None
        if ng_csr is not None:
            ngi, ngn, ngw = np.ascontiguousarray(ng_csr[0], np.int32), np.ascontiguousarray(ng_csr[1], np.int32), _f32(ng_csr[2])
            n_graphs = len(ngi) - 1
        else:
            ngi = ngn = ngw = None
            n_graphs = 0
        bs = _f32(bn_state) if bn_state is not None else None
        bo = _f32(bn_output) if bn_output is not None else None
        loss, k = C.c_float(), C.c_float()
        _check(lib().gnn_loop_train_step(self._h, _ip(sip), _ip(sdst), _fp(sw), _fp(targets), _fp(sample_weights),
                                         C.c_int64(targets.shape[0]), C.c_int(loss_kind), C.c_int(n_graphs), _ip(ngi), _ip(ngn), _fp(ngw),
                                         _fp(ds_), _fp(do_), u8(ms), u8(mo), C.c_uint64(seed), _fp(bs), _fp(bo), C.byref(loss), C.byref(k),
                                         _fp(gs), _fp(go), _fp(bns), _fp(bno)))

        def split(flat, shp):
            out, off = [], 0
            for x in shp:
                cnt = int(np.prod(x))
                out.append(flat[off:off + cnt].reshape(x).copy())
                off += cnt
            return out

        kk = int(k.value)
        return dict(loss=float(loss.value), k=float(k.value), grads_state=split(gs, shp_s), grads_output=split(go, shp_o),
                    bn_batch_state=bns[:kk], bn_batch_output=bno)

    def set_slice_exchange(self, on=True, form=None):
        """gnn_loop_set_slice_exchange: feature-sliced all-to-all instead of the all-gather of state rows.  on: any truthy value (True, 1,
        numpy.bool_) switches it on in the default form, falsy switches it off.  form (keyword, only with on): 'oneshot' (default: whole slice,
        then one grouped all-to-all) or 'pipelined' (the return all-to-all block by block beside the aggregation on a second stream of the
        communicator - bit-identical in loopback groups and over the tests' stand-in transport, but never yet run over RCCL with more than
        one rank, hence never implied by `on`)."""
        if form not in (None, 'oneshot', 'pipelined'):
            raise ValueError("form must be 'oneshot' or 'pipelined'")
        code = 0 if not on else (1 if form == 'pipelined' else 2)
        _check(lib().gnn_loop_set_slice_exchange(self._h, C.c_int(code)))

    def update_moving_statistics(self, bn_momentum_state: float = 0.99, bn_momentum_output: float = 0.99):
        """gnn_loop_update_moving_statistics: the moving statistics of both nets from the last train_forward, on the device."""
        _check(lib().gnn_loop_update_moving_statistics(self._h, C.c_float(bn_momentum_state), C.c_float(bn_momentum_output)))

    def arm_optimizer(self, kind: int, hyper, mean: bool, bn_momentum_state: float = 0.99, bn_momentum_output: float = 0.99):
        """gnn_loop_arm_optimizer: the next train_step() also applies the optimizer update on the device."""
        h = _f32(np.asarray(list(hyper) + [0.0] * (4 - len(hyper)), np.float32))
        _check(lib().gnn_loop_arm_optimizer(self._h, C.c_int(kind), _fp(h), C.c_int(1 if mean else 0), C.c_float(bn_momentum_state),
                                            C.c_float(bn_momentum_output)))

    def optimizer_step(self, kind: int, hyper, state_grad_scale: float = 1.0, bn_momentum_state: float = 0.99, bn_momentum_output: float = 0.99):
        """gnn_loop_optimizer_step: apply the gradients of the last backward pass on the device."""
        h = _f32(np.asarray(list(hyper) + [0.0] * (4 - len(hyper)), np.float32))
        _check(lib().gnn_loop_optimizer_step(self._h, C.c_int(kind), _fp(h), C.c_float(state_grad_scale), C.c_float(bn_momentum_state),
                                             C.c_float(bn_momentum_output)))

    @staticmethod
    def _grad_shapes(net: 'Mlp'):
        out = []
        for l in range(net.n):
            out += [(int(net.dims[l]), int(net.dims[l + 1])), (int(net.dims[l + 1]),)]
        if net.batch_normalization:
            out += [(int(net.dims[-1]),), (int(net.dims[-1]),)]
        return out

    def train_forward(self, net_state: 'Mlp', net_output: 'Mlp', src_csr, dropout_state=None, dropout_output=None,
                      masks_state=None, masks_output=None, seed: int = 0, bn_state=None, bn_output=None):
        """gnn_loop_train_forward: training-mode Loop; returns (k, node-level outputs [n_masked, T]).  The loop's state() /
        output() / readout() then hold the training-mode results, and train_backward() may be called once."""
        sip, sdst, sw = (np.ascontiguousarray(src_csr[0], np.int32), np.ascontiguousarray(src_csr[1], np.int32), _f32(src_csr[2])) if src_csr is not None else (None, None, None)
        ds_ = _f32(dropout_state if dropout_state is not None else np.zeros(net_state.n + 1))
        do_ = _f32(dropout_output if dropout_output is not None else np.zeros(net_output.n + 1))
        ms = np.ascontiguousarray(masks_state, np.uint8) if masks_state is not None else None
        mo = np.ascontiguousarray(masks_output, np.uint8) if masks_output is not None else None
        u8 = lambda a: a.ctypes.data_as(C.POINTER(C.c_uint8)) if a is not None else None
        bs = _f32(bn_state) if bn_state is not None else None
        bo = _f32(bn_output) if bn_output is not None else None
        k = C.c_float()
        out = np.zeros((self.n_masked, self.T), np.float32)
        _check(lib().gnn_loop_train_forward(self._h, _ip(sip), _ip(sdst), _fp(sw), _fp(ds_), _fp(do_), u8(ms), u8(mo), C.c_uint64(seed),
                                            _fp(bs), _fp(bo), C.byref(k), _fp(out)))
        self._train_nets = (net_state, net_output, int(k.value))
        return float(k.value), out

    def train_backward(self, d_out_nodes, d_state_extra=None, want_d_nodes: bool = False, want_d_arcs: bool = False):
        """gnn_loop_train_backward: dict(grads_state, grads_output (raw sums over the iterations), bn_batch_state,
        bn_batch_output, d_nodes [N, NL] or None)."""
        net_state, net_output, kk = self._train_nets
        shp_s, shp_o = self._grad_shapes(net_state), self._grad_shapes(net_output)
        gs = np.zeros(sum(int(np.prod(x)) for x in shp_s), np.float32)
        go = np.zeros(sum(int(np.prod(x)) for x in shp_o), np.float32)
        bns = np.zeros((max(1, kk), 2, int(net_state.dims[-1])), np.float32)
        bno = np.zeros((2, int(net_output.dims[-1])), np.float32)
        d_out = _f32(d_out_nodes)
        if d_out.shape != (self.n_masked, self.T): raise ValueError(f'd_out_nodes shape {d_out.shape} != {(self.n_masked, self.T)}')
        dse = None
        if d_state_extra is not None:
            dse = _f32(d_state_extra)
            if dse.shape != (self.n_rows, self.Ds): raise ValueError(f'd_state_extra shape {dse.shape} != {(self.n_rows, self.Ds)}')
        dims = self.graph.dims()
        dn = np.zeros((self.n_rows, dims['NL']), np.float32) if want_d_nodes else None
        da = np.zeros((dims['n_arcs'], dims['AL']), np.float32) if want_d_arcs else None
        _check(lib().gnn_loop_train_backward(self._h, _fp(d_out), _fp(dse), _fp(gs), _fp(go), _fp(bns), _fp(bno), _fp(dn), _fp(da)))

        def split(flat, shp):
            out, off = [], 0
            for x in shp:
                cnt = int(np.prod(x))
                out.append(flat[off:off + cnt].reshape(x).copy())
                off += cnt
            return out

        return dict(grads_state=split(gs, shp_s), grads_output=split(go, shp_o), bn_batch_state=bns[:kk], bn_batch_output=bno, d_nodes=dn, d_arcs=da)

    def set_impl(self, impl: int) -> int:
        used = C.c_int(0)
        _check(lib().gnn_loop_set_impl(self._h, C.c_int(impl), C.byref(used)))
        return used.value

    def gate_info(self) -> tuple:
        """gnn_loop_gate_info: (the last run was repeated on the bit-exact path because a gate of the default path was not certified, how
        often that has happened on this loop)."""
        a, b = C.c_int(0), C.c_int(0)
        _check(lib().gnn_loop_gate_info(self._h, C.byref(a), C.byref(b)))
        return bool(a.value), int(b.value)

    def set_persistent(self, enable: bool) -> bool:
        """gnn_loop_set_persistent: allow / forbid the one-launch-per-Loop path of small graphs; returns whether it will be used."""
        used = C.c_int(0)
        _check(lib().gnn_loop_set_persistent(self._h, C.c_int(bool(enable)), C.byref(used)))
        return bool(used.value)

    def set_tile_form(self, form: int) -> int:
        """gnn_loop_set_tile_form: 1 = one wave per 32-node tile, 2 = a wave pair per tile, 0 = the library's choice; returns the form the
        next run takes (0: the fused path does not cover this loop)."""
        used = C.c_int(0)
        _check(lib().gnn_loop_set_tile_form(self._h, C.c_int(int(form)), C.byref(used)))
        return used.value

    def counters(self) -> dict:
        """gnn_counters_get: algorithmic bytes / FLOPs of one iteration on the owned rows, and the last run's iteration count and times."""
        b, f, it, tot, avg = C.c_double(), C.c_double(), C.c_int(), C.c_float(), C.c_float()
        _check(lib().gnn_counters_get(self._h, C.byref(b), C.byref(f), C.byref(it), C.byref(tot), C.byref(avg)))
        return dict(bytes_per_iteration=b.value, flops_per_iteration=f.value, iterations=it.value, total_ms=tot.value, avg_iteration_ms=avg.value)

    @staticmethod
    def lgnn_run(loops, graphs, get_state: bool, get_output: bool) -> list:
        """gnn_lgnn_run: LGNN.Loop of a whole stack in one call (loops[i] created on graphs[i], graphs[0] the original graph)."""
        n = len(loops)
        ks = (C.c_float * n)()
        _check(lib().gnn_lgnn_run((C.c_void_p * n)(*[x._h for x in loops]), (C.c_void_p * n)(*[x._h for x in graphs]), C.c_int(n),
                                  C.c_int(bool(get_state)), C.c_int(bool(get_output)), ks))
        return [float(k) for k in ks]

    def drop_cached_aggregates(self):
        _check(lib().gnn_loop_drop_cached_aggregates(self._h))

    def set_profiling(self, on: bool):
        _check(lib().gnn_loop_set_profiling(self._h, C.c_int(bool(on))))

    def run(self, training: bool = False) -> float:
        k = C.c_float(0)
        _check(lib().gnn_loop_run(self._h, C.c_int(bool(training)), C.byref(k)))
        return float(k.value)

    @staticmethod
    def run_many(loops) -> list:
        """gnn_loop_run_many: independent loops (batches of a dataset) in one call; small graphs run side by side.  Returns [k per loop]."""
        n = len(loops)
        hs = (C.c_void_p * n)(*[l._h for l in loops])
        ks = (C.c_float * n)()
        _check(lib().gnn_loop_run_many(hs, C.c_int(n), ks))
        return [float(x) for x in ks]

    @staticmethod
    def run_group(loops) -> float:
        """gnn_loop_run_group: one Loop on all ranks of a loopback group (rank order)."""
        n = len(loops)
        hs = (C.c_void_p * n)(*[l._h for l in loops])
        k = C.c_float(0)
        _check(lib().gnn_loop_run_group(hs, C.c_int(n), C.byref(k)))
        return float(k.value)

    @staticmethod
    def readout_group(loops, ng_indptr, ng_node, ng_w) -> np.ndarray:
        n = len(loops)
        hs = (C.c_void_p * n)(*[l._h for l in loops])
        ng_indptr = np.ascontiguousarray(ng_indptr, dtype=np.int32)
        ng_node = np.ascontiguousarray(ng_node, dtype=np.int32)
        ng_w = _f32(ng_w)
        g = len(ng_indptr) - 1
        out = np.empty((g, loops[0].T), dtype=np.float32)
        _check(lib().gnn_loop_readout_group(hs, C.c_int(n), C.c_int(g), _ip(ng_indptr), _ip(ng_node), _fp(ng_w), _fp(out)))
        return out

    def timing(self):
        tot, avg, n = C.c_float(), C.c_float(), C.c_int()
        _check(lib().gnn_loop_get_timing(self._h, C.byref(tot), C.byref(avg), C.byref(n)))
        gap = C.c_float()
        _check(lib().gnn_loop_get_exchange_timing(self._h, C.byref(gap)))
        return dict(total_ms=tot.value, avg_iter_ms=avg.value, n_iter_timed=n.value, avg_between_bodies_ms=gap.value)

    def state(self) -> np.ndarray:
        out = np.empty((self.n_rows, self.Ds), dtype=np.float32)
        _check(lib().gnn_loop_get_state(self._h, _fp(out)))
        return out

    def output(self) -> np.ndarray:
        out = np.empty((self.n_masked, self.T), dtype=np.float32)
        m = C.c_int64(0)
        _check(lib().gnn_loop_get_output(self._h, _fp(out), C.byref(m)))
        return out

    def readout(self, ng_indptr, ng_node, ng_w) -> np.ndarray:
        ng_indptr = np.ascontiguousarray(ng_indptr, dtype=np.int32)
        ng_node = np.ascontiguousarray(ng_node, dtype=np.int32)
        ng_w = _f32(ng_w)
        g = len(ng_indptr) - 1
        out = np.empty((g, self.T), dtype=np.float32)
        _check(lib().gnn_loop_readout(self._h, C.c_int(g), _ip(ng_indptr), _ip(ng_node), _fp(ng_w), _fp(out)))
        return out

    def close(self):
        if self._h:
            lib().gnn_loop_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
