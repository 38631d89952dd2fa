"""The 64-node-tile form of the fused iteration kernel (csrc/gnn_fused64_kernel.h: one wave per SIMD, every weight fragment feeding two
32-node halves, the next tile's gather inside the current tile's matrix phase through an LDS-DMA ring) against the 32-node-tile form
(k_fused) and the C oracle.  Both forms evaluate the same products in the same order per accumulator: states, outputs and k must be
BIT-IDENTICAL between them (reference loop: GNN/GNN.py:202-280); against the oracle the split arithmetic's tolerance applies."""
import numpy as np
import pytest

from oracle import c_oracle as corc
from oracle import gnn_oracle as orc
from util import make_mlp, random_arcs

pytestmark = pytest.mark.gpu


def _engine():
    from GNN import _engine
    return _engine


def _device_graph(g):
    e = _engine()
    mask = np.logical_and(g['set_mask'], g['output_mask'])
    arc_labels = np.asarray(g['arcs'], np.float32)[:, 2:]
    return e.Graph(g['nodes'].shape[0], g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], arc_labels[g['arcT'][1]], g['nodes'], mask)


def _graph(rng, n, nl, al, deg, skew=False):
    if not skew:
        arcs = random_arcs(rng, n, deg * n, al, sort=True)
    else:       # hubs with thousands of in-arcs, runs of isolated nodes, whole tiles of rows with exactly 16 / 17 entries
        d = np.zeros(n, np.int64)
        d[rng.choice(n, 5, replace=False)] = rng.integers(1500, 3000, 5)
        body = rng.choice(n, n // 2, replace=False)
        d[body] = np.maximum(d[body], rng.integers(1, 40, len(body)))
        d[128:192] = 16; d[192:256] = 17; d[256:448] = 0
        dst = np.repeat(np.arange(n), d)
        src = rng.integers(0, n, len(dst))
        keep = src != dst
        pairs = np.unique(np.stack([src[keep], dst[keep]], 1), axis=0)
        arcs = np.concatenate([pairs.astype(np.float32), (2 * rng.random((len(pairs), al)) - 1).astype(np.float32)], axis=1)
    nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
    return orc.make_graph_dict(arcs, nodes, 'average')


@pytest.mark.parametrize('n,nl,al,hidden,act,deg,skew', [
    (4096, 3, 1, (128, 128), 'selu', 10, False),          # BASELINE configs[2] shape
    (4096, 5, 1, (128, 128), 'selu', 10, False),          # configs[4] layer > 0 (labels widened by the previous output)
    (1024, 3, 2, (128,), 'tanh', 4, False),               # two layers, a gather event at every weight unit
    (2048, 3, 1, (96, 128), 'tanh', 7, True),             # hidden width below 128 (zero-padded feature tiles), skewed degrees
    (640, 6, 2, (128, 128), 'selu', 30, False),           # wide label block, dense rows: most of the gather in the clean-up loop
    (64, 3, 1, (128, 128), 'selu', 3, False),             # one tile: nothing to pipeline
])
def test_wide_tiles_bit_identical_to_32_node_tiles(n, nl, al, hidden, act, deg, skew):
    e = _engine()
    rng = np.random.default_rng(900 + n + nl)
    d = 64
    g = _graph(rng, n, nl, al, deg, skew)
    g['set_mask'] = rng.random(n) < 0.9
    st = make_mlp(rng, al + 2 * (nl + d), list(hidden) + [d], act, gain=0.6, bn_random=True)
    ou = make_mlp(rng, nl + d, [2], 'softmax', bn_random=True)
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    graph = _device_graph(g)
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    for thr, max_it in ((0.0, 5), (0.01, 12)):
        kc, sc, oc = corc.loop_node(g, st, ou, d, max_it, thr, s0)
        res = {}
        for mode in (1, 2):
            loop = e.Loop(graph, mst, mou, d, max_it, thr)
            assert loop.set_impl(2) == 2
            loop.set_tile_shape(mode)
            loop.set_state0(s0)
            k = loop.run()
            assert loop.gate_info()[0] or loop.tile_shape() == (32 if mode == 1 else 64)
            res[mode] = (k, loop.state(), loop.output())
            k_again = loop.run()                         # a second run on the same handle (ring / image state left by the first)
            assert k_again == k and np.array_equal(loop.state(), res[mode][1])
            loop.close()
        assert res[1][0] == res[2][0] == kc, (res[1][0], res[2][0], kc)
        assert np.array_equal(res[1][1], res[2][1]), f'{int(np.sum(res[1][1] != res[2][1]))} of {res[1][1].size} state values differ between the tile shapes'
        assert np.array_equal(res[1][2], res[2][2])
        err = float(np.max(np.abs(res[2][1] - sc)))
        assert err < 2e-6 * max(1.0, float(np.max(np.abs(sc)))), err


def test_wide_tiles_selected_automatically_and_fall_back():
    """Default policy: 64-node tiles from 262,144 owned rows on; a row count that is not a multiple of 64, another state width or a narrow net
    stay with the 32-node kernel whatever is asked."""
    e = _engine()
    from GNN import GNN_utils as utils
    rng = np.random.default_rng(5)
    d, nl, al = 64, 3, 1
    st = make_mlp(rng, al + 2 * (nl + d), [128, 128, d], 'selu', gain=0.6, bn_random=True)
    ou = make_mlp(rng, nl + d, [2], 'softmax', bn_random=True)
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    for n, expect in ((270_336, 64), (270_337, 32), (100_032, 32)):
        s = utils.syntheticGraph(n, 10.0, nl, al, 2, seed=3)
        graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
        loop = e.Loop(graph, mst, mou, d, 3, 0.0)
        loop.set_state0(None, seed=1)
        assert loop.run() == 3 and loop.tile_shape() == expect, (n, loop.tile_shape())
        if expect == 64:                                 # ... and the same bits as the 32-node form at this size
            s64 = loop.state()
            loop.set_tile_shape(1)
            assert loop.run() == 3 and loop.tile_shape() == 32 and np.array_equal(loop.state(), s64)
        loop.close(); graph.close()
    # a net with 64-wide hidden layers: not covered, stays on 32-node tiles even when asked
    st2 = make_mlp(rng, al + 2 * (nl + d), [64, d], 'selu', gain=0.6, bn_random=True)
    s = utils.syntheticGraph(4096, 10.0, nl, al, 2, seed=3)
    graph = e.Graph(4096, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(4096, np.uint8))
    loop = e.Loop(graph, e.Mlp(st2['weights'], st2['activations'], True), mou, d, 3, 0.0)
    loop.set_tile_shape(2)
    loop.set_state0(None, seed=1)
    assert loop.run() == 3 and loop.tile_shape() == 32
    loop.close(); graph.close()
