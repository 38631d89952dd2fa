"""The k contract of the default path (impl 2) where it could break: loops that STOP deep (k in [15, 30]) on slowly converging and
non-contractive state maps, and the zero-node corner of the certified gate (csrc/gnn_common.h; reference GNN/GNN.py:202-220, :271)."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'tools'))

pytestmark = pytest.mark.gpu


def _engine():
    from GNN import _engine
    return _engine


def test_certified_gate_on_deep_stops_of_non_contractive_maps():
    """Gains 0.9 / 1.0 / 1.1 (tanh: the trajectory keeps moving for tens of bodies; selu: near-neutral, 1.1 diverges and is skipped by
    construction), 100,000 nodes, 20 seeds; thresholds taken from the exact chain's own per-body maximum of distance / norm so that the
    Loop stops at a body in [15, 30): just above that body's maximum (2 %) and between it and the previous minimum.  Required: the k of
    the default path equals the exact chain's k on every run (a repeat on impl 1 is allowed - it is the mechanism - but must be rare),
    and the measured divergence of the two arithmetics at the stop stays below HALF the gate's band (so that no gate can flip
    unnoticed); the numbers go to stdout for the record (profiles/r05_gate_study.txt holds a run of tools/gate_study.py)."""
    import gate_study as gs
    e = _engine()
    n, d, depth = 100_000, 64, 30
    runs = flips = reps = 0
    worst_div, worst_ratio = 0.0, 0.0
    for gain, act in ((0.9, 'tanh'), (1.0, 'tanh'), (1.1, 'tanh'), (0.9, 'selu'), (1.0, 'selu')):
        for seed in range(4):
            graph, mst, mou, s0 = gs.setup(n, 50 + seed, gain, act, d)
            states = [np.ones_like(s0), s0]
            for b in range(1, depth + 1):
                states.append(gs.run(graph, mst, mou, d, b, 0.0, s0, 1)[1])
            r = [float(np.max(gs.ratios(states[b + 1], states[b]))) for b in range(depth + 1)]
            del states
            cands, run_min = [], min(r[:15])
            for b in range(15, depth):
                if r[b] < run_min:
                    if r[b] * 1.02 < 0.99 * run_min: cands.append((b, r[b] * 1.02))
                    if r[b] < 0.9 * run_min: cands.append((b, float(np.sqrt(r[b] * run_min))))
                    run_min = r[b]
            for b, thr in cands[:4]:
                k1, s1, _ = gs.run(graph, mst, mou, d, depth, thr, s0, 1)
                k2, s2, rep = gs.run(graph, mst, mou, d, depth, thr, s0, 2)
                assert k1 == b, (gain, act, seed, b, thr, k1, r[max(0, b - 2):b + 1])      # the threshold does what it was picked for
                if rep:
                    assert np.array_equal(s2, s1)                      # a repeated Loop returns the exact path's bits
                    s2 = gs.run(graph, mst, mou, d, k1, 0.0, s0, 2)[1]
                div = float(np.max(np.abs(s2 - s1))) / float(np.max(np.abs(s1)))
                band = 1e-5 + 1e-3 * thr
                runs += 1; reps += rep; flips += (k1 != k2)
                worst_div = max(worst_div, div); worst_ratio = max(worst_ratio, div / band)
                assert k2 == k1, (gain, act, seed, b, thr, k1, k2, rep, div)
            graph.close()
    print(f'deep stops: {runs} runs, {flips} flips, {reps} repeats, largest relative divergence {worst_div:.3e} = {worst_ratio:.2f} of the band')
    assert runs >= 40 and flips == 0
    assert reps <= runs // 10
    assert worst_ratio < 0.5


def test_zero_state_nodes_are_not_borderline():
    """A node whose state is exactly zero and stays zero (here: a linear net with zero weights and biases from a zero initial state) has distance 0
    and norm 0: `0 > threshold * 0` is false under every arithmetic - it must not count as borderline, or every such Loop would be run twice
    (ADVICE r4)."""
    import gate_study as gs
    e = _engine()
    from GNN import GNN_utils as utils
    n, d = 4096, 64
    s = utils.syntheticGraph(n, 10.0, 3, 1, 2, seed=3)
    rng = np.random.default_rng(3)
    st = gs.make_net(rng, 1 + 2 * (3 + d), [128, 128, d], 'relu', 1.0)
    st['weights'] = [np.zeros_like(w) if i < 6 else w for i, w in enumerate(st['weights'])]      # W = 0, b = 0; BatchNormalization defaults
    ou = gs.make_net(rng, 3 + d, [2], 'softmax', 1.0)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    for form in (1, 2):
        lp = e.Loop(graph, mst, mou, d, 10, 0.01)
        assert lp.set_impl(2) == 2
        lp.set_tile_form(form)
        lp.set_persistent(False)
        lp.set_state0(np.zeros((n, d), np.float32))
        k = lp.run()
        repeated, _ = lp.gate_info()
        assert k == 1 and not repeated, (form, k, repeated)       # first condition: |0 - 1| > thr * |1| runs body 0; its gate: nothing moves
        assert not lp.state().any()
        lp.close()
    graph.close()
