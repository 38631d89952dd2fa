"""Certified gate of the default path (impl 2; csrc/gnn_common.h, reference GNN/GNN.py:202-220) studied where it could break or cost:

  A. non-contractive / slowly converging state maps that STOP at k in [15, 30]: gains 0.9 / 1.0 / 1.1, thresholds taken
     from the exact chain's own per-body ratio sequence so that the loop stops deep; k of impl 2 against k of impl 1 (bit-identical to the C
     oracle, asserted by the parity tests), whether the Loop was repeated, and the divergence |impl 2 - impl 1| of the stopping state
     next to the band;
  B. how often a CONVERGING run at threshold 0.01 is repeated on impl 1: contractive maps, mid and full size.

GPU box only.  python tools/gate_study.py [A|B|AB] [n_seeds] > gpurun_out/gate_study.txt
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

from GNN import _engine as e, GNN_utils as utils      # noqa: E402


def make_net(rng, n_in, widths, act, gain, out_act=None):
    w, acts = [], []
    for i, u in enumerate(widths):
        w += [(gain * rng.standard_normal((n_in, u)) / np.sqrt(n_in)).astype(np.float32), (0.1 * rng.standard_normal(u) / np.sqrt(u)).astype(np.float32)]
        acts.append(out_act if (out_act and i == len(widths) - 1) else act)
        n_in = u
    w += [np.ones(n_in, np.float32), np.zeros(n_in, np.float32), np.zeros(n_in, np.float32), np.ones(n_in, np.float32)]
    return dict(weights=w, activations=acts, batch_normalization=True)


def setup(n, seed, gain, act, d=64, hidden=(128, 128)):
    s = utils.syntheticGraph(n, 10.0, 3, 1, 2, seed=1000 + seed)
    rng = np.random.default_rng(seed)
    st = make_net(rng, 1 + 2 * (3 + d), list(hidden) + [d], act, gain)
    ou = make_net(rng, 3 + d, [2], 'softmax', 1.0)
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    return graph, mst, mou, s0


def run(graph, mst, mou, d, max_it, thr, s0, impl):
    lp = e.Loop(graph, mst, mou, d, max_it, thr)
    assert lp.set_impl(impl) == impl
    lp.set_persistent(False)
    lp.set_state0(s0)
    k = lp.run()
    st, rep = lp.state(), lp.gate_info()[0]
    lp.close()
    return int(k), st, bool(rep)


def ratios(s_new, s_old):
    """per-node distance / norm in float32 (the quantity condition() compares with the threshold; summation order differs from the kernel's
    in the last bits, which is irrelevant for picking thresholds)"""
    dist = np.sqrt(np.sum((s_new - s_old) ** 2, axis=1, dtype=np.float32))
    nrm = np.sqrt(np.sum(s_old ** 2, axis=1, dtype=np.float32))
    with np.errstate(divide='ignore', invalid='ignore'):
        return np.where(nrm > 0, dist / nrm, np.inf)


def study_a(n_seeds, n=100_000, d=64, depth=30):
    print(f'# A. deep stops on slowly converging / non-contractive maps: N = {n}, d = {d}, 135->128->128->64, {n_seeds} seeds per (gain, activation)')
    print('# gain act seed | stop body, threshold (x = just above that body\'s max ratio, g = between it and the previous minimum) | k1 k2 repeated | '
          'max|s2-s1| / max|s1| at the stop | band / norm = 1e-5 + 1e-3 thr')
    flips = runs = reps = 0
    worst = 0.0
    for gain, act in ((0.9, 'tanh'), (1.0, 'tanh'), (1.1, 'tanh'), (0.9, 'selu'), (1.0, 'selu'), (1.1, 'selu')):
        for seed in range(n_seeds):
            graph, mst, mou, s0 = setup(n, seed, gain, act, d)
            states = [np.ones_like(s0), s0]                     # condition() first compares the initial state with ones
            for b in range(1, depth + 1):
                states.append(run(graph, mst, mou, d, b, 0.0, s0, 1)[1])
            # r[b] = max ratio of the gate BEHIND body b (b = 0: the first condition); the loop with threshold t stops at the first b with r[b] <= t
            r = [float(np.max(ratios(states[b + 1], states[b]))) for b in range(depth + 1)]
            cands = []
            run_min = min(r[:15])
            for b in range(15, depth):
                if r[b] < run_min:
                    if r[b] * 1.02 < 0.99 * run_min: cands.append((b, 'x', r[b] * 1.02))
                    if r[b] < 0.9 * run_min: cands.append((b, 'g', float(np.sqrt(r[b] * run_min))))
                    run_min = r[b]
            if not cands:
                print(f'{gain} {act} {seed} | no new minimum of the max ratio in bodies 15..{depth - 1} (min of the first 15: {min(r[:15]):.3g}, last: {r[-1]:.3g})')
            for b, kind, thr in cands[:6]:
                k1, s1, _ = run(graph, mst, mou, d, depth, thr, s0, 1)
                k2, s2, rep = run(graph, mst, mou, d, depth, thr, s0, 2)
                # divergence of the two arithmetics at the stop (after a repeat the states ARE impl 1's: measure on a threshold-0 run of k1 bodies)
                if rep:
                    s2 = run(graph, mst, mou, d, max(k1, 1), 0.0, s0, 2)[1]
                div = float(np.max(np.abs(s2 - s1))) / max(1e-30, float(np.max(np.abs(s1))))
                runs += 1; reps += rep; flips += (k1 != k2); worst = max(worst, div)
                print(f'{gain} {act} {seed} | body {b} {kind} thr {thr:.6g} | {k1} {k2} {int(rep)} | {div:.3e} | {1e-5 + 1e-3 * thr:.3e}' + ('   <-- K FLIPPED' if k1 != k2 else ''), flush=True)
            graph.close()
    print(f'# A: {runs} runs, {flips} with k2 != k1, {reps} repeated on impl 1, largest relative divergence at a stop {worst:.3e}')
    return flips


def study_b(n_seeds):
    print('# B. repeat rate of converging runs at threshold 0.01 (max_iter 50)')
    print('# N gain act | seeds | k (min..max) | repeated | ms per Loop: impl 2 (incl. repeats) / impl 1')
    for n, seeds in ((100_000, n_seeds), (1_000_000, max(3, n_seeds // 5))):
        for gain, act in ((0.5, 'tanh'), (0.7, 'tanh'), (0.5, 'selu'), (0.7, 'selu'), (0.8, 'selu')) if n < 1_000_000 else ((0.5, 'selu'), (0.7, 'selu')):
            ks, rep, t2, t1 = [], 0, 0.0, 0.0
            for seed in range(seeds):
                graph, mst, mou, s0 = setup(n, 100 + seed, gain, act)
                for impl in (2, 1):
                    lp = e.Loop(graph, mst, mou, 64, 50, 0.01)
                    lp.set_impl(impl); lp.set_persistent(False); lp.set_state0(s0)
                    lp.run()
                    t = time.perf_counter()
                    k = lp.run()
                    dt = time.perf_counter() - t
                    if impl == 2:
                        ks.append(int(k)); rep += int(lp.gate_info()[0]); t2 += dt
                    else:
                        assert int(k) == ks[-1], (k, ks[-1])
                        t1 += dt
                    lp.close()
                graph.close()
            print(f'{n} {gain} {act} | {seeds} | {min(ks)}..{max(ks)} | {rep} | {1e3 * t2 / seeds:.2f} / {1e3 * t1 / seeds:.2f}', flush=True)


if __name__ == '__main__':
    what = sys.argv[1] if len(sys.argv) > 1 else 'AB'
    n_seeds = int(sys.argv[2]) if len(sys.argv) > 2 else 20
    e.require_device(0)
    bad = 0
    if 'A' in what:
        bad = study_a(n_seeds)
    if 'B' in what:
        study_b(n_seeds)
    sys.exit(1 if bad else 0)
