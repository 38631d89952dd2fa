"""Latency of the launch-bound configurations (BASELINE configs[0] / [1]): starter random graphs and MUTAG batches of 32.
Prints graphs/s and node-updates/s for the fused and per-op paths.  Run on the GPU box: python tools/bench_small.py"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e, GNN_utils as utils          # noqa: E402
from GNN.graph_class import GraphObject, GraphTensor      # noqa: E402
from util import make_mlp                                 # noqa: E402
import load_MUTAG                                         # noqa: E402


def run(name, batches, st, ou, d, max_it, thr, graph_based):
    rng = np.random.default_rng(0)
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    for impl in (1, 0):
        loops = []
        for b in batches:
            gt = GraphTensor.fromGraphObject(b)
            loop = e.Loop(gt.device_graph(), mst, mou, d, max_it, thr)
            loop.set_impl(impl)
            if d:
                loop.set_state0((0.1 * rng.standard_normal((b.nodes.shape[0], d))).astype(np.float32))
            loops.append((loop, b))
        for loop, _ in loops:
            loop.run()
        t = time.perf_counter()
        reps, updates, iters = 20, 0, 0
        for _ in range(reps):
            for loop, b in loops:
                k = loop.run()
                updates += k * b.nodes.shape[0]
                iters += k
        dt = time.perf_counter() - t
        n_graphs = sum(b.targets.shape[0] if graph_based else 1 for _, b in loops) * reps
        print(f'{name:8s} impl={impl} loops/s={reps * len(loops) / dt:9.1f}  graphs/s={n_graphs / dt:10.1f}  node-updates/s={updates / dt:12.3e}  '
              f'us/iteration={1e6 * dt / iters:7.2f}  mean k={iters / (reps * len(loops)):.1f}')


def main():
    rng = np.random.default_rng(1)
    graphs = load_MUTAG.load(limit=320)
    batches = [GraphObject.merge(graphs[i:i + 32], problem_based='g', aggregation_mode='average') for i in range(0, 320, 32)]
    run('MUTAG', batches, make_mlp(rng, 31, [32, 32, 14], 'selu', gain=0.7), make_mlp(rng, 14, [2], 'softmax'), 0, 50, 0.01, True)
    np.random.seed(3)
    rg = [utils.randomGraph(int(np.random.choice(range(15, 40))), 3, 1, 2, 0.7) for _ in range(96)]
    rb = utils.getbatches(rg, problem_based='n', aggregation_mode='average', batch_size=32)
    run('starter', rb, make_mlp(rng, 7, [3], 'selu', gain=0.7), make_mlp(rng, 3, [2], 'softmax'), 0, 5, 0.01, False)


if __name__ == '__main__':
    main()
