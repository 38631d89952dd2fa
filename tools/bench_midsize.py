"""Per-iteration time of the fused kernel across graph sizes (latency-bound to throughput-bound): the bench workload's generator,
weights and arithmetic at N nodes.  Prints ms per iteration (HIP events around every launch), the algorithmic TB/s and the fraction of
the 8 TB/s roof.  GNN_TILE_FORM=1 / 2 forces one wave per tile (k_fused) / a wave pair per tile (k_fused_pair) on the default path; 0 (default) is the
library's choice.     python tools/bench_midsize.py [N ...]"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd')):
    sys.path.insert(0, p)
from GNN import _engine as e, GNN_utils as utils      # noqa: E402
import bench                                           # noqa: E402


def main():
    sizes = [int(x) for x in sys.argv[1:]] or [31_250, 62_500, 125_000, 250_000, 500_000, 1_000_000]
    d, nl, al, t = 64, 3, 1, 2
    rng = np.random.default_rng(20261003)
    st = bench.make_net(rng, al + 2 * (nl + d), [128, 128, d], 'selu')
    ou = bench.make_net(rng, nl + d, [t], 'softmax')
    mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
    print(f'# GNN_TILE_FORM={os.environ.get("GNN_TILE_FORM", "0")} library={os.path.basename(e.LIB_PATH)}')
    for n in sizes:
        s = utils.syntheticGraph(n, 10.0, nl, al, t, seed=20261003)
        n = s['n_nodes']
        s0 = (0.1 * np.random.default_rng(1).standard_normal((n, d))).astype(np.float32)
        graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
        res = {}
        form = int(os.environ.get('GNN_TILE_FORM', '0'))      # default path: 1 one wave per tile, 2 wave pair per tile, 0 library's choice
        for impl in (2, 1):
            loop = e.Loop(graph, mst, mou, d, 30, 0.0)
            loop.set_impl(impl)
            if impl == 2: loop.set_tile_form(form)
            loop.set_state0(s0)
            loop.run()
            loop.set_profiling(True)
            ms = []
            for _ in range(3):
                loop.run()
                ms.append(loop.timing()['avg_iter_ms'])
            res[impl] = (float(np.median(ms)), loop.state())
            loop.close()
        by = bench.algorithmic_bytes_per_iteration(n, s['n_arcs'], d, nl, al)
        ms2, ms1 = res[2][0], res[1][0]
        diff = float(np.max(np.abs(res[2][1] - res[1][1])))
        print(f'N={n:8d} E={s["n_arcs"]:9d}  impl 2: {ms2:.4f} ms/iteration = {by / ms2 / 1e9:6.2f} TB/s = {by / ms2 / 1e9 / 8:.3f} of roof   '
              f'impl 1: {ms1:.4f} ms   max |impl2 - impl1| after 30 bodies {diff:.2e}', flush=True)
        graph.close()


if __name__ == '__main__':
    main()
