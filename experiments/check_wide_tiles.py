"""The round-4 experiment k_fused64 (csrc/experiments/gnn_fused64_kernel.h: 64-node tiles on one wave per SIMD, the next tile's gather inside the
current tile's matrix phase through an LDS-DMA ring) against k_fused, with the DIAGNOSTIC library (make -C gnn_tf_2.x_amd/csrc DIAG=1):
states, outputs and k of a set of shapes must be BIT-IDENTICAL between GNN_FUSED_WIDE=0 and =1 (same products, same order per accumulator),
and the per-launch time of both on the BASELINE workload is printed.  Results of round 4: profiles/r04_fused64_stamps.txt.
Usage: python tools/check_wide_tiles.py            (spawns one worker process per setting: the switch is read once per process)"""
import os, subprocess, sys, tempfile
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')]

SHAPES = [(4096, 3, 1, (128, 128), 'selu', 10), (4096, 5, 1, (128, 128), 'selu', 10), (1024, 3, 2, (128,), 'tanh', 4), (2048, 3, 1, (96, 128), 'tanh', 7),
          (640, 6, 2, (128, 128), 'selu', 30), (64, 3, 1, (128, 128), 'selu', 3)]


def worker(out_path):
    import bench
    from GNN import _engine as e, GNN_utils as utils
    from oracle import gnn_oracle as orc
    from util import make_mlp, random_arcs
    res = {}
    d = 64
    for idx, (n, nl, al, hidden, act, deg) in enumerate(SHAPES):
        rng = np.random.default_rng(900 + n + nl)
        arcs = random_arcs(rng, n, deg * n, al, sort=True)
        nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
        g = orc.make_graph_dict(arcs, nodes, 'average')
        st = make_mlp(rng, al + 2 * (nl + d), list(hidden) + [d], act, gain=0.6, bn_random=True)
        ou = make_mlp(rng, nl + d, [2], 'softmax', bn_random=True)
        s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
        arc_labels = np.asarray(g['arcs'], np.float32)[:, 2:]
        graph = e.Graph(n, g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], arc_labels[g['arcT'][1]], g['nodes'], np.ones(n, np.uint8))
        loop = e.Loop(graph, e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 12, 0.01)
        loop.set_state0(s0)
        res[f'k{idx}'] = np.float32(loop.run())
        res[f's{idx}'], res[f'o{idx}'] = loop.state(), loop.output()
        loop.run()
        assert np.array_equal(loop.state(), res[f's{idx}'])          # a second run on the same handle
        loop.close(); graph.close()
    # BASELINE workload: per-launch time
    n, nl, al, t = 1_000_000, 3, 1, 2
    s = utils.syntheticGraph(n, 10.0, nl, al, t, seed=20261003)
    rng = np.random.default_rng(20261003)
    st = bench.make_net(rng, al + 2 * (nl + d), [128, 128, d], 'selu')
    ou = bench.make_net(rng, nl + d, [t], 'softmax')
    s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
    graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
    loop = e.Loop(graph, e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True), d, 30, 0.0)
    loop.set_state0(s0)
    loop.run()
    loop.set_profiling(True)
    ms = []
    for _ in range(3):
        loop.run()
        ms.append(loop.timing()['avg_iter_ms'])
    res['ms'] = np.float32(min(ms))
    res['big'] = loop.state()[::997].copy()
    np.savez(out_path, **res)


def main():
    diag = os.path.join(ROOT, 'gnn_tf_2.x_amd', 'GNN', 'libgnn_hip_diag.so')
    if not os.path.exists(diag):
        raise SystemExit('build the diagnostic library first: make -C gnn_tf_2.x_amd/csrc DIAG=1')
    outs = []
    with tempfile.TemporaryDirectory() as tmp:
        for wide in (0, 1):
            path = os.path.join(tmp, f'w{wide}.npz')
            env = dict(os.environ, GNN_HIP_LIBRARY=diag, GNN_FUSED_WIDE=str(wide))
            subprocess.check_call([sys.executable, os.path.abspath(__file__), '--worker', path], env=env)
            outs.append(dict(np.load(path)))
    a, b = outs
    bad = [k for k in a if k != 'ms' and not np.array_equal(a[k], b[k])]
    print(f'{len(SHAPES)} shapes + the BASELINE graph: ' + ('k, states and outputs BIT-IDENTICAL between k_fused and k_fused64' if not bad else f'DIFFERENCES in {bad}'))
    print(f'per launch at BASELINE size: k_fused {float(a["ms"]):.4f} ms, k_fused64 {float(b["ms"]):.4f} ms')
    raise SystemExit(1 if bad else 0)


if __name__ == '__main__':
    if len(sys.argv) == 3 and sys.argv[1] == '--worker':
        worker(sys.argv[2])
    else:
        main()
