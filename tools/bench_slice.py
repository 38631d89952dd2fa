"""Per-rank device time of one iteration of BASELINE configs[3] (1 M nodes sharded over 8 ranks) for the three exchange layouts,
measured on ONE GPU: the 8 ranks of a loopback group run one after the other on the same device, so every kernel's duration is
that of one rank working alone (the inter-GPU transfers are device copies here and are NOT what xGMI would take).
Run on the GPU box under rocprofv3 --kernel-trace and summarise with tools/summarize_db.py, or alone for the wall time."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e, GNN_utils as utils
from util import make_mlp

world = int(os.environ.get('WORLD', 8))
layout = os.environ.get('LAYOUT', 'slice')
n, d, iters = int(os.environ.get('NODES', 1_000_000)), 64, 6
s = utils.syntheticGraph(n, 10, 3, 1, 2, seed=3)
rng = np.random.default_rng(0)
st = make_mlp(rng, 1 + 2 * (3 + d), [128, 128, d], 'selu', gain=0.6)
ou = make_mlp(rng, 3 + d, [2], 'softmax')
state0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
comms = e.Comm.loopback(world)
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
plan = e.halo_plan(n, world, s['indptr'], s['adj_src']) if layout == 'halo' else None
loops, graphs = [], []
for r in range(world):
    rb, nr, ip, src, w, aw, al_ = e.shard_csr(n, r, world, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'])
    if layout == 'halo':
        h = e.shard_halo(n, r, world, s['indptr'], s['adj_src'], s['nodes'], plan)
        gr = e.Graph.halo(n, r, world, h['block'], h['send_rows'], ip, h['adj_src'], w, aw, al_, h['nodes'], np.ones(nr, np.uint8))
    else:
        gr = e.Graph(n, ip, src, w, aw, al_, s['nodes'], np.ones(nr, np.uint8), row_begin=rb)
    lp = e.Loop(gr, mst, mou, d, iters, 0.0, comms[r])
    lp.set_impl(2)
    lp.set_tile_form(int(os.environ.get('GNN_TILE_FORM', 0)))       # 1 / 2: force one wave / a wave pair per tile on the owned rows
    lp.set_state0(state0[rb:rb + nr])
    if layout == 'slice':
        gr.set_full_adjacency(n, s['indptr'], s['adj_src'], s['adj_w'])
        lp.set_slice_exchange(True, form='pipelined' if int(os.environ.get('SLICE_FORM', 1)) == 1 else 'oneshot')      # 1: return all-to-all block by block beside the aggregation; 2: whole slice, then all-to-all
    loops.append(lp); graphs.append(gr)
e.Loop.run_group(loops)
t = time.perf_counter()
k = e.Loop.run_group(loops)
dt = time.perf_counter() - t
print(f'layout={layout} world={world} n={n}: {iters} iterations of all {world} ranks in {1e3 * dt:.2f} ms '
      f'= {1e3 * dt / iters / world:.3f} ms per rank and iteration (device copies in place of xGMI)')
