"""debug: contribution of a block of layer-0 input rows, impl 2 (16-node tiles) vs impl 1, linear net"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e, GNN_utils as utils
from util import make_mlp
np.set_printoptions(linewidth=250, precision=4, suppress=True)
n, d = 64, 64
s = utils.syntheticGraph(n, 6.0, 3, 1, 2, seed=3)
n = s['n_nodes']
rng = np.random.default_rng(0)
st0 = make_mlp(rng, 135, [128, 128, 64], 'linear', gain=0.6)
ou = make_mlp(rng, 67, [2], 'softmax')
s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
mou = e.Mlp(ou['weights'], ou['activations'], True)
W0 = st0['weights'][0].copy()
def run(lo, hi):
    keep = np.zeros(135, bool); keep[lo:hi] = True
    w = [x.copy() for x in st0['weights']]
    w[0] = W0 * keep[:, None].astype(np.float32)
    mst = e.Mlp(w, st0['activations'], True)
    out = {}
    for impl in (1, 2):
        lp = e.Loop(graph, mst, mou, d, 1, 0.0)
        lp.set_impl(impl); lp.set_state0(s0); lp.run()
        out[impl] = lp.state().astype(np.float64)
    return out
base = run(0, 0)
print('baseline diff', np.abs(base[1] - base[2]).max())
for lo, hi in [(0, 8), (8, 16), (24, 32), (32, 40), (40, 48), (56, 64), (64, 67), (68 - 1, 75), (131, 135)]:
    r = run(lo, hi)
    d1, d2 = r[1] - base[1], r[2] - base[2]
    ratio = float((d1 * d2).sum() / max((d1 * d1).sum(), 1e-30))
    print(f'rows {lo}:{hi}  |d1| {np.abs(d1).max():.4f}  |d2| {np.abs(d2).max():.4f}  projection d2 on d1 {ratio:.4f}  max|d2 - d1| {np.abs(d2 - d1).max():.2e}')
