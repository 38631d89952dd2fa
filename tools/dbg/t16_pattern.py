"""debug: one body, impl 2 (16-node-tile kernel for small grids) vs impl 1: where do they differ?"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e, GNN_utils as utils
from util import make_mlp
np.set_printoptions(linewidth=250, precision=3, suppress=True)
n, d = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 64
act = sys.argv[2] if len(sys.argv) > 2 else 'selu'
s = utils.syntheticGraph(n, 6.0, 3, 1, 2, seed=3)
n = s['n_nodes']
rng = np.random.default_rng(0)
st = make_mlp(rng, 135, [128, 128, 64], act, gain=0.6, bn_random=True)
ou = make_mlp(rng, 67, [2], 'softmax')
s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
lo, hi = (int(x) for x in (sys.argv[3].split(':') if len(sys.argv) > 3 else '0:135'.split(':')))
keep = np.zeros(135, bool); keep[lo:hi] = True
st['weights'][0] = st['weights'][0] * keep[:, None].astype(np.float32)
print('== layer-0 rows kept', lo, hi)
graph = e.Graph(n, s['indptr'], s['adj_src'], s['adj_w'], s['arc_w'], s['arc_labels_csr'], s['nodes'], np.ones(n, np.uint8))
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], True)
res = {}
for impl in (1, 2):
    lp = e.Loop(graph, mst, mou, d, 1, 0.0)
    lp.set_impl(impl); lp.set_state0(s0); lp.run()
    res[impl] = lp.state()
diff = np.abs(res[2] - res[1])
print('max diff', diff.max(), 'mean |s|', np.abs(res[1]).mean())
print('per node max diff (first 48):', diff.max(1)[:48])
print('per feature max diff:', diff.max(0))
bad = np.argwhere(diff > 1e-4)
print('bad count', len(bad), 'of', diff.size)
print('node 0 impl1', res[1][0, :16]); print('node 0 impl2', res[2][0, :16])
