"""k_fwd3_split / k_bwd3_split on hidden widths that are not 128: one training step with the fused passes on and off (diagnostic build:
GNN_TRAIN_FWD3 / GNN_TRAIN_BWD3 are read once per process, so each setting is its own process).  python tools/dbg/fwd3_widths.py H1 H2 [n]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN import _engine as e
from oracle import gnn_oracle as orc
from util import make_mlp, random_arcs

h1, h2 = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 4200
rng = np.random.default_rng(n)
d, nl, al, max_it = 64, 3, 1, 3
arcs = random_arcs(rng, n, 4 * n, al)
nodes = (2 * rng.random((n, nl)) - 1).astype(np.float32)
g = orc.make_graph_dict(arcs, nodes, 'average')
st = make_mlp(rng, al + 2 * (d + nl), [h1, h2, d], 'selu', gain=0.7, bn_random=True)
ou = make_mlp(rng, d + nl, [2], 'softmax', batch_normalization=False)
mask = np.ones(n, bool)
targets = np.eye(2)[rng.integers(0, 2, n)].astype(np.float32)
weights = (rng.uniform(0.5, 1.5, n) / n).astype(np.float32)
s0 = (0.1 * rng.standard_normal((n, d))).astype(np.float32)
graph = e.Graph(n, g['adjT'][0], g['adjT'][1], g['adjT'][2], g['arcT'][2], np.asarray(g['arcs'])[:, 2:][g['arcT'][1]], nodes, mask)
mst, mou = e.Mlp(st['weights'], st['activations'], True), e.Mlp(ou['weights'], ou['activations'], False)
loop = e.Loop(graph, mst, mou, d, max_it, 0.0)
loop.set_state0(s0)
r = loop.train_step(mst, mou, None, targets, weights, 0, None, dropout_state=[0, 0, 0, 0], dropout_output=[0, 0], bn_state=np.concatenate(st['weights'][-4:-2]), bn_output=None)
out = os.environ.get('OUT')
np.savez(out, loss=r['loss'], **{f'gs{i}': a for i, a in enumerate(r['grads_state'])})
if os.environ.get('ORACLE'):
    from oracle import gnn_train_oracle as tro
    g['set_mask'] = mask
    ref = tro.train_step(g, st, ou, d, max_it, 0.0, s0, [{} for _ in range(max_it)], {}, targets, weights, loss='categorical_crossentropy', mean=False, graph_based=False)
    np.savez(os.environ['ORACLE'], loss=ref['loss'], **{f'gs{i}': a for i, a in enumerate(ref['grads_state'])})
print(os.environ.get('GNN_TRAIN_FWD3'), os.environ.get('GNN_TRAIN_BWD3'), 'loss', r['loss'])
