"""Generate tests/golden/graph_fixtures.npz by running the REFERENCE's own NumPy/SciPy graph code.

Run in the build container only (needs /root/reference; the GPU box never has it):

    python tests/golden/make_fixtures.py

What is imported: ``GNN.graph_class.GraphObject`` and ``GNN.GNN_utils`` from /root/reference, unmodified.
Both do ``import tensorflow`` at module import although the functions used here are pure NumPy/SciPy/sklearn;
TensorFlow is not installed in this image, so the *name* ``tensorflow`` is satisfied by a placeholder created in a
temp dir (``tensorflow.keras.backend.floatx() -> 'float32'`` and an empty ``tensorflow.Tensor`` class used only as a
type annotation, graph_class.py:40,365).  No TensorFlow arithmetic is emulated: nothing downstream of a TF op is
captured here, which is why the TF half of the path stays "parity unpinned" (DESIGN.md).

The output holds only DATA (inputs and the matrices the reference built from them).
"""
import os
import sys
import tempfile

import numpy as np

REF = '/root/reference'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'graph_fixtures.npz')


def _placeholder_tensorflow(tmp):
    pkg = os.path.join(tmp, 'tensorflow', 'keras')
    os.makedirs(pkg)
    with open(os.path.join(tmp, 'tensorflow', '__init__.py'), 'w') as f:
        f.write('from . import keras\nclass Tensor: pass\n')
    with open(os.path.join(pkg, '__init__.py'), 'w') as f:
        f.write('from . import backend\n')
    with open(os.path.join(pkg, 'backend.py'), 'w') as f:
        f.write("def floatx(): return 'float32'\n")


def _dump(out, prefix, g):
    out[f'{prefix}/arcs'] = g.arcs
    out[f'{prefix}/nodes'] = g.nodes
    out[f'{prefix}/targets'] = g.targets
    out[f'{prefix}/set_mask'] = g.set_mask
    out[f'{prefix}/output_mask'] = g.output_mask
    out[f'{prefix}/sample_weights'] = g.sample_weights
    an, ad = g.ArcNode.tocoo(), g.Adjacency.tocoo()
    out[f'{prefix}/ArcNode_row'], out[f'{prefix}/ArcNode_col'], out[f'{prefix}/ArcNode_data'] = an.row, an.col, an.data
    out[f'{prefix}/Adj_row'], out[f'{prefix}/Adj_col'], out[f'{prefix}/Adj_data'] = ad.row, ad.col, ad.data
    out[f'{prefix}/Adj_dense'] = g.Adjacency.toarray()
    if g.NodeGraph is not None:
        out[f'{prefix}/NodeGraph'] = g.NodeGraph
    # the two loop-invariant aggregates the Loop derives from these matrices (GNN.py:259,263), via SciPy on the
    # reference-built matrices (float64 accumulation of float32 entries; a value check, not a TF output)
    out[f'{prefix}/AdjT_nodes'] = g.Adjacency.T.astype(np.float64).dot(g.nodes.astype(np.float64))
    out[f'{prefix}/ArcNodeT_arclabels'] = g.ArcNode.T.astype(np.float64).dot(g.arcs[:, 2:].astype(np.float64))


def main():
    sys.dont_write_bytecode = True
    with tempfile.TemporaryDirectory() as tmp:
        _placeholder_tensorflow(tmp)
        sys.path[:0] = [tmp, REF]
        from GNN.graph_class import GraphObject
        from GNN import GNN_utils as utils

        out = {}
        for mode in ['average', 'sum', 'normalized']:
            for pb in ['n', 'g']:
                _dump(out, f'simple/{mode}/{pb}', utils.simple_graph(pb, aggregation_mode=mode))

        # randomGraph with fixed NumPy seeds (GNN_utils.py:16-84)
        rnd = []
        for seed, n in [(1, 17), (2, 23), (3, 31), (4, 15), (5, 39)]:
            np.random.seed(seed)
            g = utils.randomGraph(nodes_number=n, dim_node_label=3, dim_arc_label=1, dim_target=2, density=0.7,
                                  aggregation_mode='average', problem_based='n')
            _dump(out, f'random/{seed}', g)
            rnd.append(g)

        # merge (graph_class.py:285-319): node-based, every aggregation mode
        for mode in ['average', 'sum', 'normalized']:
            _dump(out, f'merge_n/{mode}', GraphObject.merge(rnd[:3], problem_based='n', aggregation_mode=mode))

        # graph-based merge: block-diagonal NodeGraph
        gg = []
        for seed, n in [(11, 6), (12, 9), (13, 5)]:
            np.random.seed(seed)
            gg.append(utils.randomGraph(nodes_number=n, dim_node_label=2, dim_arc_label=2, dim_target=2, density=0.8,
                                        aggregation_mode='average', problem_based='g'))
        for i, g in enumerate(gg):
            _dump(out, f'gsingle/{i}', g)
        _dump(out, 'merge_g/average', GraphObject.merge(gg, problem_based='g', aggregation_mode='average'))

        # merge of two copies of simple_graph (SURVEY.md 8c KAT)
        s = utils.simple_graph('g')
        _dump(out, 'merge_simple2', GraphObject.merge([s, s.copy()], problem_based='g', aggregation_mode='average'))

        # getbatches (GNN_utils.py:177-195) on 70 small graphs: batch shapes only
        np.random.seed(20261003)
        many = [utils.randomGraph(int(np.random.choice(range(15, 40))), 3, 1, 2, 0.7) for _ in range(70)]
        batches = utils.getbatches(many, problem_based='n', aggregation_mode='average', batch_size=32)
        out['getbatches/shapes'] = np.array([[b.nodes.shape[0], b.arcs.shape[0]] for b in batches])
        out['getbatches/sizes'] = np.array([[g.nodes.shape[0], g.arcs.shape[0]] for g in many])

        np.savez_compressed(OUT, **out)
        print('wrote', OUT, os.path.getsize(OUT), 'bytes,', len(out), 'arrays')


if __name__ == '__main__':
    main()
