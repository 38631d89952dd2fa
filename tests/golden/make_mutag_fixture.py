"""Pack the reference's MUTAG_raw/*.txt (TU 'Mutagenicity' dataset: DATA files, not code) into one small .npz so that the
MUTAG configuration (BASELINE.json configs[1]) can be exercised where /root/reference does not exist (the GPU box).

    python tests/golden/make_mutag_fixture.py
"""
import os

import numpy as np

SRC = '/root/reference/MUTAG_raw/'
OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mutag_raw.npz')

edges = np.loadtxt(SRC + 'Mutagenicity_edges.txt', dtype=np.int32, delimiter=',')
out = dict(edges=edges,
           edge_labels=np.loadtxt(SRC + 'Mutagenicity_edge_labels.txt', dtype=np.int8),
           node_labels=np.loadtxt(SRC + 'Mutagenicity_node_labels.txt', dtype=np.int8),
           graph_indicator=np.loadtxt(SRC + 'Mutagenicity_graph_indicator.txt', dtype=np.int32),
           graph_labels=np.loadtxt(SRC + 'Mutagenicity_graph_labels.txt', dtype=np.int8))
np.savez_compressed(OUT, **out)
print({k: v.shape for k, v in out.items()}, os.path.getsize(OUT), 'bytes')
