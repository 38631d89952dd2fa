"""MI355X-native engine for the GNN / LGNN fixed-point state-propagation loop.

Import paths mirror the reference package ``GNN`` (``GNN.GNN``, ``GNN.LGNN``, ``GNN.MLP``, ``GNN.graph_class``,
``GNN.GNN_utils``) so that a starter script switches over by putting ``gnn_tf_2.x_amd`` first on ``sys.path``.
All arithmetic of the hot path runs in ``libgnn_hip.so`` (HIP, gfx950) through the C ABI of ``include/gnn_hip.h``.
"""
__version__ = '0.1.0'
