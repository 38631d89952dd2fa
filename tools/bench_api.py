"""Cost of the Python mirror on top of the engine: GNNgraphBased.Loop / evaluate / one training epoch on MUTAG batches of 32 through the
reference's object API (GraphObject -> GraphTensor -> Loop), beside the engine-level figures of tools/bench_small.py.  Run on the GPU box."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, 'gnn_tf_2.x_amd'), os.path.join(ROOT, 'tests')):
    sys.path.insert(0, p)
from GNN.graph_class import GraphObject, GraphTensor
from GNN.GNN import GNNgraphBased
from GNN.MLP import MLP, get_inout_dims
from GNN import optimizers, losses
import load_MUTAG

graphs = load_MUTAG.load(limit=320)
batches = [GraphObject.merge(graphs[i:i + 32], problem_based='g', aggregation_mode='average') for i in range(0, 320, 32)]
tens = [GraphTensor.fromGraphObject(b) for b in batches]
g0 = batches[0]
ins, ls = get_inout_dims('state', g0.DIM_NODE_LABEL, g0.DIM_ARC_LABEL, g0.DIM_TARGET, 'g', 0, [32, 32])
ino, lo = get_inout_dims('output', g0.DIM_NODE_LABEL, g0.DIM_ARC_LABEL, g0.DIM_TARGET, 'g', 0, [])
net_state = MLP(input_dim=ins, layers=ls, activations=['selu'] * len(ls), kernel_initializer='lecun_normal', bias_initializer='lecun_normal')
net_output = MLP(input_dim=ino, layers=lo, activations=['softmax'], kernel_initializer='glorot_normal', bias_initializer='glorot_normal', batch_normalization=False)
gnn = GNNgraphBased(net_state, net_output, optimizers.Adam(1e-3), losses.categorical_crossentropy, {}, 0, 50, 0.01, 'c', path_writer='/tmp/gnn_bench_api/', namespace='bench')
for t in tens: gnn.Loop(t)
reps = 20
t0 = time.perf_counter()
for _ in range(reps):
    for t in tens: gnn.Loop(t)
dt = time.perf_counter() - t0
print(f'GNNgraphBased.Loop(GraphTensor), MUTAG batches of 32: {reps * len(tens) / dt:9.1f} per second ({1e6 * dt / (reps * len(tens)):.0f} us each)')
t0 = time.perf_counter()
for _ in range(5): gnn.evaluate(tens)
dt = time.perf_counter() - t0
print(f'evaluate(10 batches): {1e3 * dt / 5:.2f} ms')
t0 = time.perf_counter()
gnn.train(tens, 5, None, update_freq=100, max_fails=100, verbose=0)
dt = time.perf_counter() - t0
print(f'train: {1e3 * dt / (5 * len(tens)):.2f} ms per batch step ({5 * len(tens)} steps)')
if os.environ.get('PROFILE'):
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        for t in tens: gnn.Loop(t)
    pr.disable()
    print('--- profile of GNNgraphBased.Loop x 100'); pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
    pr = cProfile.Profile(); pr.enable()
    gnn.train(tens, 3, None, update_freq=100, max_fails=100, verbose=0)
    pr.disable()
    print('--- profile of train, 30 steps'); pstats.Stats(pr).sort_stats('cumulative').print_stats(35)
