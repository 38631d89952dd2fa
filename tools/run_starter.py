"""End-to-end drop-in check (BASELINE configs[0]): import the TensorFlow-free starter (random graphs, node-focused binary
classification), train the GNN and the LGNN for a few epochs, test.  Run on the GPU box: python tools/run_starter.py"""
import os, sys, time
os.environ.setdefault('GNN_STARTER_OUTPUT_BN', '0')      # see the note in starter.py: the reference's default cannot learn
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, 'gnn_tf_2.x_amd'))
os.chdir(os.path.join(ROOT, 'gnn_tf_2.x_amd'))
import starter as s          # noqa: E402

t = time.perf_counter()
before = s.gnn.test(s.gTe)
s.gnn.train(s.gTr, 30, s.gVa, update_freq=10, max_fails=10, verbose=0)
after = s.gnn.test(s.gTe)
print(f'GNN : test loss {before["Loss"]:.3f} -> {after["Loss"]:.3f}, Acc {before["Acc"]:.3f} -> {after["Acc"]:.3f}, {time.perf_counter() - t:.1f} s for 30 epochs x {len(s.gTr)} batches')
t = time.perf_counter()
before = s.lgnn.test(s.gTe)
s.lgnn.train(s.gTr, 10, s.gVa, update_freq=5, max_fails=10, training_mode='parallel', verbose=0)
after = s.lgnn.test(s.gTe)
print(f'LGNN ({s.lgnn.LAYERS} layers, parallel): test loss {before["Loss"]:.3f} -> {after["Loss"]:.3f}, Acc {before["Acc"]:.3f} -> {after["Acc"]:.3f}, {time.perf_counter() - t:.1f} s for 10 epochs')
