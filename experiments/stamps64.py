"""Summarise GNN_FUSED_STAMPS output of the 64-node-tile kernel (k_fused64, diagnostic build): per-tile phase durations in shader cycles."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[(a[:, 0] > 0) & (a[:, 7] > 0)]
names = ['layer 0', 'layer 1 (+ gather events)', 'layer 2 (+ gather events)', 'gather clean-up', 'requests (ticket, old rows, next own rows)', 'epilogue + stores', 'next labels -> image']
d = np.diff(a, axis=1)
print('tiles', len(a), 'per-tile total median', np.median(a[:, 7] - a[:, 0]))
for i, n in enumerate(names):
    print(f'{n:44s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}')
t0 = a[:, 0].min()
print('first tile start .. last tile end (cycles):', (a[:, 7] - t0).max())
