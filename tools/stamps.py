"""Summarise GNN_FUSED_STAMPS output: per-wave phase durations (s_memtime ticks = shader cycles)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 0] > 0]
names = ['A0-2 load own/labels', 'A3 gather', 'layer0 MFMA', 'epilogue0+swap', 'layers 1..', 'last epilogue -> LDS', 'norm + store']
d = np.diff(a, axis=1)
print('waves', len(a), 'total median', np.median(a[:, 7] - a[:, 0]))
for i, n in enumerate(names):
    print(f'{n:28s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}')
t0 = a[:, 0].min()
print('kernel span (cycles)', a[:, 7].max() - t0)
