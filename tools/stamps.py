"""Summarise GNN_FUSED_STAMPS output: per-tile phase durations (s_memtime ticks = shader cycles) and where the tiles sit in time."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8).astype(np.int64)
a = a[a[:, 0] > 0]
used = [0, 2, 3, 4, 5, 6, 7]                      # slot 1 is not stamped
names = ['tile load + gather', 'layer 0', 'epilogue 0 + swap', 'layers 1..', 'last epilogue -> LDS', 'norm + store']
b = a[:, used]
d = np.diff(b, axis=1)
print('tiles', len(a), 'per-tile total median', np.median(b[:, -1] - b[:, 0]))
for i, n in enumerate(names):
    print(f'{n:24s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}')
t0 = b[:, 0].min()
start, end = b[:, 0] - t0, b[:, -1] - t0
print('first tile start .. last tile end (cycles):', end.max())
print('tile START times: p1 %d  p10 %d  p50 %d  p90 %d  max %d' % tuple(np.percentile(start, [1, 10, 50, 90, 100])))
print('tile END times:   p1 %d  p10 %d  p50 %d  p90 %d  max %d' % tuple(np.percentile(end, [1, 10, 50, 90, 100])))
