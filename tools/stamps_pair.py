"""Summarise GNN_FUSED_STAMPS output of the wave-pair kernel (k_fused_pair, diagnostic build, gnn_loop_set_tile_form 2): per-tile phase
durations of side 0 (s_memtime ticks = shader cycles)."""
import sys
import numpy as np
a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 16).astype(np.int64)
a = a[a[:, 0] > 0]
names = ['gather + labels', 'layer-0 operand cut', 'meeting 1 (operand complete)', 'layer 0', 'activation + cut', 'meetings 2 + 3 + piece stores', 'layer 1',
         'activation + cut', 'meetings 4 + 5 + piece stores', 'last layer', 'epilogue -> LDS', 'meeting 6', 'condition + row stores', 'meeting 7 (+ ticket)']
b = a[:, :15]
d = np.diff(b, axis=1)
print('tiles', len(a), 'per-tile total median', np.median(b[:, -1] - b[:, 0]))
for i, n in enumerate(names):
    print(f'{n:32s} median {np.median(d[:, i]):9.0f}  p10 {np.percentile(d[:, i], 10):9.0f}  p90 {np.percentile(d[:, i], 90):9.0f}')
print('sum of the meeting waits (median of per-tile sums):', np.median(d[:, 2] + d[:, 11] + d[:, 13]), '+ inside the exchange steps')
