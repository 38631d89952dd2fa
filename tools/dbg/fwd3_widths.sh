#!/bin/bash
export GNN_HIP_LIBRARY=$PWD/gnn_tf_2.x_amd/GNN/libgnn_hip_diag.so
for w in "100 96" "128 128" "104 96" "96 96" "100 128"; do
  set -- $w
  GNN_TRAIN_FWD3=0 GNN_TRAIN_BWD3=0 OUT=/tmp/a.npz ORACLE=/tmp/o.npz python3 tools/dbg/fwd3_widths.py $1 $2 > /dev/null
  GNN_TRAIN_FWD3=1 GNN_TRAIN_BWD3=0 OUT=/tmp/b.npz python3 tools/dbg/fwd3_widths.py $1 $2 > /dev/null
  python3 - $1 $2 <<'PY'
import numpy as np, sys
a, b, o = (np.load(f'/tmp/{x}.npz') for x in 'abo')
print(f'hidden {sys.argv[1]} {sys.argv[2]}: per array, max |error| vs the float64 oracle / max |entry|:   per-layer forward | fused forward (k_fwd3_split)')
for k in a.files:
    if k == 'loss': continue
    m = float(np.max(np.abs(o[k])))
    print(f'   {k} {str(o[k].shape):12s} {float(np.max(np.abs(a[k] - o[k]))) / m:.2e} | {float(np.max(np.abs(b[k] - o[k]))) / m:.2e}')
PY
done
