#!/bin/bash
export GNN_HIP_LIBRARY=$PWD/gnn_tf_2.x_amd/GNN/libgnn_hip_diag.so
for w in "100 96" "128 96" "100 128" "96 128" "128 128" "72 128"; do
  set -- $w
  GNN_TRAIN_FWD3=0 GNN_TRAIN_BWD3=0 OUT=/tmp/a.npz python3 tools/dbg/fwd3_widths.py $1 $2 > /dev/null
  GNN_TRAIN_FWD3=1 GNN_TRAIN_BWD3=0 OUT=/tmp/b.npz python3 tools/dbg/fwd3_widths.py $1 $2 > /dev/null
  GNN_TRAIN_FWD3=1 GNN_TRAIN_BWD3=1 OUT=/tmp/c.npz python3 tools/dbg/fwd3_widths.py $1 $2 > /dev/null
  GNN_TRAIN_FWD3=0 GNN_TRAIN_BWD3=1 OUT=/tmp/d.npz python3 tools/dbg/fwd3_widths.py $1 $2 > /dev/null
  python3 - $1 $2 <<'PY'
import numpy as np, sys
a, b, c, d = (np.load(f'/tmp/{x}.npz') for x in 'abcd')
def rel(x, y): return max(float(np.max(np.abs(x[k] - y[k])) / max(1e-30, np.max(np.abs(x[k])))) for k in x.files if k != 'loss')
print(f'hidden {sys.argv[1]} {sys.argv[2]}: largest relative gradient difference to the per-layer path: fwd3 only {rel(a, b):.2e}, fwd3 + bwd3 {rel(a, c):.2e}, bwd3 only {rel(a, d):.2e}; loss {float(a["loss"]):.6f} {float(b["loss"]):.6f}')
PY
done
