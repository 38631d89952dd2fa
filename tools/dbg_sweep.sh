#!/bin/bash
# timing experiments: GNN_FUSED_DEBUG bit sweep (results are numerically meaningless; only the kernel time is read)
for dbg in "$@"; do
  GNN_FUSED_DEBUG=$dbg python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print($dbg, d['roofline']['avg_launch_ms'])"
done
