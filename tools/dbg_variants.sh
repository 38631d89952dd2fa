for v in 1 33 65 97; do
  GNN_HIP_LIBRARY=$PWD/gnn_tf_2.x_amd/GNN/libgnn_hip_diag.so GNN_FUSED_VARIANT=$v timeout -k 10 120 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('variant $v', round(d['roofline']['avg_launch_ms'],4), 'ms per launch')"
done
