#!/bin/bash
# on the GPU box: bench.py once per library of the list, ROUNDS times round-robin (box noise is 1 - 2 %, so A B A B, not A A B B);
# one line per run: library, ms per Loop, kernel ms per launch (live HIP events), roofline fraction, exact-path ms per Loop
# usage: tools/ab_run.sh OUTNAME ROUNDS name1 name2 ...     (name "product" = libgnn_hip.so, else libgnn_hip_ab_<name>.so)
OUT=gpurun_out/ab_$1.txt; ROUNDS=$2; shift 2
: > $OUT
for r in $(seq $ROUNDS); do
  for n in "$@"; do
    if [ "$n" = product ]; then L=$PWD/gnn_tf_2.x_amd/GNN/libgnn_hip.so; else L=$PWD/gnn_tf_2.x_amd/GNN/libgnn_hip_ab_$n.so; fi
    GNN_HIP_LIBRARY=$L python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs ${AB_BENCH_ARGS} 2>>$OUT.err | python3 -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j['roofline']; x=j['config'].get('exact_f32_mfma_path') or {}
    print('%-14s ms/step %.3f kernel_ms %.4f frac %.3f exact_ms/step %.3f' % ('$n', j['ms_per_step'], r['avg_launch_ms'], r['frac'], x.get('ms_per_step',0)))
" >> $OUT
  done
done
cat $OUT
