#!/bin/bash
# tuning sweep of the fused kernel on the GPU box (diagnostic build: make -C gnn_tf_2.x_amd/csrc DIAG=1, loaded through GNN_HIP_LIBRARY):
# one bench line per setting, kernel ms and roofline fraction
export GNN_HIP_LIBRARY=${GNN_HIP_LIBRARY:-$PWD/gnn_tf_2.x_amd/GNN/libgnn_hip_diag.so}
OUT=gpurun_out/sweep_${1:-x}.txt
: > $OUT
run() {
  echo "== $*" >> $OUT
  env "$@" python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs 2>>$OUT | python3 -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j['roofline']; x=j['config'].get('exact_f32_mfma_path') or {}
    print('ms/step %.3f kernel_ms %.4f frac %.3f exact_ms %.3f maxdiff %.2e onebody %.2e' % (j['ms_per_step'], r['avg_launch_ms'], r['frac'], x.get('ms_per_step',0), x.get('max_abs_state_difference_to_default_path',0), x.get('max_abs_state_difference_after_one_body',0)))
" >> $OUT
}
shift
for s in "$@"; do run $s; done
cat $OUT
