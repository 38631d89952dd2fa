#!/bin/bash
# on the GPU box: bench.py with --tile-form 1 / 2 round-robin (ROUNDS times), optional extra bench args after the count
OUT=gpurun_out/ab_forms_$1.txt; ROUNDS=$2; shift 2
: > $OUT
for r in $(seq $ROUNDS); do
  for f in 1 2; do
    python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-other-configs --tile-form $f "$@" 2>>$OUT.err | python3 -c "
import json,sys
for l in sys.stdin:
    try: j=json.loads(l)
    except Exception: continue
    r=j['roofline']
    print('form $f  ms/step %.3f kernel_ms %.4f frac %.3f  (%s)' % (j['ms_per_step'], r['avg_launch_ms'], r['frac'], j['config'].get('tile_form')))
" >> $OUT
  done
done
cat $OUT
