#!/bin/bash
# round-4 profile runs on the GPU box: (1) tools/profile.sh r04 (kernel trace + PMC passes of bench.py), (2) the per-rank kernels of configs[3]'s
# layouts (8 loopback ranks on one GPU) under the kernel trace.  Output under gpurun_out/.
export TMPDIR=/tmp
bash tools/profile.sh r04 > gpurun_out/profile_r04.log 2>&1
tail -25 gpurun_out/profile_r04.log
for lay in full slice; do
  OUT=gpurun_out/prof_slice_r04_$lay
  mkdir -p $OUT
  LAYOUT=$lay SLICE_FORM=2 rocprofv3 --kernel-trace --stats -d $OUT -o s --output-format csv -- python3 tools/bench_slice.py > $OUT.log 2>&1
  echo "== layout $lay"; tail -1 $OUT.log
  python3 - <<PY
import csv,glob
f=glob.glob('$OUT/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]: print(f"{r['Name'][:90]:90s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
done
