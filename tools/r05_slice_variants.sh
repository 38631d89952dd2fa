#!/bin/bash
# sliced aggregation variants (GNN_SPMM_NARROW_U: entries in flight per lane; 4 = shipped form with a serial tail, 5 = masked batches of 5, 8, 16)
export TMPDIR=/tmp
O=$PWD/gpurun_out
for u in 4 5 8 16; do
  WORLD=8 LAYOUT=slice SLICE_FORM=2 GNN_SPMM_NARROW_U=$u rocprofv3 --kernel-trace --stats -d $O/prof_sv_$u -o s --output-format csv -- python3 tools/bench_slice.py > $O/sv_$u.log 2>&1
  python3 - $u <<'PY'
import csv, glob, sys
u = sys.argv[1]
for p in sorted(glob.glob(f'gpurun_out/prof_sv_{u}/**/*kernel_stats.csv', recursive=True)):
    for r in list(csv.DictReader(open(p)))[:3]:
        if 'k_spmm' in r['Name']: print(f'U={u}: %-60s calls %5s avg %8.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs']) / 1e3))
PY
  rm -rf $O/prof_sv_$u
done
