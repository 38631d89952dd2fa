#!/bin/bash
# round-5 profile runs on the GPU box:
#  (1) tools/profile.sh r05 (kernel trace + PMC passes of bench.py) and the bench line itself, (2) the extra PMC passes of the dominant kernel
#  (matrix / vector co-execution, vector-L1 traffic and latency, TA busy), (3) the per-rank kernels of configs[3]'s layouts (8 loopback ranks on one
#  GPU) under the kernel trace, (4) the C3-shaped and the MUTAG training step under the kernel trace.  Output under gpurun_out/; copy with
#  tools/collect_profile.py r05.
export TMPDIR=/tmp
O=gpurun_out
python3 bench.py > $O/bench_r05.json 2> $O/bench_r05.err
bash tools/profile.sh r05 > $O/profile_r05.log 2>&1
tail -25 $O/profile_r05.log
OUT=$O/prof_r05x
mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR -d $OUT/coexec -o pmc --output-format csv -- $BENCH > $OUT/coexec.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/tcp -o pmc --output-format csv -- $BENCH > $OUT/tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE -d $OUT/ta -o pmc --output-format csv -- $BENCH > $OUT/ta.log 2>&1
python3 - > $O/r05_k_fused_counters_extra.txt <<'PY'
import csv, glob, collections
print('# extra PMC passes of round 5 (tools/profile_r05.sh), per launch of the dominant kernel, averages over the launches of `python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs`')
for sub in ('coexec', 'tcp', 'ta'):
    f = glob.glob(f'gpurun_out/prof_r05x/{sub}/**/*counter_collection.csv', recursive=True)
    if not f:
        print(sub, 'no counter file'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        if 'k_fused' in k and 'true, true, false' in k:
            print(sub, k[:60], {c: sum(v) / len(v) for c, v in acc[k].items()}, 'launches', len(next(iter(acc[k].values()))))
PY
cat $O/r05_k_fused_counters_extra.txt
: > $O/r05_exchange_layouts.txt
for lay in full halo slice; do
  P=$O/prof_lay_r05_$lay
  mkdir -p $P
  LAYOUT=$lay SLICE_FORM=2 rocprofv3 --kernel-trace --stats -d $P -o s --output-format csv -- python3 tools/bench_slice.py > $P.log 2>&1
  (echo "== layout $lay"; tail -1 $P.log
  python3 - <<PY
import csv,glob
f=glob.glob('$P/**/*kernel_stats.csv', recursive=True)
rows=list(csv.DictReader(open(f[0])))
rows.sort(key=lambda r:-float(r['TotalDurationNs']))
for r in rows[:8]: print(f"{r['Name'][:100]:100s} calls {r['Calls']:>5s} avg {float(r['AverageNs'])/1e3:9.1f} us  {float(r['Percentage']):5.1f} %")
PY
  echo) >> $O/r05_exchange_layouts.txt
  rm -rf $P
done
cat $O/r05_exchange_layouts.txt
C3_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/prof_train -o train --output-format csv -- python3 tools/bench_train.py > $O/train_c3.log 2>&1
python3 - > $O/r05_train_c3.txt <<'PY'
import csv, glob
print('# C3-shaped training step (tools/bench_train.py C3_ONLY=1: 1 M nodes, 135->128->128->64 selu+BN, 5 bodies) under rocprofv3 --kernel-trace --stats, 4 steps')
print([l for l in open('gpurun_out/train_c3.log').read().splitlines() if l.startswith('C3-shaped')][-1])
for p in glob.glob('gpurun_out/prof_train/**/*kernel_stats.csv', recursive=True):
    for r in list(csv.DictReader(open(p)))[:22]:
        print('%-100s calls %5s avg %10.1f us  %6.2f %%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, float(r['Percentage'])))
PY
cat $O/r05_train_c3.txt
rm -rf $O/prof_train
