#!/bin/bash
# GPU box, round 5 (first batch; the training table of the FINAL code is profiles/r05_train_c3.txt): training-step kernel breakdown, 4-rank stand-in rehearsals of bench.py at full size, counters of the sliced aggregation
export TMPDIR=/tmp
O=$PWD/gpurun_out
mkdir -p $O
# 1. C3-shaped training step and the MUTAG step under the kernel trace
C3_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/prof_train -o train --output-format csv -- python3 tools/bench_train.py > $O/train_c3.log 2>&1
python3 - <<'PY' > $O/train_c3_kernels.txt 2>&1
import csv, glob, collections
f = glob.glob('gpurun_out/prof_train/**/*kernel_stats.csv', recursive=True)
print(open('gpurun_out/train_c3.log').read().strip().splitlines()[-1])
for p in f:
    for r in list(csv.DictReader(open(p)))[:24]:
        print('%-100s calls %5s avg %10.1f us  %5s %%' % (r['Name'][:100], r['Calls'], float(r['AverageNs']) / 1e3, r['Percentage']))
PY
cat $O/train_c3_kernels.txt
SMALL_ONLY=1 rocprofv3 --kernel-trace --stats -d $O/prof_train_mutag -o train --output-format csv -- python3 tools/bench_train_mutag.py > $O/train_mutag.log 2>&1
tail -3 $O/train_mutag.log
# 2. stand-in rehearsals: 4 rank processes on this one GPU over tests/mock_rccl (timings are those of /dev/shm and a shared GPU, NOT of xGMI)
for ex in halo full slice; do
  GNN_RCCL_LIBRARY=$PWD/tests/mock_rccl/libmock_rccl.so GNN_BENCH_ONE_DEVICE=1 OMP_NUM_THREADS=2 timeout -k 10 500 python3 bench.py --gpus 4 --steps 1 --warmup 1 --exchange $ex > $O/standin_$ex.json 2> $O/standin_$ex.err
  python3 -c "
import json,sys
j=json.load(open('$O/standin_$ex.json'))
print('$ex', 'parity_ok', j['parity_ok'], j['multi_gpu']['sharded_check'])
"
done
# 3. sliced aggregation (P = 8, 8 columns per rank): kernel trace, then cache counters in their own passes
for pass in trace "pmc FETCH_SIZE" "pmc TCC_HIT_sum TCC_MISS_sum WRITE_SIZE" "pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum"; do
  tag=$(echo $pass | tr ' ' '_')
  if [ "$pass" = trace ]; then
    WORLD=8 LAYOUT=slice SLICE_FORM=2 rocprofv3 --kernel-trace --stats -d $O/prof_slice_$tag -o s --output-format csv -- python3 tools/bench_slice.py > $O/slice_$tag.log 2>&1
  else
    WORLD=8 LAYOUT=slice SLICE_FORM=2 rocprofv3 --kernel-trace --${pass} -d $O/prof_slice_$tag -o s --output-format csv -- python3 tools/bench_slice.py > $O/slice_$tag.log 2>&1
  fi
done
python3 - <<'PY' > $O/slice_counters.txt 2>&1
import csv, glob, collections
for p in sorted(glob.glob('gpurun_out/prof_slice_*/**/*kernel_stats.csv', recursive=True)):
    for r in list(csv.DictReader(open(p)))[:6]:
        print('%-90s calls %5s avg %10.1f us' % (r['Name'][:90], r['Calls'], float(r['AverageNs']) / 1e3))
for p in sorted(glob.glob('gpurun_out/prof_slice_pmc*/**/*counter_collection.csv', recursive=True)):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(p)):
        if 'k_spmm<4, true>' in r['Kernel_Name']:
            a = acc[r['Counter_Name']]; a[0] += 1; a[1] += float(r['Counter_Value'])
    for k, (n, v) in acc.items(): print(f'k_spmm<4,true>  {k:32s} per launch {v / n:.4g}  ({n} launches)')
PY
cat $O/slice_counters.txt
rm -rf $O/prof_train $O/prof_train_mutag $O/prof_slice_*
