#!/bin/bash
# extra PMC passes of round 4 for the dominant kernel: matrix / vector co-execution, the vector L1's traffic towards L2 and its latency, TA busy.
# Each --pmc set in its own run, with --kernel-trace only.  Output under gpurun_out/prof_r04x/.
export TMPDIR=/tmp
OUT=gpurun_out/prof_r04x
mkdir -p $OUT
BENCH="python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-other-configs"
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_INST_CYCLES_VMEM_WR -d $OUT/coexec -o pmc --output-format csv -- $BENCH > $OUT/coexec.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum -d $OUT/tcp -o pmc --output-format csv -- $BENCH > $OUT/tcp.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUFFER_READ_WAVEFRONTS_sum TCP_TCC_WRITE_REQ_sum GRBM_GUI_ACTIVE -d $OUT/ta -o pmc --output-format csv -- $BENCH > $OUT/ta.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for sub in ('coexec', 'tcp', 'ta'):
    f = glob.glob(f'gpurun_out/prof_r04x/{sub}/**/*counter_collection.csv', recursive=True)
    if not f:
        print(sub, 'no counter file'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f[0])):
        acc[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    for k in acc:
        if 'k_fused' in k and 'true, true, false' in k:
            print(sub, k[:60], {c: sum(v) / len(v) for c, v in acc[k].items()}, 'launches', len(next(iter(acc[k].values()))))
PY
