#!/bin/bash
# two PMC passes on the fused kernel only
TAG=${1:-q}
OUT=gpurun_out/pmc_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline ${BENCH_ARGS:-}"
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES -d $OUT/a -o pmc --output-format csv -- $BENCH > $OUT/a.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT/b -o pmc --output-format csv -- $BENCH > $OUT/b.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_FLAT SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/c -o pmc --output-format csv -- $BENCH > $OUT/c.log 2>&1
python3 tools/summarize_prof.py $OUT 2>&1 | grep -E "k_fused|==" 
