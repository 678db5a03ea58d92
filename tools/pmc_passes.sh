#!/bin/bash
# PMC passes for the step kernel, one rocprofv3 run per counter group (never combined with tracing domains other
# than --kernel-trace).  Usage (on the GPU box): [PMC_ARGS="--workload deep50"] [PMC_ONLY="sq1 fetch write"] bash tools/pmc_passes.sh <outdir-under-gpurun_out>
# (the program goes directly after `--`: no env / bash -c hop between rocprofv3 and python3)
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
run() { name=$1; shift; if [ -n "${PMC_ONLY:-}" ] && [[ " $PMC_ONLY " != *" $name "* ]]; then return; fi; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --repeats 2 --no-cpu-baseline --no-facade ${PMC_ARGS:-} > $OUT.$name.log 2>&1; tail -1 $OUT.$name.log | cut -c1-120; }
mkdir -p $OUT
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU
run sq2 SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM
run fetch FETCH_SIZE
run write WRITE_SIZE TCC_HIT_sum TCC_MISS_sum
run lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_FLAT SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU

run icache SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SMEM SQC_DCACHE_MISSES
ls $OUT
