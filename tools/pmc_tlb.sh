#!/bin/bash
# Address-translation and load-latency counters of the step kernel (one rocprofv3 run per group; names that this part does not
# have only fail their own run).  Usage (on the GPU box): bash tools/pmc_tlb.sh <outdir-under-gpurun_out>
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
rocprofv3 --list-avail > $OUT/avail.txt 2>&1
grep -i -E "utcl|tlb|translat|latency" $OUT/avail.txt | head -80 > $OUT/avail_tlb.txt
run() { name=$1; shift; timeout 120 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/$name -- python3 $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 5 --repeats 2 --no-cpu-baseline --no-facade ${PMC_ARGS:-} > $OUT.$name.log 2>&1; tail -1 $OUT.$name.log | cut -c1-120; }
run utcl1 TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_PERMISSION_MISS_sum
run lat TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
run tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_PENDING_STALL_CYCLES_sum
ls $OUT
