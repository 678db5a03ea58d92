#!/bin/bash
# Dynamic instruction counts per message kind: rocprofv3 --pmc on tools/msg_cost_probe.py single configs (R messages
# per env-step for every env), counters per dispatch of the step kernel.  Usage on the GPU box: bash tools/pmc_msg.sh
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/pmc_msg
cd /tmp && export TMPDIR=/tmp
mkdir -p $OUT
for cfgs in "pair 2 12" "pair 2 48" "pair 40 12" "pair 40 48" "market 0 12" "market 0 48"; do
  set -- $cfgs
  tag=$1_$2_$3
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/$tag -- python3 $GRAFT_REPO_ROOT/tools/msg_cost_probe.py 4096 single $1 $2 $3 > $OUT/$tag.log 2>&1
  echo "$tag: $(tail -1 $OUT/$tag.log)"
  python3 $GRAFT_REPO_ROOT/tools/pmc_summary.py $OUT/$tag lob_step_kernel
done
