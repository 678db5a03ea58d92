#!/bin/bash
# Registers / spills / scratch of the kernels of ONE part of the library (build container, ~70 s): tools/part_resources.sh PART [-DFLAG ...]
# (PART 1 .. 4 = (ladder 64 | 128) x (agent table 128 | 256): rl4mm_amd/_lib.py; 3 = the default handle's kernels)
ROOT=$(cd "$(dirname "$0")/.." && pwd)
PART=$1; shift
OUT=$(mktemp /tmp/part_XXXX.log)
cd $ROOT/rl4mm_amd/csrc
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-sched-strategy=max-ilp -mllvm -disable-machine-licm -Wno-pass-failed \
  -DLOBSIM_PART=$PART "$@" -Rpass-analysis=kernel-resource-usage -c -o /dev/null lobsim_kernels.hip > $OUT 2>&1
grep -E "error" -A3 $OUT | head -20
python3 $ROOT/tools/res_usage.py $OUT
rm -f $OUT
