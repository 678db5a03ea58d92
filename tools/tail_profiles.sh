#!/bin/bash
# The slowest-wave profiles of profiles/<round>/ (stamp build: tools/build_variant.sh prof -DLOBSIM_PROFILE -DLOBSIM_STEP_WAVES=4 - the stamps
# and event counters need registers: on the product's 5-wave budget the stamp build spills and reads 15 % slow).  bash tools/tail_profiles.sh <tag>
set -u
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/$1; mkdir -p $OUT; cd $R
P=build/variants/prof.so
PROF_LIB=$P timeout 400 python3 tools/tail_profile.py 4096 rows > $OUT/tail_profile.txt 2>/dev/null
for w in deep50 rs; do PROF_LIB=$P TP_WORKLOAD=$w timeout 400 python3 tools/tail_profile.py > $OUT/tail_profile_$w.txt 2>/dev/null; done
PROF_LIB=$P TP_GENERIC=1 timeout 400 python3 tools/tail_profile.py > $OUT/tail_profile_generic.txt 2>/dev/null
PROF_LIB=$P TP_AGENT_ORDERS=128 timeout 400 python3 tools/tail_profile.py > $OUT/tail_profile_agent128.txt 2>/dev/null
PROF_LIB=$P TP_SKIP=5 TP_K=105 timeout 400 python3 tools/tail_profile.py > $OUT/tail_profile_steps5_105.txt 2>/dev/null
PROF_LIB=$P TP_SKIP=500 TP_K=600 timeout 400 python3 tools/tail_profile.py > $OUT/tail_profile_steps500_600.txt 2>/dev/null
grep -h "MAX" $OUT/tail_profile*.txt
