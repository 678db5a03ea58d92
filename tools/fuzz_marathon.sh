#!/bin/bash
# A longer run of the GPU-side fuzzers on fresh seeds: bash tools/fuzz_marathon.sh <tag> -> gpurun_out/<tag>/ (about half an hour on one box)
set -u
OUT=gpurun_out/${1:-marathon}; mkdir -p $OUT
timeout 1500 python3 tests/diag/config_fuzz.py 40000 44000 64 > $OUT/config_fuzz.log 2>&1; tail -1 $OUT/config_fuzz.log
FUZZ_BOUNDS=1 timeout 900 python3 tests/diag/config_fuzz.py 45000 46000 64 > $OUT/config_fuzz_bounds.log 2>&1; tail -1 $OUT/config_fuzz_bounds.log
timeout 600 python3 tests/diag/simulator_fuzz.py 5000 5600 > $OUT/simulator_fuzz.log 2>&1; tail -1 $OUT/simulator_fuzz.log
timeout 600 python3 tests/diag/vector_env_fuzz.py 10000 12000 48 > $OUT/vector_env_fuzz.log 2>&1; tail -1 $OUT/vector_env_fuzz.log
grep -h -A3 "DIFF\|ERROR" $OUT/*.log | cut -c1-600 | head -40
