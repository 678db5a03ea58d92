#!/bin/bash
# The round's closing validation on one GPU box: bash tools/final_validation.sh <tag> -> gpurun_out/<tag>/ (the whole GPU suite, then the fuzzers on
# fresh seeds: the Python boundary on the diagnostic build and at 256 envs, drawn configurations on the product library)
set -u
OUT=gpurun_out/${1:-final}; mkdir -p $OUT
timeout 1500 python3 -m pytest tests -m gpu -q --timeout 300 --timeout-method=thread > $OUT/pytest_gpu.log 2>&1; tail -3 $OUT/pytest_gpu.log
LOBSIM_LIB=rl4mm_amd/csrc/liblobsim_bounds.so timeout 600 python3 tests/diag/vector_env_fuzz.py 3000 3400 32 > $OUT/vector_env_fuzz_bounds.log 2>&1; tail -1 $OUT/vector_env_fuzz_bounds.log
timeout 600 python3 tests/diag/vector_env_fuzz.py 5000 5300 256 > $OUT/vector_env_fuzz_256.log 2>&1; tail -1 $OUT/vector_env_fuzz_256.log
timeout 900 python3 tests/diag/config_fuzz.py 30000 31200 64 > $OUT/config_fuzz.log 2>&1; tail -1 $OUT/config_fuzz.log; grep -c " OK " $OUT/config_fuzz.log
grep -h -A3 "DIFF" $OUT/*.log | cut -c1-500 | head -40
