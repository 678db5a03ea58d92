#!/bin/bash
# Build a variant of the library in the build container: tools/build_variant.sh NAME [-DFLAG=..] ... -> build/variants/NAME.so
# (travels with the gpurun snapshot; tools/ab_libs.py compares prebuilt variants on the GPU box)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build/variants
cd $ROOT/rl4mm_amd/csrc
hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -fPIC -shared -std=c++17 -mllvm -amdgpu-atomic-optimizer-strategy=None -mllvm -amdgpu-sched-strategy=max-ilp "$@" -o $ROOT/build/variants/$NAME.so lobsim_kernels.hip
