#!/bin/bash
# Build a variant of the library in the build container: tools/build_variant.sh NAME [-DFLAG=..] ... -> build/variants/NAME.so
# (travels with the gpurun snapshot; tools/ab_libs.py compares prebuilt variants on the GPU box).  Same flags and the same parallel
# part build as the product library (rl4mm_amd/_lib.py:compile_library).
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; shift
mkdir -p $ROOT/build/variants
cd $ROOT
python -c 'import sys; from pathlib import Path; from rl4mm_amd import _lib; _lib.compile_library(Path("build/variants") / (sys.argv[1] + ".so"), sys.argv[2:])' "$NAME" "$@"
