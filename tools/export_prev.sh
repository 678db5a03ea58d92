#!/bin/bash
# Export the kernel sources of a git revision to build/prev_csrc/ for A/B runs with tools/ab_variants.py.
rev=${1:-HEAD}
mkdir -p build/prev_csrc/rl4mm_amd/csrc build/prev_csrc/include
for f in lobsim_kernels.hip lobsim_host.inc lobsim_layout.h; do git show $rev:rl4mm_amd/csrc/$f > build/prev_csrc/rl4mm_amd/csrc/$f; done
git show $rev:include/lobsim.h > build/prev_csrc/include/lobsim.h
echo "exported $rev"
