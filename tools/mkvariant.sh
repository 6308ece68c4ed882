#!/bin/bash
# build a library variant for an A/B on the GPU box: bash tools/mkvariant.sh <name> [extra hipcc flags...]  -> exp/libcg_<name>.so
name=$1; shift
mkdir -p exp
hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -shared -I/opt/rocm/include -Wall -Wno-unused-function "$@" \
  conjugate-gradient_amd/csrc/cg_lib.hip -o exp/libcg_$name.so -ldl
