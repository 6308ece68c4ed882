#!/bin/bash
# interleaved A/B of library variants on strict pass 1: bash tools/abp1.sh lib1 lib2 ...
export CG_PLACEMENT_DRAWS=1
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = base ]; then unset CG_HIP_LIB; else export CG_HIP_LIB=$PWD/exp/libcg_$lib.so; fi
  timeout -k 10 120 python -u tools/t_pass1.py 2>&1 | tail -1 || exit 1
done; done
