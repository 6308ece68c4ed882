#!/bin/bash
# interleaved A/B of library builds in ONE gpurun call (same box): bash tools/ab.sh "<sweep args>" lib1 lib2 ...
args=$1; shift
for rep in 1 2 3; do for lib in "$@"; do
  if [ "$lib" = base ]; then unset CG_HIP_LIB; else export CG_HIP_LIB=$PWD/exp/libcg_$lib.so; fi
  echo -n "$lib: "; timeout -k 10 120 python -u tools/sweep.py $args 2>&1 | tail -1
done; done
