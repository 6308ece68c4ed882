#!/bin/bash
# interleaved A/B of library variants in ONE gpurun call: pass times and strict solve rate.  bash tools/abk.sh lib1 lib2 ...  (base = the in-tree library)
for rep in 1 2; do for lib in "$@"; do
  if [ "$lib" = base ]; then unset CG_HIP_LIB; else export CG_HIP_LIB=$PWD/exp/libcg_$lib.so; fi
  timeout -k 10 120 python -u tools/time_strict_kernels.py 2>&1 | tail -1 || exit 1
  timeout -k 10 120 python -u tools/ab_strict.py 16383 21 200 2>&1 | tail -1 || exit 1
done; done
