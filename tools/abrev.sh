#!/bin/bash
export CG_HIP_LIB=$PWD/exp/libcg_rev.so
for rep in 1 2 3; do for rev in 0 1; do
  echo -n "reverse=$rev "; CG_REVERSE_PASS2=$rev timeout -k 10 120 python -u tools/ab_strict.py 16383 21 200 2>&1 | tail -1 || exit 1
done; done
