#!/bin/bash
# 20-step bench at several graph lengths, alternating, in one gpurun call: bash tools/ab_graph.sh <lib> <reps> len1 len2 ...
lib=$1; reps=$2; shift 2
export CG_HIP_LIB=$PWD/exp/libcg_$lib.so
for rep in $(seq $reps); do for g in "$@"; do
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side --graph-iters $g > gpurun_out/ab_graph.json 2> gpurun_out/ab_graph.err || exit 1
  python - "$g" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_graph.json").read().strip().splitlines()[-1])
t, s = d["ms_per_step"], d["steady_state"]["ms_per_step"]
print(f"graph {sys.argv[1]:3s} {d['value']:7.1f} it/s  timed {t:.3f}  steady {s:.3f}  fixed {(t - s) * d['steps']:.2f} ms", flush=True)
PY
done; done
