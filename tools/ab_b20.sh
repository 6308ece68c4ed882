#!/bin/bash
# alternating 20-step bench runs of library variants in one gpurun call: bash tools/ab_b20.sh <reps> lib1 lib2 ...
# prints per run: value, ms/step of the timed region, of the steady state, and the region's fixed cost in ms
reps=$1; shift
for rep in $(seq $reps); do for lib in "$@"; do
  if [ "$lib" = base ]; then unset CG_HIP_LIB; else export CG_HIP_LIB=$PWD/exp/libcg_$lib.so; fi
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --no-side > gpurun_out/ab_b20.json 2> gpurun_out/ab_b20.err || exit 1
  python - "$lib" <<'PY'
import json, sys
d = json.loads(open("gpurun_out/ab_b20.json").read().strip().splitlines()[-1])
t, s = d["ms_per_step"], d["steady_state"]["ms_per_step"]
print(f"{sys.argv[1]:8s} {d['value']:7.1f} it/s  timed {t:.3f}  steady {s:.3f}  fixed {(t - s) * d['steps']:.2f} ms", flush=True)
PY
done; done
