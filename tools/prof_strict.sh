#!/bin/bash
# kernel trace of a strict-order solve (run on the GPU box from the repo root): bash tools/prof_strict.sh <tag> <n> <iters>
set -u
tag=$1; n=${2:-16383}; iters=${3:-20}
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
CG_STRICT=1 timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- "$root/src/cg" $n $iters > "$out/trace.log" 2>&1 || { tail -5 "$out/trace.log"; exit 1; }
cd "$root"; f=$(find "$out/trace" -name "*kernel_stats.csv" | head -1); cp "$f" profiles/${tag}_kernel_stats.csv
python3 - "$f" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:14]:
    print(f'{r["Name"][:70]:70s} calls={r["Calls"]:>5s} avg_us={float(r["AverageNs"])/1e3:9.1f} pct={r["Percentage"]}')
PY
tail -2 "$out/trace.log"
