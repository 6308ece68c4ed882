#!/bin/bash
# rocprofv3 evidence for the benchmark command: kernel trace + stats, then HBM byte counters in
# separate passes (FETCH_SIZE and WRITE_SIZE do not fit one pass).  Run on the GPU box from the repo root:
#   bash tools/prof.sh <tag> [bench.py args...]
# Raw output lands in gpurun_out/prof_<tag>/; tools/prof_summary.py condenses it for profiles/.
set -u
tag=$1; shift
root=$(pwd)
out=$root/gpurun_out/prof_$tag
mkdir -p "$out"
export TMPDIR=/tmp
cd /tmp
args="--steps 20 --warmup 6 --no-cpu-baseline --no-parity $*"
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/bench.py" $args > "$out/trace.log" 2>&1 || { echo "trace pass failed"; tail -5 "$out/trace.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$out/fetch" -- python3 "$root/bench.py" $args > "$out/fetch.log" 2>&1 || { echo "fetch pass failed"; tail -5 "$out/fetch.log"; exit 1; }
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$out/write" -- python3 "$root/bench.py" $args > "$out/write.log" 2>&1 || { echo "write pass failed"; tail -5 "$out/write.log"; exit 1; }
cd "$root"
python3 tools/prof_summary.py "$out" "$tag"
