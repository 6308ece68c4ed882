#!/bin/bash
# Kernel trace of the single-GPU rehearsal of a multi-rank iteration (tools/rccl_rehearsal.py): names and durations
# of the RCCL kernels beside the solver's own.  Run on the GPU box from the repo root: bash tools/prof_rccl.sh <tag>
set -u
tag=$1
root=$(pwd); out=$root/gpurun_out/prof_$tag; mkdir -p "$out"
export TMPDIR=/tmp; cd /tmp
export CG_REHEARSAL_EAGER=1     # the strict flavours with RCCL inside a captured graph crash the tracer (not the program)
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$out/trace" -- python3 "$root/tools/rccl_rehearsal.py" 16383 8 200 > "$out/trace.log" 2>&1 || { tail -5 "$out/trace.log"; exit 1; }
cd "$root"
python3 tools/prof_rccl_summary.py "$out" "$tag"
