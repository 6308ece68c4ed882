#!/bin/bash
# TCC / TLB counters of the six-stream pass on a kept (fast) and on a rejected (slow) vector set: bash tools/prof_placement.sh <tag>
set -u
tag=${1:-r03}; root=$(pwd); out=$root/gpurun_out/place_$tag; mkdir -p "$out"
export TMPDIR=/tmp CG_DEBUG_PLACEMENT=1
cd /tmp
run() {   # $1 keep, $2 pass name, $3... counters
  keep=$1; name=$2; shift 2
  CG_PLACEMENT_KEEP=$keep timeout -s KILL 150 rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d "$out/${keep}_$name" -- python3 "$root/tools/placement_probe.py" 16383 10 > "$out/${keep}_$name.log" 2>&1 || { tail -5 "$out/${keep}_$name.log"; return 1; }
  grep "keep=" "$out/${keep}_$name.log"
}
for keep in best worst; do
  run $keep ea TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum || exit 1
  run $keep lvl TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_TAG_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum || exit 1
  run $keep tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum || exit 1
  run $keep chan TCC_EA0_RDREQ TCC_EA0_WRREQ || exit 1
done
cd "$root"
python3 tools/prof_placement_summary.py "$out" "$tag"
