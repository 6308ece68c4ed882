#!/bin/bash
# round-3 first look on the GPU box: strict baseline, walk tallies, instruction costs, SQ counters of the strict passes
set -u
root=$(pwd); out=$root/gpurun_out/probe; mkdir -p "$out"
export TMPDIR=/tmp
timeout -k 10 200 python -u tools/ab_strict.py 16383 21 200 > "$out/ab.log" 2>&1 || { tail -5 "$out/ab.log"; exit 1; }
cat "$out/ab.log"
CG_DEBUG_WALK=1 timeout -k 10 200 python -u tools/walk_stats.py 16383 21 100 300 > "$out/walk.log" 2>&1 || { tail -5 "$out/walk.log"; exit 1; }
cat "$out/walk.log"
timeout -k 10 200 python -u tools/time_strict_kernels.py > "$out/tk.log" 2>&1 || { tail -5 "$out/tk.log"; exit 1; }
cat "$out/tk.log"
timeout -k 10 200 tools/ibench > "$out/ibench.log" 2>&1 || { tail -5 "$out/ibench.log"; exit 1; }
cd /tmp
CG_STRICT=1 CG_PLACEMENT_DRAWS=1 timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAIT_INST_LDS SQ_BUSY_CYCLES --output-format csv -d "$out/pmc1" -- "$root/src/cg" 16383 8 > "$out/pmc1.log" 2>&1 || { tail -5 "$out/pmc1.log"; exit 1; }
CG_STRICT=1 CG_PLACEMENT_DRAWS=1 timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d "$out/pmc2" -- "$root/src/cg" 16383 8 > "$out/pmc2.log" 2>&1 || { tail -5 "$out/pmc2.log"; exit 1; }
cd "$root"
python3 - "$out" <<'PY'
import csv, glob, sys, collections
for p in ("pmc1", "pmc2"):
    for f in glob.glob(f"{sys.argv[1]}/{p}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(f)):
            acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, d in acc.items():
            if "k_stencil" in k or "k_sp_" in k or "walk" in k or "plan" in k:
                print(p, k, {c: round(sum(v) / len(v)) for c, v in d.items()}, "launches", len(next(iter(d.values()))))
PY
