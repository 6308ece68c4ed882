#!/bin/bash
# mean duration of the strict order's small kernels for library variants, from a kernel trace: bash tools/walk_prof.sh lib1 lib2 ...  (base = in-tree)
root=$(pwd); export TMPDIR=/tmp
for lib in "$@"; do
  if [ "$lib" = base ]; then unset CG_HIP_LIB; else export CG_HIP_LIB=$root/exp/libcg_$lib.so; fi
  out=$root/gpurun_out/walkprof_$lib; rm -rf $out; mkdir -p $out
  (cd /tmp && timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $root/tools/ab_strict.py 16383 21 200 > $out/log 2>&1) || { tail -3 $out/log; exit 1; }
  python3 - "$lib" $out <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[2] + "/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[len(rows) // 2:]                       # the steady half
acc = {}
for r in rows:
    n = r["Kernel_Name"].replace("cg::", "").replace("exact::", "").replace("void ", "").split("(")[0]
    if "k_stencil" in n: continue
    acc.setdefault(n, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
print(sys.argv[1], "  ".join(f"{n[:28]} {sum(v)/len(v):.1f}" for n, v in sorted(acc.items()) if len(v) > 20))
PY
done
