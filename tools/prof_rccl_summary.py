#!/usr/bin/env python3
"""Condense a tools/prof_rccl.sh run into profiles/<tag>.md: python tools/prof_rccl_summary.py <raw dir> <tag>"""
import csv, glob, os, sys
src, tag = sys.argv[1], sys.argv[2]
f = glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True)[0]
rows = list(csv.DictReader(open(f)))
log = [l for l in open(os.path.join(src, "trace.log")) if "us per iteration" in l or l.startswith("kernels only") or l.startswith("block ")]
lines = [f"# rocprofv3 kernel trace: {tag}", "",
         "Command: `python3 tools/rccl_rehearsal.py 16383 8 200` under `rocprofv3 --kernel-trace --stats` -- rank 1 of 8 of the",
         "16384^2 grid (2048 x 16384 block) on ONE GPU, the iteration's RCCL calls addressed to the rank itself",
         "(`cg_comm_init_self`): one grouped ncclSend/ncclRecv of two 128 KiB rows, two ncclAllGather of the partial sums; then the same in the",
         "reference's summation order, the accumulator passing through the ranks by published briefs or by the ranks walking in turn.", "",
         "Wall time per iteration (the tool's own output; profiler attached, so slightly slower than without):", "", "```"] + [l.rstrip() for l in log] + ["```", "",
         "| kernel | calls | mean us | total % |", "|---|---|---|---|"]
nccl = [r for r in rows if "nccl" in r["Name"].lower()]
if nccl:
    lines.insert(lines.index("| kernel | calls | mean us | total % |"), "RCCL kernels in the trace: " + ", ".join(f"`{r['Name'][:40]}` x {r['Calls']}" for r in nccl) + "\n")
for r in rows[:24]:
    lines.append(f"| `{r['Name'][:90]}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {r['Percentage']} |")
open(os.path.join("profiles", f"{tag}.md"), "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
