#!/usr/bin/env python3
"""Condenses tools/prof_placement.sh's raw output into profiles/<tag>_placement_{good,bad}.md."""
import csv, glob, os, re, statistics, sys
out, tag = sys.argv[1], sys.argv[2]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

def last_launches(d, kernel="k_stencil<2, 5", count=10):
    """{counter: [value per launch]} and durations of the last `count` launches of the six-stream pass."""
    cc = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
    kt = glob.glob(f"{d}/**/*kernel_trace.csv", recursive=True)
    vals, durs = {}, []
    if cc:
        rows = [r for r in csv.DictReader(open(cc[0])) if kernel in r["Kernel_Name"]]
        ids = sorted({int(r["Dispatch_Id"]) for r in rows})[-count:]
        for r in rows:
            if int(r["Dispatch_Id"]) in ids:
                vals.setdefault(r["Counter_Name"], {}).setdefault(int(r["Dispatch_Id"]), []).append(float(r["Counter_Value"]))
    if kt:
        rows = [r for r in csv.DictReader(open(kt[0])) if kernel in r["Kernel_Name"]]
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        durs = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in rows[-count:]]
    return vals, durs

for keep, label in (("best", "good"), ("worst", "bad")):
    lines = [f"# Six-stream pass (pass 2 + u, 16384^2) on a {'kept (fastest of eight draws)' if keep == 'best' else 'rejected (slowest of eight draws)'} vector set: {tag}", "",
             "`tools/prof_placement.sh`: `rocprofv3 --kernel-trace --pmc <counters>` around `tools/placement_probe.py`, one run per counter group; "
             "the last ten launches of `k_stencil<2, 5, ...>` of each run. Each run is a new process with its own eight draws, so the "
             "sets differ from group to group; the pass time under the profiler is given with every group.", ""]
    for name in ("ea", "lvl", "tlb"):
        vals, durs = last_launches(f"{out}/{keep}_{name}")
        log = open(f"{out}/{keep}_{name}.log").read() if os.path.exists(f"{out}/{keep}_{name}.log") else ""
        m = re.search(r"keep=.*", log)
        lines.append(f"## counters: {name}  (probe: {m.group(0) if m else 'n/a'}; profiled launches: {statistics.mean(durs):.0f} us mean)" if durs else f"## counters: {name}")
        lines.append("")
        lines.append("| counter | mean per launch |")
        lines.append("|---|---|")
        for cname, per in sorted(vals.items()):
            tot = [sum(v) for v in per.values()]
            lines.append(f"| `{cname}` | {statistics.mean(tot):.4g} |")
        lines.append("")
    vals, durs = last_launches(f"{out}/{keep}_chan")
    if vals:
        lines.append(f"## per-channel requests (raw counters, one value per L2 channel instance; profiled launches: {statistics.mean(durs):.0f} us mean)" if durs else "## per-channel requests")
        lines.append("")
        lines.append("| counter | channels | min | mean | max | max/mean |")
        lines.append("|---|---|---|---|---|---|")
        for cname, per in sorted(vals.items()):
            # mean over launches of each channel's count (channels come in a fixed order per dispatch)
            width = min(len(v) for v in per.values())
            chans = [statistics.mean(v[i] for v in per.values()) for i in range(width)]
            if chans:
                mean = statistics.mean(chans)
                lines.append(f"| `{cname}` | {len(chans)} | {min(chans):.4g} | {mean:.4g} | {max(chans):.4g} | {max(chans) / mean if mean else 0:.3f} |")
        lines.append("")
    open(os.path.join(root, "profiles", f"{tag}_placement_{label}.md"), "w").write("\n".join(lines))
    print("\n".join(lines))
