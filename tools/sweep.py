#!/usr/bin/env python3
"""Kernel-level sweep of launch geometry on the 16384^2 grid (HIP-event timed, GB/s of algorithmic bytes)."""
import importlib, itertools, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
tiles = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [64, 128, 256, 512, 1024]
scheds = {"noq": cg.SCHED_FUSED_NOQ, "classic": cg.SCHED_CLASSIC, "fused": cg.SCHED_FUSED, "lean": cg.SCHED_LEAN}
which_s = sys.argv[3].split(",") if len(sys.argv) > 3 else ["noq"]
pfs = [int(x) for x in sys.argv[4].split(",")] if len(sys.argv) > 4 else [0]
wgs = [int(x) for x in sys.argv[5].split(",")] if len(sys.argv) > 5 else [0]
extra = dict(kv.split("=") for kv in sys.argv[6].split(",")) if len(sys.argv) > 6 else {}     # e.g. u_defer=4,prefetch_read=8
extra = {k: int(v) for k, v in extra.items()}
s = cg.CGSolver(n)
for name in which_s:
    sched = scheds[name]
    for pf in pfs:
        for th, wg in itertools.product(tiles, wgs):
            s.configure(schedule=sched, tile_h=th, xcd_remap=1, prefetch=pf, tile_wgs=wg, **extra).init_problem()
            line = f"pf={pf} {name:8s} tile_h={th:5d} wgs={wg:4d}"
            tot = 0.0
            for which in (0, 1) + ((2,) if name == "lean" else ()):
                ms, nb = s.time_kernel(which, reps=10)
                tot += ms / 2 if (name == "lean" and which > 0) else ms     # lean: passes k1 and k2 alternate
                line += f"  k{which}: {ms:7.3f} ms {nb/ms/1e6:7.0f} GB/s"
            print(line + f"  sum {tot:.3f}", flush=True)
