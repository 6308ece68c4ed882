#!/usr/bin/env python3
"""Kernel timings of ONE rank's row block of a multi-GPU decomposition (no communication involved):
   python tools/sweep_block.py <n> <nranks> <tile_h list>"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n, P = int(sys.argv[1]), int(sys.argv[2])
tiles = [int(x) for x in sys.argv[3].split(",")] if len(sys.argv) > 3 else [0]
G = n + 1
rows = G // P
s = cg.CGSolver(n, row0=rows, rows=rows, rank=1, nranks=P)          # an interior block: halo rows on both sides
for th in tiles:
    s.configure(tile_h=th)
    line, tot = f"rows={rows} tile_h={th:4d}", 0.0
    for which in (0, 1, 2):
        ms, nb = s.time_kernel(which, reps=20)
        tot += ms if which == 0 else ms / 2
        line += f"  k{which}: {ms * 1e3:7.1f} us {nb / ms / 1e6:6.0f} GB/s"
    print(line + f"  per iteration {tot * 1e3:.1f} us", flush=True)
