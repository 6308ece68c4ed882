#!/usr/bin/env python3
"""Strict-order solve rate against launch geometry: python tools/sweep_strict.py [n] [iters]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 120
for wgs in (0, 512):
    for pf in (0, 1, 2, 4):
        for pfr in (0, 4):
            s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, tile_wgs=wgs, prefetch=pf, prefetch_read=pfr).init_problem()
            s.solve(21)
            r = s.solve(iters)
            print(f"tile_wgs={wgs:4d} prefetch={pf} prefetch_read={pfr}: {1e3 * r.seconds / r.iters:7.3f} ms/iter = {r.iters / r.seconds:6.1f} it/s", flush=True)
            s.close()
