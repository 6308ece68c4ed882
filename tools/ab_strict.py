#!/usr/bin/env python3
"""Strict-order solve rate of the library given by CG_HIP_LIB: python tools/ab_strict.py [n] [warm] [iters]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
warm = int(sys.argv[2]) if len(sys.argv) > 2 else 21
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 100
s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT).init_problem()
s.solve(warm)
r = s.solve(iters)
ev = s.get_option(cg.lib.OPT_STRICT_EVENTS)
print(f"{os.environ.get('CG_HIP_LIB', 'default')}: {1e3 * r.seconds / r.iters:7.3f} ms/iter = {r.iters / r.seconds:6.1f} it/s  q1={float(r.q1).hex()}"
      f"  chunks taken apart by the last walks: d.q {ev // 1000000}, g.g {ev % 1000000}; chunks redone: {s.get_option(cg.lib.OPT_STRICT_MISSES)}", flush=True)
