#!/usr/bin/env python3
"""Strict-order pass 1 (and the fast one beside it) timed in one process: python tools/t_pass1.py [n] [reps]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT).init_problem()
s.solve(6)
out = []
for k in range(3):
    out.append(s.time_kernel(0, reps=reps)[0] * 1e3)
s.configure(reduce=cg.REDUCE_FAST).init_problem()
s.solve(4)
f = s.time_kernel(0, reps=reps)[0] * 1e3
print(f"{os.path.basename(os.environ.get('CG_HIP_LIB', 'default'))}: strict pass 1 " + " ".join(f"{t:.0f}" for t in out) + f" us; fast {f:.0f} us", flush=True)
