#!/usr/bin/env python3
"""Solve rate with an integer option at several values, alternated inside ONE process (one context, one placement):
   python tools/ab_option.py <option key> <v1,v2,...> [n] [iters] [reps] [strict|fast]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
key = int(sys.argv[1])
vals = [int(v) for v in sys.argv[2].split(",")]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 16383
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 100
reps = int(sys.argv[5]) if len(sys.argv) > 5 else 3
red = cg.REDUCE_FAST if (len(sys.argv) > 6 and sys.argv[6] == "fast") else cg.REDUCE_STRICT
s = cg.CGSolver(n).configure(reduce=red).init_problem()
s.solve(20)
for rep in range(reps):
    for v in vals:
        s.set_option(key, v)
        s.solve(20)                      # graph for this setting, warm
        r = s.solve(iters)
        print(f"option {key} = {v}: {1e3 * r.seconds / r.iters:7.3f} ms/iter = {r.iters / r.seconds:6.1f} it/s  q1={float(r.q1).hex()}", flush=True)
