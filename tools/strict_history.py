#!/usr/bin/env python3
"""q1 after every iteration of a strict-order solve, as hex: python tools/strict_history.py <n> <iters> [strict_impl: 1 auto, 2 literal replay]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]); iters = int(sys.argv[2]); impl = int(sys.argv[3]) if len(sys.argv) > 3 else 1
s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, strict_impl=impl).init_problem()
s.solve(iters)
h = s.history(iters)
print(" ".join(float(x).hex() for x in h))
