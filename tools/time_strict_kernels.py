#!/usr/bin/env python3
"""Pass times of the lean schedule, fast order against strict order with in-pass summaries, in one process:
   python tools/time_strict_kernels.py [n]   (CG_HIP_LIB selects a library variant)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
out = []
for red in (cg.REDUCE_FAST, cg.REDUCE_STRICT):
    s = cg.CGSolver(n).configure(reduce=red).init_problem()
    s.solve(4)                      # real data and hypotheses in place
    out.append([s.time_kernel(w, reps=20)[0] * 1e3 for w in (0, 1, 2)])
    s.close()
f, t = out
print(f"{os.environ.get('CG_HIP_LIB', 'default')}: " + "  ".join(f"{nm} {a:.0f} -> {b:.0f} us (+{b - a:.0f})" for nm, a, b in zip(("pass1", "pass2", "pass2+u"), f, t)), flush=True)
