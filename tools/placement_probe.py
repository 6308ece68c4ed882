#!/usr/bin/env python3
"""The six-stream pass (pass 2 with the u update) on the vector set a context keeps, for the profiler:
   CG_PLACEMENT_KEEP=best|worst python tools/placement_probe.py [n] [reps]
prints the pass times; under `rocprofv3 --pmc ... --kernel-trace` the last `reps` launches of k_stencil<2, 5, ...>
are the ones to read (the placement draws of cg_create launch the same kernels before them)."""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
s = cg.CGSolver(n).init_problem()
s.solve(4)
t = [s.time_kernel(w, reps=reps)[0] for w in (0, 1, 2)]
print(f"keep={os.environ.get('CG_PLACEMENT_KEEP', 'best')}: pass1 {1e3 * t[0]:.0f} us  pass2 {1e3 * t[1]:.0f} us  pass2+u {1e3 * t[2]:.0f} us", flush=True)
