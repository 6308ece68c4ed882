#!/usr/bin/env python3
"""What the strict-order walks descend into as a solve proceeds: python tools/walk_stats.py [n] [checkpoints...]
   (run with CG_DEBUG_WALK=1 to get the whole tally of the last iteration's two walks on stderr)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
marks = [int(a) for a in sys.argv[2:]] or [21, 100, 300, 600, 1000]
s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT).init_problem()
done = 0
for m in marks:
    r = s.solve(m - done)
    done = m
    ev = s.get_option(cg.lib.OPT_STRICT_EVENTS)
    print(f"after {m:5d}: {1e3 * r.seconds / r.iters:7.3f} ms/iter  chunks taken apart d.q {ev // 1000000} g.g {ev % 1000000}  "
          f"redone {s.get_option(cg.lib.OPT_STRICT_MISSES)}  q1={float(r.q1).hex()}", flush=True)
