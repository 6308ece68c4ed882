"""Per-100-iteration times of a cold strict solve continued in slices (nothing read in between): python tools/t_trend.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
s = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
for rep in range(2):
    s.init_problem()
    out = []
    for k in range(10):
        r = s.solve(100)
        ev = s.get_option(cg.lib.OPT_STRICT_EVENTS)
        out.append(f"{1e3 * r.seconds / 100:.3f} ({ev // 1000000}+{ev % 1000000}, {s.get_option(cg.lib.OPT_STRICT_MISSES)})")
    print(f"rep {rep}: " + "  ".join(out), flush=True)
