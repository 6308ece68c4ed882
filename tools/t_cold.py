"""Cold starts in strict order: wall time of init_problem + solve(k) for a few k, and the bits of q1: python tools/t_cold.py [n]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
s = cg.CGSolver(n).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
s.init_problem(); s.solve(30)                 # graphs, workspace
out = []
for k in (1, 2, 3, 5, 11, 21, 41, 81):
    best = 1e9
    for rep in range(3):
        s.init_problem()
        r = s.solve(k)
        best = min(best, r.seconds)
    out.append(f"{k}: {best * 1e3:.2f} ms ({float(r.q1).hex()})")
print(os.path.basename(os.environ.get("CG_HIP_LIB", "default")), "  ".join(out), flush=True)
s.close()
