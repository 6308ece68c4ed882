"""One cold start in strict order for a kernel trace: python tools/t_cold1.py [n] [iters]"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
k = int(sys.argv[2]) if len(sys.argv) > 2 else 31
s = cg.CGSolver(n).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
s.init_problem(); s.solve(30)                 # graphs, workspace
s.init_problem()
r = s.solve(k)
print(f"cold solve({k}): {r.seconds * 1e3:.2f} ms", flush=True)
s.close()
