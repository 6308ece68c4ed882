import importlib, sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
for n in (5791, 4095):
    for g in (0, 10, 0, 10):
        s = cg.CGSolver(n).configure(graph_iters=g).init_problem()
        s.solve(50); r = s.solve(1000)
        print(f"n={n} graph_iters={g}: {r.seconds:.4f} s  {1000/r.seconds:.0f} it/s  {r.seconds:.6f}", flush=True)
        s.close()
