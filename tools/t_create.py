"""Time cg_create at the benchmark size and the strict solve rate that follows: python tools/t_create.py [n] [iters]"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 200
t0 = time.time()
s = cg.CGSolver(n, device=0)
t1 = time.time()
s.configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
s.init_problem()
s.solve(21)
r = s.solve(iters)
print(f"create {t1 - t0:.2f} s   strict {r.seconds / r.iters * 1e3:.3f} ms/iter = {r.iters / r.seconds:.1f} it/s", flush=True)
s.close()
