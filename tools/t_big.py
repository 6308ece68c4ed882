"""Larger grids on one GPU, both orders: python tools/t_big.py n iters"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n, iters = int(sys.argv[1]), int(sys.argv[2])
t0 = time.time()
s = cg.CGSolver(n)
print(f"n={n}: create {time.time() - t0:.1f} s", flush=True)
for red, name in ((cg.REDUCE_STRICT, "strict"), (cg.REDUCE_FAST, "fast")):
    s.configure(schedule=cg.SCHED_LEAN, reduce=red).init_problem()
    s.solve(int(os.environ.get("T_BIG_WARM", "12")))
    r = s.solve(iters)
    print(f"  {name}: {1e3 * r.seconds / r.iters:.3f} ms/it = {(n + 1) ** 2 * 56 / (r.seconds / r.iters) / 1e12:.2f} TB/s  q1={float(r.q1).hex()}", flush=True)
s.close()
