import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
cg = importlib.import_module("conjugate-gradient_amd")
n = 16383
for pfr in (0, 4, 2):
    s = cg.CGSolver(n).configure(reduce=cg.REDUCE_STRICT, prefetch_read=pfr).init_problem()
    s.solve(4)
    t = [s.time_kernel(w, reps=20)[0] * 1e3 for w in (0, 1, 2)]
    s.init_problem(); s.solve(21)
    r = s.solve(200)
    print(f"prefetch_read={pfr}: pass1 {t[0]:.0f} pass2 {t[1]:.0f} pass2+u {t[2]:.0f} us; solve {1e3*r.seconds/r.iters:.3f} ms/iter q1={float(r.q1).hex()}", flush=True)
    s.close()
