"""Do the lean passes run at the same speed on zeros and on a solve's data?  python tools/t_data.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
s = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
def times(label):
    t = [s.time_kernel(w, reps=10)[0] * 1e3 for w in (0, 1, 2)]
    print(f"{label}: pass1 {t[0]:.0f}  pass2 {t[1]:.0f}  pass2+u {t[2]:.0f} us", flush=True)
times("zeros (fresh context)")
times("zeros again")
s.init_problem(); s.solve(40)
times("after solve(40)")
times("again (data left by the timing launches)")
s.init_problem(); s.solve(41)
times("after solve(41)")
s.init_problem()
times("after init_problem only")
