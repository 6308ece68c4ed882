"""Pure continuations (nothing read in between) behind a leading iteration that was forced into either assignment of
the two kinds of pass 2 to the buffer pairs: python tools/t_cont.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
s = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
s.init_problem(); s.solve(30); s.sync_vectors()
for forced in ("0", "1", "0", "1"):
    os.environ["CG_LEAD_DEFER_U"] = forced
    r = s.solve(3)                       # a leading iteration and two behind it
    os.environ.pop("CG_LEAD_DEFER_U")
    out = [f"forced {forced}: solve(3) {1e3 * r.seconds / 3:.3f}"]
    for k in (12, 100, 100):
        r = s.solve(k)
        out.append(f"continue {k}: {1e3 * r.seconds / k:.4f}")
    s.sync_vectors()
    print(" | ".join(out) + f" ms/it  q1={float(r.q1).hex()}", flush=True)
