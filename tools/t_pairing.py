"""Both pairings of the u updates from both start directions, timed over long solves: python tools/t_pairing.py
(CG_DEBUG_PLACEMENT=1 prints the direction each solve starts from and the library's own estimate)"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
s = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
s.init_problem(); s.solve(30); s.sync_vectors()
def run(defer, k):
    os.environ["CG_LEAD_DEFER_U"] = defer
    r = s.solve(k)
    s.sync_vectors()
    print(f"defer={defer} k={k}: {1e3 * r.seconds / k:.4f} ms/it", flush=True)
for rep in range(2):
    run("0", 101); run("1", 101)      # 100 iterations behind the leading one: the next solve starts from the same direction
    run("0", 102)                     # 101: the next one from the other
    run("0", 101); run("1", 101)
    run("1", 102)
