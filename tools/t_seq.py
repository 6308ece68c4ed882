"""The bench's sequence of solves in strict order, each timed by the library, with the repair count behind each: python tools/t_seq.py"""
import importlib, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = 16383
s = cg.CGSolver(n).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
s.init_problem()
s.set_option(cg.lib.OPT_RESUM_Q0, 1)
out = []
def step(label, k, sync):
    r = s.solve(k)
    m = s.get_option(cg.lib.OPT_STRICT_MISSES)
    ev = s.get_option(cg.lib.OPT_STRICT_EVENTS)
    if sync:
        s.sync_vectors()
    out.append(f"{label} {r.seconds / r.iters * 1e3:.3f} ms/it (redone {m}, apart {ev // 1000000}+{ev % 1000000})")
step("warm5", 5, True); step("reh20", 20, True); step("timed20", 20, True)
s.set_option(cg.lib.OPT_RESUM_Q0, 0)
step("two", 2, False); step("steady20", 20, False); step("steady20b", 20, False); step("steady100", 100, False)
print(os.path.basename(os.environ.get("CG_HIP_LIB", "default")), " | ".join(out), flush=True)
s.close()
