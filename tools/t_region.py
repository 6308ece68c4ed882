"""Where the bench's timed region spends its wall time (strict order, K = 20): python tools/t_region.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
cg = importlib.import_module("conjugate-gradient_amd")
s = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT)
s.init_problem()
s.set_option(cg.lib.OPT_RESUM_Q0, 1)
s.solve(5); s.sync_vectors(); s.solve(20); s.sync_vectors()
for rep in range(4):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = s.solve(20)
    t1 = time.perf_counter()
    s.sync_vectors()
    t2 = time.perf_counter()
    torch.cuda.synchronize()
    t3 = time.perf_counter()
    print(f"solve call {1e3 * (t1 - t0):.3f} ms (library clock {1e3 * r.seconds:.3f}), sync_vectors {1e3 * (t2 - t1):.3f}, fence {1e3 * (t3 - t2):.3f}; total {1e3 * (t3 - t0):.3f} ms = {20 / (t3 - t0):.1f} it/s", flush=True)
s.set_option(cg.lib.OPT_RESUM_Q0, 0)
s.solve(2)
r = s.solve(20); print(f"steady {1e3 * r.seconds / 20:.4f} ms/it")
r = s.solve(200); print(f"steady200 {1e3 * r.seconds / 200:.4f} ms/it")
