"""How much of a cold solve's wall time is set-up inside cg_solve (workspace, history, graph capture)?  python tools/t_setup.py"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
cg = importlib.import_module("conjugate-gradient_amd")
for order in (cg.REDUCE_STRICT, cg.REDUCE_FAST):
    s = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=order)
    for rep in range(2):
        s.init_problem()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = s.solve(1000)
        t1 = time.perf_counter()
        s.sync_vectors()
        t2 = time.perf_counter()
        print(f"order {order} rep {rep}: solve call {t1 - t0:.4f} s (library clock {r.seconds:.4f}), sync_vectors {1e3 * (t2 - t1):.2f} ms -> {1000 / (t2 - t0):.1f} it/s", flush=True)
    s.close()
