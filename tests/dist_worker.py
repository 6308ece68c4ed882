"""One rank of a multi-process row-block solve (started by test_dist_gloo.py / test_gpu_multirank.py).

  python dist_worker.py <mode> <n> <iters> <outdir> [schedule] [rtol] [strict|strict_replay|strict_parallel] [digest]

"digest" stores the SHA-256 of each vector's bytes instead of the vector (grids too large to write out).

mode "oracle": the compute is the CPU oracle (checker code), the communication is the product's
               RowTransport over gloo -- this pins the transport's semantics without a GPU.
mode "gpu":    the compute is the HIP library (one context per rank, all ranks on GPU 0), halo rows
               and partial sums travel through the same RowTransport via cg_comm_init_host.
Rank/world/rendezvous come from RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT.
"""
import hashlib
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))


def oracle_rank(tr, dist_mod, n, iters, order):
    """Row-block CG on the oracle's operators; `order` = "chained" (np=1 summation order) or "ranked"."""
    import oracle
    G = n + 1
    row0, rows = dist_mod.partition(G, tr.world, tr.rank)
    d = oracle.init_d(n, row0, rows)
    g = oracle.init_g(d)
    u = np.zeros_like(d)
    q = np.zeros_like(d)

    def dot(v, w):
        if order == "chained":                 # one accumulator handed from rank to rank
            acc = 0.0
            for r in range(tr.world):
                mine = oracle.dot(v, w, start=acc) if r == tr.rank else 0.0
                acc = tr.allgather(mine)[r]
            return acc
        parts = tr.allgather(oracle.dot(v, w))   # per-rank sums added in rank order
        acc = parts[0]
        for p in parts[1:]:
            acc = acc + p
        return acc

    q0 = dot(g, g)
    hist = []
    for _ in range(iters):
        top, bot = tr.exchange(d[0], d[-1])
        oracle.stencil(d, row0=row0, G=G, top=top, bot=bot, q=q)
        tau = q0 / dot(d, q)
        oracle.axpy(tau, d, u)
        oracle.axpy(tau, q, g)
        q1 = dot(g, g)
        oracle.xpay(q1 / q0, d, g)
        q0 = q1
        hist.append(q1)
    return dict(u=u, g=g, d=d, hist=np.array(hist), row0=row0)


def gpu_rank(tr, cg, n, iters, schedule, rtol, order="fast", digest=False):
    G = n + 1
    row0, rows = cg.dist.partition(G, tr.world, tr.rank)
    s = cg.CGSolver(n, device=0, row0=row0, rows=rows, rank=tr.rank, nranks=tr.world)
    ids = {"classic": cg.SCHED_CLASSIC, "fused": cg.SCHED_FUSED, "fused_noq": cg.SCHED_FUSED_NOQ, "lean": cg.SCHED_LEAN}
    schedule, _, second = schedule.partition(">")    # "a>b": the first solve runs schedule a, the continuation schedule b
    s.configure(schedule=ids[schedule])
    if order != "fast":                           # one accumulator handed from rank to rank
        s.configure(reduce=cg.REDUCE_STRICT, schedule=s.schedule,
                    strict_impl={"strict": cg.STRICT_AUTO, "strict_replay": cg.STRICT_REPLAY,
                                 "strict_parallel": cg.STRICT_PARALLEL}[order])
    s.configure(overlap=os.environ.get("CG_TEST_OVERLAP", "1") != "0")   # the split second pass (edge strips first) unless told otherwise
    if "CG_TEST_STRICT_CHAIN" in os.environ:      # 0: the ranks walk in turn; 1 (the default): published briefs
        s.set_option(cg.lib.OPT_STRICT_CHAIN, int(os.environ["CG_TEST_STRICT_CHAIN"]))
    if "CG_TEST_FAULT_EVERY" in os.environ:       # test hook: every n-th dot product is completed by the ranks walking in turn
        s.set_option(cg.lib.OPT_STRICT_FAULT_EVERY, int(os.environ["CG_TEST_FAULT_EVERY"]))
    s.comm_init_host(tr)
    s.comm_selftest()                             # tagged all-gather and row exchange arrive where they should
    s.init_problem()
    if rtol > 0:
        r = s.solve(iters, rtol)
        total, hist, q_init = r.iters, s.history(iters), r.q_init
    else:                                         # two calls: the second continues the first
        first = s.solve(iters // 2)
        h1 = s.history(iters)
        if second:
            s.configure(schedule=ids[second])
        r = s.solve(iters - iters // 2)
        total, hist, q_init = first.iters + r.iters, np.concatenate([h1, s.history(iters)]), first.q_init
    out = dict(row0=row0, hist=hist, q1=r.q1, iters=total, converged=int(r.converged), q_init=q_init,
               faults=s.get_option(cg.lib.OPT_STRICT_FAULTS))
    for key, which in (("u", cg.VEC_U), ("g", cg.VEC_G), ("d", cg.VEC_D)):
        v = s.vector(which)
        out[key] = np.array(hashlib.sha256(v).hexdigest()) if digest else v
    s.close()
    return out


def main():
    mode, n, iters, outdir = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    extra = sys.argv[5:]
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=int(os.environ["RANK"]), world_size=int(os.environ["WORLD_SIZE"]))
    cg = importlib.import_module("conjugate-gradient_amd")
    from importlib import import_module
    cg.dist = import_module("conjugate-gradient_amd.dist")
    tr = cg.dist.RowTransport()
    if mode == "oracle":
        res = oracle_rank(tr, cg.dist, n, iters, extra[0] if extra else "chained")
    else:
        res = gpu_rank(tr, cg, n, iters, extra[0] if extra else "fused_noq", float(extra[1]) if len(extra) > 1 else 0.0,
                       extra[2] if len(extra) > 2 else "fast", len(extra) > 3 and extra[3] == "digest")
    np.savez(os.path.join(outdir, f"rank{tr.rank}.npz"), **res)
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
