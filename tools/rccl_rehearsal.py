#!/usr/bin/env python3
"""What the RCCL calls of a multi-rank iteration cost, measured with the one GPU at hand.

  python tools/rccl_rehearsal.py [n] [nranks] [iters]

One rank's row block of an n+1 grid split over `nranks` (an interior block: halo rows on both sides) runs the
real lean schedule three ways: (a) kernels only (cg_time_kernel sums), (b) with the iteration's RCCL calls
addressed to the rank itself (cg_comm_init_self: one grouped send/receive of two rows, two all-gathers of the
partial sums), launched eagerly, (c) the same captured in a hipGraph.  The solve's numbers mean nothing here
(the peers are missing); the per-iteration times do.
"""
import importlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16383
P = int(sys.argv[2]) if len(sys.argv) > 2 else 8
iters = int(sys.argv[3]) if len(sys.argv) > 3 else 400
G = n + 1
rows = G // P
print(f"block {rows} x {G} (rank 1 of {P})", flush=True)
s = cg.CGSolver(n, row0=rows, rows=rows, rank=1, nranks=P)
k = [s.time_kernel(w, reps=50)[0] for w in (0, 1, 2)]
kern = k[0] + (k[1] + k[2]) / 2
print(f"kernels only: pass1 {k[0]*1e3:.1f} us, pass2 {k[1]*1e3:.1f} us, pass2+u {k[2]*1e3:.1f} us -> {kern*1e3:.1f} us per iteration", flush=True)
per_fast = {}
for label, graph, overlap in (("eager, exchange in line", 0, False), ("eager, exchange overlapped", 0, True),
                              ("hipGraph of 10, exchange in line", 10, False), ("hipGraph of 10, exchange overlapped", 10, True)):
    t = cg.CGSolver(n, row0=rows, rows=rows, rank=1, nranks=P)
    t.comm_init_self()
    t.configure(graph_iters=graph, overlap=overlap).init_problem()
    t.solve(21)                                            # warm-up: RCCL's first calls, graph capture
    r = t.solve(iters)
    per = 1e3 * r.seconds / max(r.iters, 1)               # ms, like the kernel times
    per_fast[(graph, overlap)] = per
    print(f"{label:38s}: {per*1e3:8.1f} us per iteration  (+{(per - kern)*1e3:6.1f} us over the kernels, "
          f"{100 * kern / per:5.1f} % of the time is kernels)", flush=True)
    t.close()
# The reference's summation order on several ranks: the accumulator of each dot product has to pass through the ranks' row
# blocks.  Its cost depends on the data (binade crossings, hypotheses), so this part runs a WHOLE grid of about a rank's
# share of points (n2 + 1 = 6144: 1.1 x the 2048 x 16384 block) on a single-rank context that issues the multi-rank
# traffic to itself (CG_OPT_COMM_REHEARSAL = P): real numbers, real crossings.  "published": one all-gather of the
# approximate totals, one of the plans' briefs, every rank finishes the walk; "in turn": one all-gather per rank and dot
# product.  RCCL calls per iteration: halo 1 + dot products 2 x 2 (published) or 2 x (P + 1) (in turn); fast order 1 + 2.
n2 = 6143
rate = {}
for label, red, reh, chain, graph in (("fast order, no traffic", cg.REDUCE_FAST, 0, 1, 10), ("fast order, rehearsed", cg.REDUCE_FAST, P, 1, 10),
                                      ("strict, no traffic", cg.REDUCE_STRICT, 0, 1, 10),
                                      ("strict, published briefs, hipGraph", cg.REDUCE_STRICT, P, 1, 10),
                                      ("strict, published briefs, eager", cg.REDUCE_STRICT, P, 1, 0),
                                      ("strict, ranks in turn, hipGraph", cg.REDUCE_STRICT, P, 0, 10)):
    if reh and graph and red == cg.REDUCE_STRICT and os.environ.get("CG_REHEARSAL_EAGER"):
        continue                                           # (rocprofv3's kernel trace crashes on these captured graphs; the eager flavour is traced instead)
    t = cg.CGSolver(n2)
    if reh:
        t.comm_init(cg.CGSolver.comm_unique_id())
    t.configure(reduce=red, graph_iters=graph, overlap=False, comm_rehearsal=reh)
    t.set_option(cg.lib.OPT_STRICT_CHAIN, chain)
    t.init_problem()
    t.solve(41)
    r = t.solve(iters)
    per = 1e6 * r.seconds / max(r.iters, 1)
    rate[label] = per
    calls = 0 if not reh else (3 if red == cg.REDUCE_FAST else 1 + 2 * (2 if chain else P + 1))
    print(f"{n2 + 1}^2 {label:36s}: {per:8.1f} us per iteration, RCCL calls per iteration: {calls}, q1={float(r.q1).hex()}", flush=True)
    t.close()
pub = rate.get("strict, published briefs, hipGraph", rate["strict, published briefs, eager"])
print(f"strict (published briefs) / fast order, both with the traffic of {P} ranks: {pub / rate['fast order, rehearsed']:.2f}", flush=True)
