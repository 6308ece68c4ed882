"""Two independent strict evaluations against each other on the benchmark grid, far beyond the golden's 1000 iterations:
in-pass summaries with restarts from materialised vectors (every 250 iterations u is read) against stored products in one
go.  Every q1 of the history and the final u, g, d must agree bit for bit.  python tools/soak.py [iters]"""
import hashlib, importlib, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
cg = importlib.import_module("conjugate-gradient_amd")
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 3000
a = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT, strict_spec=True).init_problem()
ha, done = [], 0
while done < iters:
    k = min(250 if (done // 250) % 2 == 0 else 249, iters - done)
    r = a.solve(k); ha.append(a.history(k)); done += k
    a.vector_rows(cg.VEC_U, 0, 1)                     # materialises: the next solve restarts with a leading iteration
ha = np.concatenate(ha)
da = {w: hashlib.sha256(a.vector(w)).hexdigest() for w in (cg.VEC_U, cg.VEC_G, cg.VEC_D)}
print(f"in-pass summaries, restarts: q1={float(r.q1).hex()}", flush=True)
a.close()
b = cg.CGSolver(16383).configure(schedule=cg.SCHED_LEAN, reduce=cg.REDUCE_STRICT, strict_spec=False).init_problem()
r = b.solve(iters)
hb = b.history(iters)
db = {w: hashlib.sha256(b.vector(w)).hexdigest() for w in (cg.VEC_U, cg.VEC_G, cg.VEC_D)}
print(f"stored products, one go:     q1={float(r.q1).hex()}  ({r.iters / r.seconds:.1f} it/s)", flush=True)
same = np.array_equal(ha.view(np.uint64), hb.view(np.uint64))
print("history bit-identical:", same, " vectors:", da == db, flush=True)
sys.exit(0 if same and da == db else 1)
