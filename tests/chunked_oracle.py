"""First CG iteration of the oracle, evaluated in row chunks so that grids far larger than host memory
can be checked: the dot products are chained through the chunks with one accumulator (index order, as
utils.c:372-379), everything else is local to a chunk and its two neighbouring rows."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np


def first_iteration(oracle, n, chunk_rows, sample_rows, workers=8):
    """Returns dict(q0, dq, tau, q1, beta, rows={i: (u1[i], g1[i], d1[i])}) of cg.c:55-89 for iteration 0."""
    G = n + 1
    assert G % chunk_rows == 0
    nchunks = G // chunk_rows

    def chunk(k):                                      # d0 and q = A d0 on rows [k*CH, (k+1)*CH)
        lo = k * chunk_rows
        has_top, has_bot = lo > 0, lo + chunk_rows < G
        dd = oracle.init_d(n, lo - (1 if has_top else 0), chunk_rows + has_top + has_bot)
        top = dd[0] if has_top else None
        bot = dd[-1] if has_bot else None
        blk = np.ascontiguousarray(dd[(1 if has_top else 0):(len(dd) - (1 if has_bot else 0))])
        return blk, oracle.stencil(blk, row0=lo, G=G, top=top, bot=bot)

    q0 = dq = 0.0
    with ThreadPoolExecutor(max_workers=workers) as pool:
        for blk, q in pool.map(chunk, range(nchunks)):
            q0 = oracle.dot(blk, blk, start=q0)        # g0 = -d0: the products are the same doubles
            dq = oracle.dot(blk, q, start=dq)
    tau = q0 / dq

    def chunk2(k):
        blk, q = chunk(k)
        u = np.zeros_like(blk)
        oracle.axpy(tau, blk, u)
        g = oracle.init_g(blk)
        oracle.axpy(tau, q, g)
        return k, blk, u, g

    q1 = 0.0
    kept = {}
    with ThreadPoolExecutor(max_workers=workers) as pool:
        for k, blk, u, g in pool.map(chunk2, range(nchunks)):
            q1 = oracle.dot(g, g, start=q1)
            for i in sample_rows:
                if k * chunk_rows <= i < (k + 1) * chunk_rows:
                    j = i - k * chunk_rows
                    kept[i] = (u[j].copy(), g[j].copy(), blk[j].copy())
    beta = q1 / q0
    rows = {}
    for i, (u1, g1, d0) in kept.items():
        d1 = d0.copy()
        oracle.xpay(beta, d1, g1)                      # d1 = beta*d0 - g1
        rows[i] = (u1, g1, d1)
    return dict(q0=q0, dq=dq, tau=tau, q1=q1, beta=beta, rows=rows)
