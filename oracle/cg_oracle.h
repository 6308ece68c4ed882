/*
 * CPU oracle for the fp64 conjugate-gradient hot path -- TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic performed by the reference
 * program (file:line citations are to /root/reference).  It exists so that the
 * HIP kernels can be checked on a machine where the reference itself is not
 * present.  Nothing in the product path (the HIP library, the host mirror, the
 * C driver) may call into this file; only tests, the smoke check and the
 * cpu_baseline leg of the benchmark do, and only as the checker / the CPU
 * figure reported beside the GPU figure.
 *
 * Parity is PINNED: the restatement reproduces bit-for-bit the q1, sum(u) and
 * full u/g/d arrays obtained by running the reference itself (np=1) in the
 * build container; see tests/golden/ and tests/test_oracle.py.
 *
 * Arithmetic contract reproduced here (all fp64, no FMA contraction, no
 * re-association; compile with -ffp-contract=off):
 *   q[i,j]  = ((((0 + -1*d[i-1,j]) + -1*d[i,j-1]) + 4*d[i,j]) + -1*d[i,j+1]) + -1*d[i+1,j]
 *   dot     = one accumulator, strict index order
 *   u      += tau*d ; g += tau*q ; d = beta*d - g   (separate multiply and add)
 *   tau     = q0/(d.q) ; beta = q1/q0
 */
#ifndef CG_ORACLE_H
#define CG_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

/* d0 on global rows [row0, row0+rows) of the (n+1)x(n+1) grid, all columns.
 * Follows init_locald, utils.c:319-337 (formula at :334, boundary at :329-331). */
void oracle_init_d(long n, long row0, long rows, double *d);

/* g0 = -d0.  Follows init_localg, utils.c:342-350. */
void oracle_init_g(long count, const double *d, double *g);

/* q = A d on a block of `rows` full-width rows starting at global row `row0`
 * of a G x G grid.  `top`/`bot` are the rows just above / below the block
 * (NULL when the block touches the global boundary there).  Points on the
 * global boundary are not written.  Follows apply_stencil_serial
 * (utils.c:15-35) and the pad-row loops of apply_stencil_parallel
 * (utils.c:74-104); accumulation order up, left, centre, right, down. */
void oracle_stencil(long G, long row0, long rows, const double *d,
                    const double *top, const double *bot, double *q);

/* Sequential dot product, one accumulator, index order.  utils.c:368-379. */
double oracle_dot(long count, const double *v, const double *w);

/* Same, but the accumulator starts at `start` (chained row-block sums). */
double oracle_dot_from(double start, long count, const double *v, const double *w);

/* y[k] = y[k] + a*x[k].   cg.c:63-76. */
void oracle_axpy(long count, double a, const double *x, double *y);

/* d[k] = b*d[k] - g[k].   cg.c:81-87. */
void oracle_xpay(long count, double b, double *d, const double *g);

/* Whole single-rank solve as in cg.c:44-90.  Arrays are (n+1)^2 doubles and
 * are fully initialised here.  Runs at most `max_iters` iterations; when
 * `rtol` > 0 it stops after the first iteration whose q1 satisfies
 * q1 <= rtol*rtol*q_init (the reference has no stop rule; rtol = 0 gives the
 * reference's fixed count).  `q1_hist` (may be NULL) receives q1 of every
 * iteration run; `*q_init` (may be NULL) receives g0.g0.
 * Returns the number of iterations executed. */
long oracle_cg(long n, long max_iters, double rtol,
               double *u, double *g, double *d, double *q,
               double *q1_hist, double *q_init);

/* The same solve carried out on `nblocks` row blocks with explicit halo rows,
 * chained (rank-ordered, strictly sequential) dot products.  Bit-identical to
 * oracle_cg by construction; it exists to check the row-block index math. */
long oracle_cg_rowblocks(long n, long nblocks, long max_iters, double rtol,
                         double *u, double *g, double *d, double *q,
                         double *q1_hist, double *q_init);

#ifdef __cplusplus
}
#endif
/* Dot product and solve in the summation layout of the reference's own multi-rank runs:
 * s x s blocks of (G/s)^2 points, each summed sequentially in its local row-wise order
 * (utils.c:368-379), the s*s sums combined pairwise at distances 1, 2, 4, ... (the
 * MPI_Allreduce of utils.c:380 under MPICH's recursive doubling).  s*s must be a power of
 * two <= 64 and s must divide G.  Pinned by the reference's np = 4 and np = 16 outputs. */
double oracle_dot_blocks2d(long G, long s, const double *v, const double *w);
long oracle_cg_blocks2d(long n, long s, long max_iters, double rtol,
                        double *u, double *g, double *d, double *q,
                        double *q1_hist, double *q_init);

/* Coefficients used by every stencil evaluation from now on: up, left, centre, right, down
 * (entries 1, 3, 4, 5, 7 of cg.c:33's array).  Process-wide; default -1, -1, 4, -1, -1. */
void oracle_set_stencil(const int c5[5]);
/* All nine entries of the 3x3 array, row-wise (cg.c:33).  The reference's kernels never read
 * entries 0, 2, 6, 8 (utils.c:27-31): with non-zero corners the operator is this file's own
 * nine-term continuation of utils.c:25-31 in array order, and parity for the corner terms is
 * UNPINNED (no reference output exists for them).  Zero corners: exactly oracle_set_stencil. */
void oracle_set_stencil9(const int c9[9]);

#endif
