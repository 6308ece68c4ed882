/*
 * C ABI of the MI355X (gfx950) conjugate-gradient hot path: libcg_hip.so.
 *
 * Plain C linkage, plain pointers and sizes.  Every `double *` that is not
 * explicitly called a host pointer is a DEVICE pointer (HBM of the GPU the
 * context or the call is bound to).  Every function returns 0 on success and a
 * negative code on failure; cg_last_error() gives the text.  Nothing here falls
 * back to the CPU: without a GPU every compute entry point fails.
 *
 * Each entry point cites the piece of the reference (omarmalik710/
 * conjugate-gradient, files cg.c / utils.c / utils.h) that it replaces.  The
 * reference keeps the vectors in host memory and runs the update loops inline
 * in main(); here the vectors live in HBM, so everything that touches them --
 * allocation, the three update loops, the scalar recurrences, the read-back --
 * sits behind this interface.  include/utils.h re-exports the reference's own
 * nine names on top of it.
 *
 * Threads: like the reference (one thread per rank, MPI_Init at cg.c:11) the
 * library expects one calling thread per GPU.  The stand-alone operators of one
 * device share that device's internal stream and scratch buffers, and a context
 * is a single stream of work: calls on the same device or context from several
 * threads at once must be serialised by the caller.  Different devices and
 * different contexts are independent.
 */
#ifndef CG_HIP_H
#define CG_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

/* ---- status ------------------------------------------------------------- */
#define CG_OK              0
#define CG_ERR_HIP        -1   /* a HIP runtime call failed                      */
#define CG_ERR_ARG        -2   /* bad argument (shape, alignment, null pointer)  */
#define CG_ERR_STATE      -3   /* call not valid in the context's current state  */
#define CG_ERR_NOGPU      -4   /* no usable gfx950 device                        */
#define CG_ERR_COMM       -5   /* RCCL could not be loaded or a call failed      */

const char *cg_last_error(void);          /* thread-local text of the last failure  */
int cg_device_count(int *count);          /* number of visible HIP devices          */

/* ---- raw device memory, so that C callers need not link HIP --------------
 * Replaces the malloc/calloc/free of the vectors in main (cg.c:44-45,143-145)
 * and in init_locald / init_localg (utils.c:278,343).                        */
int cg_malloc(int device, long count, double **out);     /* count doubles, zero-filled (calloc, cg.c:44-45) */
int cg_free(int device, double *p);
int cg_copy_to_host(int device, double *host_dst, const double *dev_src, long count);
int cg_copy_to_device(int device, double *dev_dst, const double *host_src, long count);

/* ---- stand-alone operators on caller-owned device arrays -----------------
 * All run on `device`'s internal stream and are complete when they return.  */

/* d0 for global rows [row0, row0+rows) of the (n+1)^2 grid, all n+1 columns.
 * Replaces the fill loop of init_locald, utils.c:319-337.                    */
int cg_init_d(int device, long n, long row0, long rows, double *d);
/* g0 = -d0.  Replaces init_localg, utils.c:342-350.                          */
int cg_init_g(int device, long count, const double *d, double *g);
/* q = A d on a block of `rows` full rows of a G x G grid that starts at
 * global row row0; top/bot = the rows just outside the block (device
 * pointers, NULL at the global boundary).  Global boundary points of q are
 * not written.  Replaces apply_stencil_serial (utils.c:8-36) and the row part
 * of apply_stencil_parallel (utils.c:38-104).  If dq_host != NULL it also
 * returns d.q accumulated in the fast order (fused, one pass).               */
int cg_stencil(int device, long G, long row0, long rows, const double *d,
               const double *top, const double *bot, double *q, double *dq_host);
/* The same with the coefficients of a stencil_struct (utils.h:3-7): stencil9 is the 3x3 integer
 * array, row-wise (host pointer); cg_stencil uses cg.c:33's {0,-1,0,-1,4,-1,0,-1,0}.  The reference's
 * kernels read entries 1, 3, 4, 5, 7 -- up, left, centre, right, down -- only (utils.c:27-31) although
 * the struct and its unused corner_pad (utils.h:15) are laid out for all nine.  Here all nine are
 * honoured: with non-zero corner entries the sum of utils.c:25-31 is continued over the whole array in
 * array order (row above left to right, own row, row below); with zero corners the five-term code runs
 * and the result is the reference's bit for bit. */
int cg_stencil9(int device, const int *stencil9, long G, long row0, long rows, const double *d,
                const double *top, const double *bot, double *q, double *dq_host);
/* *result_host = sum v[k]*w[k], k = 0..count-1.  strict != 0: one accumulator in
 * index order starting from `start` (bit-identical to utils.c:372-379);
 * strict = 0: the documented fast order (start is added last).
 * The strict order has two evaluations that return the same bits for any input:
 * strict = 2 replays the recurrence literally on one wavefront (one dependent
 * add per element); strict = 3 evaluates it in parallel (inside a binade the
 * recurrence adds integers, which is associative; binade crossings, ties and
 * the leading zeros fall back to the literal replay, a chunk at a time);
 * strict = 1 picks the parallel one from 32768 elements up.
 * Replaces the local part of dot, utils.c:368-379.                           */
int cg_dot(int device, long count, const double *v, const double *w, int strict,
           double start, double *result_host);
/* ---- the reference's own two-dimensional decomposition (sqrt(P) x sqrt(P) chunks, utils.c:408-417) ----
 * Operators on one rank's rows x cols chunk, stored row-major like the reference's local arrays.  The fast solve
 * (cg_solve) uses row blocks; these serve the call-by-call path of include/utils.h, which with them runs the
 * reference's decomposition as it stands (and, with strict dot products and MPI_Allreduce, gives the bits of
 * `mpirun -np P ./cg n`).
 * d0 on the chunk whose first point is global (row0, col0): init_locald's fill loop (utils.c:319-337).      */
int cg_init_d_block(int device, long n, long row0, long col0, long rows, long cols, double *d);
/* q = A d on the chunk: apply_stencil_parallel's interior (utils.c:50), edge rows (:74-104), edge columns
 * (:121-151) and corner points (:156-218).  top / bot: the `cols` values just above / below the chunk, left /
 * right: the `rows` values beside it (device pointers); NULL = the chunk touches the global boundary there and q
 * is not written on that edge.  Corner coefficients need cg_stencil_block9 where the chunk has a diagonal neighbour. */
int cg_stencil_block(int device, const int *stencil9, long rows, long cols, const double *d, const double *top,
                     const double *bot, const double *left, const double *right, double *q);
/* The same with all nine coefficients: corners[0..3] (device pointer) = the values diagonally across the chunk's
 * corners -- up-left, up-right, down-left, down-right -- which the nine-point sum needs at the chunk's four corner
 * points.  This is what d_struct::corner_pad (utils.h:15) is laid out for; the reference never fills or reads it
 * (utils.c:156-218 handles the corner points with two pads each, five coefficients).  An entry is read only where
 * both adjacent pads exist; corners may be NULL for five-point stencils and for chunks without diagonal neighbours. */
int cg_stencil_block9(int device, const int *stencil9, long rows, long cols, const double *d, const double *top,
                      const double *bot, const double *left, const double *right, const double *corners, double *q);
/* dst[r] = src[r*cols + col], r = 0..rows-1: the sending side of the reference's strided column datatype
 * (MPI_Type_vector, utils.c:392; used at utils.c:261,270).                                                   */
int cg_pack_column(int device, long rows, long cols, const double *src, long col, double *dst);

/* Checks of a solution block u (global rows [row0, row0+rows) of the (n+1)^2 grid) on the device -- the
 * reference's only inspection hook is the array dump print_local2dmesh (utils.c:352-366), whose one call
 * site is commented out (cg.c:135):
 *   *sum_u_host   = sum_start + u[0] + u[1] + ... added one by one in index order (the order of
 *                   utils.c:372-379); ranks owning consecutive row blocks pass the value on as
 *                   sum_start, which gives the single-rank bits;
 *   *max_err_host = max |u - x(1-x) y(1-y)| over the block, x = i/n, y = j/n: the distance to the
 *                   closed-form solution of the problem init_locald sets up (utils.c:334).
 * Either output pointer may be NULL.                                         */
int cg_verify_u(int device, long n, long row0, long rows, const double *u, double sum_start,
                double *sum_u_host, double *max_err_host);
/* y[k] += a*x[k].  Replaces the loops at cg.c:63-76.                         */
int cg_axpy(int device, long count, double a, const double *x, double *y);
/* d[k] = b*d[k] - g[k].  Replaces the loop at cg.c:81-87.                    */
int cg_xpay(int device, long count, double b, double *d, const double *g);

/* cg_stencil for callers that cannot allocate (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  It has no counterpart in the reference.  The same kernels, with
 * their scratch -- one partial sum per tile -- in a fixed device array of 512 KiB that every GPU the
 * library runs on holds until the process ends, instead of library-allocated memory.  It allocates
 * nothing, creates no stream and does not synchronise: the work is enqueued on `stream` (a hipStream_t;
 * NULL = the null stream) and is complete when that stream reaches it.  dq_dev (DEVICE pointer, may be
 * NULL) receives d.q in the fast order, the value cg_stencil returns through dq_host.  A block is cut into
 * ceil(G / 512) column strips times row tiles of about 512 rows (never fewer than 16); blocks that need more
 * than 65536 tiles -- some 1.7e10 points -- are rejected with CG_ERR_ARG.  All calls on one GPU share the array:
 * calls that overlap in time on the same GPU race on it.  The array's address on a device is looked up by the
 * first library call that touches the device (this one included); that first call must not be made under stream
 * capture (CG_ERR_STATE), every later one may be.  The caller's current device is left as it was.             */
int cg_stencil_noalloc(int device, long G, long row0, long rows, const double *d, const double *top,
                       const double *bot, double *q, double *dq_dev, void *stream);

/* ---- whole-solve context -------------------------------------------------
 * Replaces the body of main between set-up and output (cg.c:41-134).         */
typedef struct cg_ctx cg_ctx;

/* Reduction order of the dot products. */
#define CG_REDUCE_FAST    0    /* wavefront shuffle -> workgroup -> fixed-order grid sum */
#define CG_REDUCE_STRICT  1    /* one accumulator, index order: bit-identical to the reference on one
                                  rank (utils.c:372-379); with several ranks the accumulator is handed
                                  from rank to rank, so P row blocks still give the single-rank bits */
/* Kernel schedule of one iteration (all produce bit-identical vectors). */
#define CG_SCHED_CLASSIC  0    /* stencil+d.q | u,g update+g.g | d update        : 88 B/point */
#define CG_SCHED_FUSED    1    /* d update fused into the stencil kernel         : 80 B/point */
#define CG_SCHED_FUSED_NOQ 2   /* ... and q recomputed in the update kernel      : 64 B/point
                                  (strict order: the rounded products d*q are stored where q would go) */
#define CG_SCHED_LEAN     3    /* d' = beta*d - g recomputed in both passes instead of stored and re-read,
                                  u updated for two iterations at a time           : 56 B/point
                                  (needs a second gradient buffer; strict order: + 8 for the stored products d*q) */

/* Option keys for cg_set_option. */
#define CG_OPT_REDUCE      1   /* CG_REDUCE_*                        default FAST     */
#define CG_OPT_SCHEDULE    2   /* CG_SCHED_*                         default LEAN */
#define CG_OPT_TILE_H      3   /* rows per stencil tile, 0 = auto                     */
#define CG_OPT_GRAPH_ITERS 4   /* iterations captured per hipGraph, 0 = no graph; default 10 */
#define CG_OPT_XCD_REMAP   5   /* 1 = XCD-aware tile order (default), 0 = plain order  */
#define CG_OPT_FLAT_BLOCKS 6   /* workgroups of the flat (1-D) kernels, 0 = auto       */
#define CG_OPT_PREFETCH    7   /* rows of loads each thread keeps in flight: 1, 2, 4 or 8 for every tiled
                                  kernel; 0 = measured per-kernel defaults (default) */
#define CG_OPT_STRICT_IMPL 8   /* how the strict order is evaluated: 1 = auto (default), 2 = literal
                                  one-accumulator replay, 3 = parallel evaluation (same bits, see cg_dot) */
#define CG_OPT_PREFETCH_READ 11 /* same for LEAN's read-only pass alone; 0 = default (4; 2 when it also forms the span sums
                                  of CG_OPT_STRICT_SPEC, which then runs three workgroups per CU) */
#define CG_OPT_STRICT_BLOCKS 12 /* with CG_REDUCE_STRICT on one GPU: s = 2, 4 or 8 sums every dot product in the
                                  layout of the reference run on s*s MPI ranks -- each rank's (G/s)^2 block
                                  sequentially in its local row-wise order (utils.c:372-379, blocks as in
                                  utils.c:408-417), the s*s sums combined as MPICH's MPI_Allreduce combines
                                  them (utils.c:380) -- and so reproduces the bits of `mpirun -np s*s ./cg`.
                                  0 or 1 = the single-rank order (default) */
#define CG_OPT_STRICT_SPEC 15  /* strict order, lean schedule, rows of a multiple of 1024 points: 1 (default) =
                                  the passes summarise their products for the sequential sum while they form them, under
                                  two hypothesised exponents per span (from the previous iteration's prefix sums); spans
                                  whose hypotheses miss are redone.  0 = store the products and read them again.  Same bits. */
#define CG_OPT_STRICT_MISSES 16 /* read only (cg_get_option): chunks of 1024 products whose hypotheses missed in the last
                                  dot product of the last solve and were redone (diagnostic of CG_OPT_STRICT_SPEC) */
#define CG_OPT_STRICT_EVENTS 17 /* read only: chunks the last solve's final two ordered walks had to take apart element by
                                  element (binade crossings, half-way cases): 10^6 * (d.q walk) + (g.g walk) */
#define CG_OPT_STRICT_AUDIT 18 /* self-test of CG_OPT_STRICT_SPEC (slow): 1 = after every dot product each chunk summary is
                                  checked against one recomputed from the products, and the plan-driven walk against the
                                  plain hierarchical one; cg_get_option returns the failures counted so far (must be 0) */
#define CG_OPT_OVERLAP     13  /* several ranks, lean schedule: 1 = the new gradient's boundary rows travel on a second stream
                                  while the rows between them are computed; 0 (default) = exchange in line.  Which is faster
                                  depends on the link; time both (bench.py does) */
#define CG_OPT_RESUM_Q0    19  /* 1 = a cg_solve that continues the previous one (see cg_solve) still begins with the dot
                                  product q0 = g.g of cg.c:55 instead of taking q1 over: the reference's timed region
                                  (cg.c:50-55) with warm vectors.  Strict order: the same bits either way.  Default 0 */
#define CG_OPT_STRICT_CHAIN 20 /* several ranks, CG_REDUCE_STRICT, lean schedule: how the one accumulator passes through the ranks'
                                  row blocks.  1 (default) = every rank publishes the briefs of its plan (what a walk over its
                                  block needs when nothing unforeseen happens) in one all-gather, rank 0 walks its own block, and
                                  every rank finishes the walk from there: two collectives per dot product whatever the rank
                                  count; a dot product for which a brief does not hold is completed by the ranks walking in turn
                                  (cg_solve notices, nothing is lost).  0 = the ranks always walk in turn (one collective per rank
                                  and dot product).  Same bits */
#define CG_OPT_STRICT_FAULT_EVERY 21 /* test hook for CG_OPT_STRICT_CHAIN: n > 0 declares every n-th dot product (counted as
                                  2 * iteration + {0: d.q, 1: g.g}) failed on the published walk, so that the completion by
                                  ranks walking in turn runs; 0 (default) = off */
#define CG_OPT_STRICT_FAULTS 22 /* read only: dot products the context has completed that way so far */
#define CG_OPT_COMM_REHEARSAL 14 /* single-rank context with cg_comm_init: P >= 1 = issue the per-iteration traffic of a
                                  solve on P ranks anyway (a grouped send/receive of two rows to the rank itself, the
                                  all-gathers of the dot products -- in the strict order with the ranks walking in turn, one
                                  per rank), to put a number on its cost with one GPU.  The results are not affected */
#define CG_OPT_TILE_WGS    9   /* cap on workgroups per tiled launch; with more tiles than that each
                                  workgroup walks several tiles in turn.  0 = auto (256 with automatic
                                  tile height, one workgroup per tile with an explicit CG_OPT_TILE_H) */

/* A context owns the row block [row0, row0+rows) x (n+1) of u, g, d, q on
 * `device`.  Single GPU: row0 = 0, rows = n+1, rank 0 of 1.
 * Blocks of 2^27 points or more: the vectors are allocated up to eight times over, the passes of the default schedule
 * timed on each set (in both directions of the buffer pairs), the fastest placement kept, single vectors of it traded
 * against those of the other sets where that makes the passes faster, and the rest released (1.5-2 s, once; where the
 * driver puts a vector in HBM changes the passes' speed by 5 % and more: which L2 sets the streams of a pass compete
 * for).  In the environment, CG_PLACEMENT_REFINE=0 skips the trades (0.35 s) and CG_PLACEMENT_DRAWS=1 the whole search. */
int cg_create(cg_ctx **out, int device, long n, long row0, long rows, int rank, int nranks);
int cg_destroy(cg_ctx *ctx);
int cg_set_option(cg_ctx *ctx, int key, long value);
/* Coefficients of the context's stencil: the 3x3 integer array of a stencil_struct (cg.c:33-38), all nine
 * entries (see cg_stencil9; the reference reads entries 1, 3, 4, 5, 7 only).  Default: cg.c:33's values. */
int cg_set_stencil(cg_ctx *ctx, const int *stencil9);
/* Current value of an option; for CG_OPT_SCHEDULE, CG_OPT_TILE_H and CG_OPT_TILE_WGS the value in
 * effect after automatic selection. */
int cg_get_option(cg_ctx *ctx, int key, long *value);

/* u = 0, q = 0, d = d0, g = -d0 (cg.c:41-45 via utils.c:276-350).            */
int cg_init_problem(cg_ctx *ctx);

typedef struct cg_result {
    long iters;          /* iterations executed                                  */
    double q1;           /* last g.g: the reference's "norm_g^2" (cg.c:139)      */
    double q_init;       /* g0.g0 (cg.c:55)                                      */
    double seconds;      /* wall time of the reference's timed region
                            (cg.c:50-51,131-132): initial g.g + all iterations,
                            device-synchronised at both ends                     */
    int converged;       /* 1 if the stop rule fired                             */
} cg_result;

/* Run at most max_iters iterations from the context's current state (call
 * cg_init_problem first).  rtol > 0 adds the stop rule q1 <= rtol^2 * q_init,
 * evaluated on the device after every iteration; the vectors are left exactly
 * as they were when it fired.  rtol = 0 is the reference's fixed count.
 * A solve that follows another one on the same context continues it: q0 is the
 * q1 the last iteration left on the device (the same g.g, cg.c:55 and :89), so
 * K1 + K2 iterations in two calls give the bits of K1 + K2 in one.
 * The fused and lean schedules fold `d = beta*d - g` (cg.c:81-87) into the next
 * iteration's kernels, the lean one also every other `u += tau*d` (cg.c:63-69);
 * when a solve returns, the last of these updates may still be owed.  They are
 * applied by the next cg_solve (as part of its first iteration) or by whatever
 * looks at the vectors first: cg_copy_out, cg_device_ptr, cg_sync_vectors,
 * cg_set_option, cg_set_stencil.  With several ranks that step exchanges halo
 * rows, so those calls are collective there: make them on every rank.
 * Lean schedule: the two kinds of second pass alternate, so each of them always reads the same one of the two
 * buffer pairs, and the pairs can differ in speed (cg_create).  Which pass gets which is settled by the first
 * iteration of a solve that starts from applied updates, from the count of iterations behind it; a solve that
 * merely continues another one inherits it, and, on one rank, where cg_create measured the passes, applies the owed
 * updates and starts over when the other assignment would pay within max_iters.  `seconds` does not count that step. */
int cg_solve(cg_ctx *ctx, long max_iters, double rtol, cg_result *out);
/* Apply the updates the last solve left owed (see cg_solve): afterwards u, g and d hold
 * what the reference's loop holds after the same iterations.                 */
int cg_sync_vectors(cg_ctx *ctx);

#define CG_VEC_U 0
#define CG_VEC_G 1
#define CG_VEC_D 2
#define CG_VEC_Q 3
/* Copy one of the context's vectors (rows*(n+1) doubles) to host memory.     */
int cg_copy_out(cg_ctx *ctx, int which, double *host_dst, long count);
/* Device pointer of one of the context's vectors, for reading (valid until cg_destroy or
 * the next cg_solve, which may move d and g to their other buffers).         */
int cg_device_ptr(cg_ctx *ctx, int which, double **out);
/* q1 of every iteration of the last solve (host pointer, up to `count`; the first 2^24
 * iterations of a solve are recorded).                                       */
int cg_get_history(cg_ctx *ctx, double *host_dst, long count, long *written);

/* ---- measurement ----------------------------------------------------------
 * Time `reps` back-to-back launches of one hot-path kernel of the context's
 * current schedule with HIP events on the context's own stream.
 * which, for the default LEAN schedule (with CG_REDUCE_STRICT the passes also form their span sums):
 *        0 = pass 1 (reads d, g; d.q),            16 B/point
 *        1 = pass 2 (reads d, g; writes d', g'),  32 B/point
 *        2 = pass 2 with the paired u update,     48 B/point
 *        (the two kinds of pass 2 alternate, so in a solve each of them always reads the same one of the two buffer pairs
 *        and writes the other, and the two directions can differ by 10 %: called after a solve, 1 and 2 are timed in the
 *        direction that solve's next such pass would have run in; on a fresh context the launches alternate);
 * for CLASSIC / FUSED / FUSED_NOQ:
 *        0 = stencil kernel (as scheduled: with the fused d-update if any),
 *        1 = update kernel, 2 = d-update kernel (CLASSIC only).
 * Outputs the mean milliseconds per launch and the algorithmic bytes one
 * launch moves (the per-point figure of the schedule times the block's points).
 * The vectors are overwritten: cg_solve and cg_copy_out fail with CG_ERR_STATE until
 * cg_init_problem has been called again.                                     */
int cg_time_kernel(cg_ctx *ctx, int which, int reps, double *ms_per_launch, double *bytes_per_launch);

/* ---- multi-GPU: one process per GPU, RCCL over xGMI -----------------------
 * Replaces init_mpi_settings (utils.c:383-420), exchange_boundaries
 * (utils.c:221-274, rows only) and the MPI_Allreduce in dot (utils.c:380).
 * Rank r owns rows [r*G/P, (r+1)*G/P): equal blocks in rank order, P must divide G = n+1
 * (cg_create rejects anything else).  The 128-byte id comes from rank 0's
 * cg_comm_unique_id and reaches the other ranks through any host channel
 * (torch.distributed / MPI broadcast).                                       */
int cg_comm_unique_id(void *id128);
int cg_comm_init(cg_ctx *ctx, const void *id128);

/* The same exchanges carried by the caller instead of RCCL: the library stages
 * the rows / the scalar through pinned host memory and calls these functions
 * (all pointers are HOST pointers; return 0 on success).  Synchronous and slow --
 * meant for running every multi-rank code path with several processes that
 * share one GPU (tests), and as a fallback where RCCL is unavailable.
 *   exchange: send_up -> rank-1, send_down -> rank+1, and fill recv_top from
 *             rank-1's send_down, recv_bot from rank+1's send_up; `count` doubles
 *             each; a NULL pair means there is no neighbour on that side
 *             (utils.c:231-248).
 *   allgather: all[r*count .. r*count+count) = rank r's `mine[0..count)`, r = 0..nranks-1
 *             (stands in for the MPI_Allreduce of utils.c:380: the ranks' partial sums are
 *             concatenated in rank order and added in one fixed order on every rank). */
typedef int (*cg_exchange_fn)(void *user, const double *send_up, const double *send_down,
                              double *recv_top, double *recv_bot, long count);
typedef int (*cg_allgather_fn)(void *user, const double *mine, long count, double *all);
int cg_comm_init_host(cg_ctx *ctx, cg_exchange_fn exchange, cg_allgather_fn allgather, void *user);

/* Timing rehearsal on one GPU: the context keeps its place in a P-rank decomposition (its row block, its halo
 * rows, its slice of the all-gather buffers), but a ONE-rank RCCL communicator stands in for every peer -- rows
 * are sent to and received from the calling rank itself, the all-gathers cover its own slice.  The numbers such
 * a solve produces mean nothing; what it shows is the time one rank's iteration takes with the RCCL calls of the
 * real schedule in it.  (No counterpart in the reference.) */
int cg_comm_init_self(cg_ctx *ctx);

/* Transport self-test, collective over the ranks of the communicator: an all-gather of one
 * tagged double per rank, a tagged row exchange with the neighbours above and below, and one
 * row sent to the calling rank itself; every received value is compared with what the sender
 * must have sent.  Uses one of the direction buffers as scratch, so call it between
 * cg_comm_init / cg_comm_init_host and cg_init_problem.  cg_comm_init also accepts a
 * single-rank context so that one process can check the RCCL calls this way.
 * (The reference has no counterpart: a failed MPI call in utils.c:221-274 simply aborts.) */
int cg_comm_selftest(cg_ctx *ctx);

#ifdef __cplusplus
}
#endif
#endif
