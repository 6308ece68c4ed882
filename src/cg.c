/*
 * C driver of the MI355X conjugate-gradient solver: the reference's command line
 * and output (cg.c:8-153), with the numerical work done on the GPU.
 *
 *   cg <n> [iters] [rtol]
 *
 *   n      grid has (n+1) x (n+1) points, as in the reference (cg.c:20)
 *   iters  iteration count, default MAX_ITERS = 200 (utils.h:1, a compile-time
 *          constant in the reference)
 *   rtol   optional stop rule sqrt(q1/q_init) <= rtol (the reference has none)
 *
 * Environment:
 *   CG_PATH=solve   (default) whole loop on the device: cg_solve, device-resident
 *                   scalars, fused kernels, hipGraph
 *   CG_PATH=calls   the reference's own loop structure (cg.c:53-130), one library
 *                   call per step through the utils.h entry points
 *   CG_LAYOUT=2d    with CG_PATH=calls on P = s*s ranks: the reference's own decomposition into s x s chunks
 *                   (row and column halos, corner points) instead of row blocks; with CG_STRICT=1 the output is
 *                   `mpirun -np P ./cg n` of the reference bit for bit
 *   CG_STRICT=1     dot products in the reference's exact summation order
 *                   (bit-identical results, about four fifths of the fast rate on one GPU)
 *   CG_STRICT_NP=P  with CG_STRICT=1 on one GPU: the summation layout of the reference run
 *                   on P = 4, 16 or 64 ranks, i.e. the bits of `mpirun -np P ./cg n`
 *   CG_SCHEDULE=classic|fused|fused_noq|lean   kernel schedule of the solve path
 *                   (default lean; fused_noq when HBM is too short for lean's second gradient buffer)
 *   CG_STENCIL="up,left,centre,right,down"   other values for the five entries of cg.c:33's
 *                   coefficient array that the reference's kernels read (default -1,-1,4,-1,-1);
 *                   nine values set the whole 3x3 array row-wise, corners included (the reference
 *                   declares those entries but never reads them)
 *   CG_DUMP=prefix  write the final u, g, d as raw doubles to <prefix>.{u,g,d}.f64: the whole grid in
 *                   row-major order on any number of ranks (the ranks append their row blocks in turn).
 *                   The reference's counterpart is the array dump whose call is commented out (cg.c:135).
 *   CG_VERIFY=1     one more line "[VERIFY] sum_u = ... max_err = ...": the sum of u added one by one in
 *                   index order (handed from rank to rank) and max |u - x(1-x)y(1-y)|, the distance to the
 *                   closed-form solution of the problem; both formed on the GPUs
 *   CG_TRANSPORT=host   several ranks, solve path: halo rows and partial sums staged through the host and
 *                   carried by MPI instead of RCCL (ranks that share one GPU, or no RCCL)
 *
 * Output on rank 0: the reference's two lines (cg.c:139-140) followed by one
 * machine-readable line with full-precision q1, rates and achieved bandwidth.
 *
 * More than one rank: build with -DCG_USE_MPI and start one rank per GPU with
 * mpirun; rank r owns rows [r*(n+1)/P, (r+1)*(n+1)/P).  Any P that divides n+1
 * is accepted (the reference needs a perfect square, cg.c:15).
 */
#ifdef CG_USE_MPI
#include <mpi.h>
#endif
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "utils.h"
#include "cg_hip.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static void die(const char *what)
{
    fprintf(stderr, "[ERROR] %s: %s\n", what, cg_last_error());
    exit(1);
}
#define CK(call, what) do { if ((call) != CG_OK) die(what); } while (0)

static void barrier(void)
{
#ifdef CG_USE_MPI
    MPI_Barrier(MPI_COMM_WORLD);
#endif
}

#ifdef CG_USE_MPI
/* Host-staged transport of the solve path (cg_comm_init_host): the two exchanges carried by MPI. */
typedef struct { int rank, numprocs; } host_transport;

static int mpi_exchange(void *user, const double *send_up, const double *send_down, double *recv_top,
                        double *recv_bot, long count)
{
    const host_transport *t = (const host_transport *)user;
    MPI_Status st;
    if (send_down != NULL)                 /* last row down, the lower neighbour's first row back (utils.c:234-239) */
        MPI_Sendrecv(send_down, (int)count, MPI_DOUBLE, t->rank + 1, 11, recv_bot, (int)count, MPI_DOUBLE,
                     t->rank + 1, 12, MPI_COMM_WORLD, &st);
    if (send_up != NULL)                   /* first row up, the upper neighbour's last row back (utils.c:243-248) */
        MPI_Sendrecv(send_up, (int)count, MPI_DOUBLE, t->rank - 1, 12, recv_top, (int)count, MPI_DOUBLE,
                     t->rank - 1, 11, MPI_COMM_WORLD, &st);
    return 0;
}

static int mpi_allgather(void *user, const double *mine, long count, double *all)
{
    (void)user;
    /* `mine` is this rank's slice of `all`: gather in place */
    (void)mine;
    return MPI_Allgather(MPI_IN_PLACE, (int)count, MPI_DOUBLE, all, (int)count, MPI_DOUBLE, MPI_COMM_WORLD) == MPI_SUCCESS ? 0 : 1;
}
#endif

/* CG_VERIFY and CG_DUMP on the final vectors (device pointers of this rank's row block). */
static void inspect(int device, int rank, int numprocs, int n, long row0, long rows, const double *u,
                    const double *g, const double *d)
{
    const long G = (long)n + 1, count = rows * G;
    const char *ver = getenv("CG_VERIFY");
    if (ver && *ver && *ver != '0') {
        double start = 0.0, sum_u = 0.0, max_err = 0.0;
#ifdef CG_USE_MPI
        MPI_Status st;
        if (rank > 0) MPI_Recv(&start, 1, MPI_DOUBLE, rank - 1, 78, MPI_COMM_WORLD, &st);
#endif
        CK(cg_verify_u(device, n, row0, rows, u, start, &sum_u, &max_err), "cg_verify_u");
#ifdef CG_USE_MPI
        if (rank < numprocs - 1) MPI_Send(&sum_u, 1, MPI_DOUBLE, rank + 1, 78, MPI_COMM_WORLD);
        MPI_Bcast(&sum_u, 1, MPI_DOUBLE, numprocs - 1, MPI_COMM_WORLD);
        double all_max = max_err;
        MPI_Allreduce(&max_err, &all_max, 1, MPI_DOUBLE, MPI_MAX, MPI_COMM_WORLD);
        max_err = all_max;
#endif
        if (rank == 0) printf("[VERIFY] sum_u = %.17e (%a) max_err = %.6e\n", sum_u, sum_u, max_err);
    }
    const char *dump = getenv("CG_DUMP");
    if (dump && *dump) {
        const double *vec[3] = {u, g, d};
        const char *ext[3] = {"u", "g", "d"};
        double *h = (double *)malloc((size_t)count * sizeof(double));
        if (!h) { fprintf(stderr, "[ERROR] out of host memory\n"); exit(1); }
        for (int v = 0; v < 3; ++v) {
            if (vec[v] == NULL) continue;
            char fn[4096];
            snprintf(fn, sizeof fn, "%s.%s.f64", dump, ext[v]);
            for (int r = 0; r < numprocs; ++r) {       /* row blocks in rank order = the grid in row-major order */
                if (r == rank) {
                    CK(cg_copy_to_host(device, h, vec[v], count), "cg_copy_to_host");
                    FILE *f = fopen(fn, r == 0 ? "wb" : "ab");
                    if (f) { fwrite(h, sizeof(double), (size_t)count, f); fclose(f); }
                }
                barrier();
            }
        }
        free(h);
    }
}

/* Every way out of main passes here: an MPI process that leaves without MPI_Finalize takes the job down noisily. */
static int quit(int code)
{
#ifdef CG_USE_MPI
    MPI_Finalize();
#endif
    return code;
}

int main(int argc, char **argv)
{
    int rank = 0, numprocs = 1;
#ifdef CG_USE_MPI
    MPI_Init(&argc, &argv);
    MPI_Comm_size(MPI_COMM_WORLD, &numprocs);
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
#endif
    if (argc < 2) {
        if (rank == 0) fprintf(stderr, "usage: %s <n> [iters] [rtol]\n", argv[0]);
        return quit(1);
    }
    const int n = atoi(argv[1]);
    const long iters = argc > 2 ? atol(argv[2]) : MAX_ITERS;
    const double rtol = argc > 3 ? atof(argv[3]) : 0.0;
    if (n < 2 || iters < 0 || rtol < 0) {
        if (rank == 0) printf("[ERROR] need n >= 2, iters >= 0, rtol >= 0.\n");
        return quit(1);
    }
    const long G = (long)n + 1;
    const char *path = getenv("CG_PATH");
    const char *strict_e = getenv("CG_STRICT");
    const int strict = strict_e && *strict_e && *strict_e != '0';
    const int use_calls = path && strcmp(path, "calls") == 0;
    const char *lay = getenv("CG_LAYOUT");
    const int two_d = use_calls && lay && strcmp(lay, "2d") == 0 && numprocs > 1;   /* the reference's own decomposition */
    if (!two_d && G % numprocs != 0) {                         /* cg.c:23-26, for row blocks */
        if (rank == 0) printf("[ERROR] Number of points '%d+1' not divisible by number of processors '%d'.\n", n, numprocs);
        return quit(1);
    }
    const long rows = G / numprocs, row0 = rank * rows, count = rows * G;   /* (row blocks) */

    int my_stencil[9] = {0, -1, 0, -1, 4, -1, 0, -1, 0};              /* cg.c:33 */
    const char *ste = getenv("CG_STENCIL");
    if (ste && *ste) {
        int v[9];
        const int got = sscanf(ste, "%d,%d,%d,%d,%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4], &v[5], &v[6], &v[7], &v[8]);
        if (got == 9) {
            memcpy(my_stencil, v, sizeof my_stencil);
        } else if (got == 5) {
            my_stencil[1] = v[0]; my_stencil[3] = v[1]; my_stencil[4] = v[2]; my_stencil[5] = v[3]; my_stencil[7] = v[4];
        } else {
            if (rank == 0) printf("[ERROR] CG_STENCIL needs five integers (up,left,centre,right,down) or all nine.\n");
            return quit(1);
        }
    }

    double q1 = 0.0, q_init = 0.0, runtime = 0.0;
    long done_iters = 0;
    int bytes_per_point = 88;
    int device = 0;

    if (use_calls) {
        /* ---- the reference's loop, one call per step (cg.c:30-130) ---- */
        bytes_per_point = 112;                                         /* six separate passes, as in the reference */
        int chunk = (int)G;                                            /* row blocks: chunklength is the row length */
        if (two_d) {
            const int side = (int)(sqrt((double)numprocs) + 0.5);
            if (side * side != numprocs) { if (rank == 0) printf("[ERROR] Number of processors must be a perfect square.\n"); return quit(1); }   /* cg.c:15-18 */
            /* cg.c:21-26 tests (n+1)^2 % numprocs, which for a square rank count is this */
            if (G % side != 0) { if (rank == 0) printf("[ERROR] Number of points '%d+1' not divisible by the %d x %d processor grid.\n", n, side, side); return quit(1); }
            chunk = (int)(G / side);                                   /* cg.c:21-22 */
        }
        MPI_Settings *settings = init_mpi_settings(numprocs, chunk);
        device = settings->device;
        settings->strict_dot = strict;
        stencil_struct *stencil = (stencil_struct *)malloc(sizeof(stencil_struct));
        stencil->size = 3;
        stencil->extent = stencil->size / 2;
        stencil->stencil = (int *)malloc(sizeof(my_stencil));
        memcpy(stencil->stencil, my_stencil, sizeof(my_stencil));

        d_struct *ld = init_locald(n, chunk, rank, settings);          /* cg.c:41 */
        const long count_l = settings->local_rows * settings->local_cols;   /* (the decomposition is settled by now) */
        double *locald = ld->locald;
        double *localg = init_localg(chunk, locald);                   /* cg.c:43 */
        double *localu = NULL, *localq = NULL;
        CK(cg_malloc(device, count_l, &localu), "cg_malloc(u)");       /* cg.c:44 */
        CK(cg_malloc(device, count_l, &localq), "cg_malloc(q)");       /* cg.c:45 */

        double q0, tau, beta;
        barrier();                                                     /* cg.c:50 */
        const double t0 = now_s();
        dot(chunk, chunk, localg, localg, settings->cartcomm, &q0);    /* cg.c:55 */
        q_init = q1 = q0;
        const double stop = rtol * rtol * q0;
        for (long it = 0; it < iters; ++it) {
            if (numprocs == 1) apply_stencil_serial(n, stencil, locald, localq);                       /* cg.c:58 */
            else apply_stencil_parallel(chunk, stencil, ld, localq, rank, settings);                   /* cg.c:97 */
            dot(chunk, chunk, locald, localq, settings->cartcomm, &tau);                               /* cg.c:61 */
            tau = q0 / tau;                                                                            /* cg.c:62 */
            CK(cg_axpy(device, count_l, tau, locald, localu), "cg_axpy(u)");                           /* cg.c:63-69 */
            CK(cg_axpy(device, count_l, tau, localq, localg), "cg_axpy(g)");                           /* cg.c:70-76 */
            dot(chunk, chunk, localg, localg, settings->cartcomm, &q1);                                /* cg.c:79 */
            beta = q1 / q0;                                                                            /* cg.c:80 */
            CK(cg_xpay(device, count_l, beta, locald, localg), "cg_xpay(d)");                          /* cg.c:81-87 */
            q0 = q1;                                                                                   /* cg.c:89 */
            done_iters = it + 1;
            if (rtol > 0 && q1 <= stop) break;
        }
        barrier();                                                     /* cg.c:131 */
        runtime = now_s() - t0;
        if (!two_d) inspect(device, rank, numprocs, n, row0, rows, localu, localg, locald);
        else if (rank == 0 && (getenv("CG_VERIFY") || getenv("CG_DUMP")))
            printf("[INFO] CG_VERIFY / CG_DUMP need whole rows on every rank: not available with CG_LAYOUT=2d\n");
        cg_free(device, localu); cg_free(device, localg); cg_free(device, localq);
        free_struct_elems(stencil, ld, settings);
        free(ld); free(stencil); free(settings);
    } else {
        /* ---- whole loop on the device ---- */
        int ndev = 0;
        CK(cg_device_count(&ndev), "cg_device_count");
        if (ndev < 1) { if (rank == 0) printf("[ERROR] no GPU visible; this build has no CPU path.\n"); return quit(1); }
        device = rank % ndev;
        cg_ctx *ctx = NULL;
        CK(cg_create(&ctx, device, n, row0, rows, rank, numprocs), "cg_create");
        const char *se = getenv("CG_SCHEDULE");
        long sched = -1;                                               /* -1: the library picks */
        if (se && strcmp(se, "classic") == 0) sched = CG_SCHED_CLASSIC;
        else if (se && strcmp(se, "fused") == 0) sched = CG_SCHED_FUSED;
        else if (se && strcmp(se, "fused_noq") == 0) sched = CG_SCHED_FUSED_NOQ;
        else if (se && strcmp(se, "lean") == 0) sched = CG_SCHED_LEAN;
        CK(cg_set_option(ctx, CG_OPT_REDUCE, strict ? CG_REDUCE_STRICT : CG_REDUCE_FAST), "cg_set_option(reduce)");
        CK(cg_set_option(ctx, CG_OPT_SCHEDULE, sched), "cg_set_option(schedule)");
        CK(cg_set_stencil(ctx, my_stencil), "cg_set_stencil");
        const char *snp = getenv("CG_STRICT_NP");
        if (strict && snp && *snp) {
            const int p = atoi(snp), side = (int)(sqrt((double)p) + 0.5);
            if (side * side != p) { if (rank == 0) printf("[ERROR] Number of processors must be a perfect square.\n"); return quit(1); }   /* cg.c:15-18 */
            CK(cg_set_option(ctx, CG_OPT_STRICT_BLOCKS, side), "cg_set_option(strict blocks)");
        }
#ifdef CG_USE_MPI
        host_transport ht = {rank, numprocs};
        if (numprocs > 1) {
            const char *tr = getenv("CG_TRANSPORT");
            if (tr && strcmp(tr, "host") == 0) {
                CK(cg_comm_init_host(ctx, mpi_exchange, mpi_allgather, &ht), "cg_comm_init_host");
            } else {
                char id[128];
                if (rank == 0) CK(cg_comm_unique_id(id), "cg_comm_unique_id");
                MPI_Bcast(id, 128, MPI_BYTE, 0, MPI_COMM_WORLD);
                CK(cg_comm_init(ctx, id), "cg_comm_init");
            }
        }
#endif
        CK(cg_init_problem(ctx), "cg_init_problem");
        cg_result res;
        barrier();
        const double t0 = now_s();
        CK(cg_solve(ctx, iters, rtol, &res), "cg_solve");
        CK(cg_sync_vectors(ctx), "cg_sync_vectors");                   /* the reference's timed region ends with u, g, d final */
        barrier();
        runtime = now_s() - t0;
        q1 = res.q1; q_init = res.q_init; done_iters = res.iters;
        CK(cg_get_option(ctx, CG_OPT_SCHEDULE, &sched), "cg_get_option(schedule)");   /* the one that ran */
        bytes_per_point = sched == CG_SCHED_CLASSIC ? 88 : sched == CG_SCHED_FUSED ? 80 : sched == CG_SCHED_FUSED_NOQ ? 64 : 56;
        double *u_dev = NULL, *g_dev = NULL, *d_dev = NULL;
        CK(cg_device_ptr(ctx, CG_VEC_U, &u_dev), "cg_device_ptr(u)");
        CK(cg_device_ptr(ctx, CG_VEC_G, &g_dev), "cg_device_ptr(g)");
        CK(cg_device_ptr(ctx, CG_VEC_D, &d_dev), "cg_device_ptr(d)");
        inspect(device, rank, numprocs, n, row0, rows, u_dev, g_dev, d_dev);
        cg_destroy(ctx);
    }

    double max_runtime = runtime;
#ifdef CG_USE_MPI
    MPI_Reduce(&runtime, &max_runtime, 1, MPI_DOUBLE, MPI_MAX, 0, MPI_COMM_WORLD);    /* cg.c:134 */
#endif
    if (rank == 0) {
        printf("[INFO] norm_g^2 = %.16lf\n", q1);                                       /* cg.c:139 */
        printf("%.16lf\n", max_runtime);                                                /* cg.c:140 */
        const double its = max_runtime > 0 ? (double)done_iters / max_runtime : 0.0;
        printf("[CG] q1 = %.17e (%a) q_init = %.17e iters=%ld np=%d path=%s order=%s it/s=%.3f hbm_GBs_per_gpu=%.1f\n",
               q1, q1, q_init, done_iters, numprocs, use_calls ? "calls" : "solve", strict ? "strict" : "fast", its,
               (double)bytes_per_point * (double)count * its / 1e9);
    }
#ifdef CG_USE_MPI
    MPI_Finalize();
#endif
    return 0;
}
