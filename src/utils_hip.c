/*
 * The reference's nine utils.h entry points (utils.h:34-42), implemented on top
 * of the HIP C ABI (include/cg_hip.h).  Plain C; links against libcg_hip.so.
 *
 * Every array argument is a device pointer.  Errors follow the reference's own
 * convention for bad input -- print "[ERROR] ..." and exit(1) (cg.c:15-18,23-26)
 * -- because the signatures have no way to return a status.
 *
 * Build with -DCG_USE_MPI (and an MPI compiler wrapper) for more than one rank:
 * one rank per GPU, row blocks, halo rows staged through the host with
 * MPI_Sendrecv and the dot products completed with MPI_Allreduce, exactly the
 * collective the reference uses (utils.c:380).  This call-by-call path exists
 * for source compatibility; the fast multi-GPU path is cg_solve with
 * cg_comm_init (RCCL, device-resident scalars, hipGraph).
 */
#ifdef CG_USE_MPI
#include <mpi.h>
#endif
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "utils.h"
#include "cg_hip.h"

/* The reference passes no context to init_localg, dot or apply_stencil_serial;
 * like the reference (shared request slots in MPI_Settings) this layer is not
 * re-entrant: it remembers the settings created by init_mpi_settings. */
static MPI_Settings *g_settings = NULL;

static void die(const char *what)
{
    fprintf(stderr, "[ERROR] %s: %s\n", what, cg_last_error());
    exit(1);
}

#define CK(call, what) do { if ((call) != CG_OK) die(what); } while (0)

static int cur_device(void) { return g_settings ? g_settings->device : 0; }

static void check_stencil(const stencil_struct *s)
{
    /* The reference's kernels are written for the 3x3 array (utils.c:27-31 index it directly);
     * all nine entries are passed on as they are (the reference itself reads the five cross entries only). */
    if (s == NULL || s->size != 3 || s->extent != 1 || s->stencil == NULL) {
        fprintf(stderr, "[ERROR] only the 3x3 stencil of extent 1 is supported\n");
        exit(1);
    }
}

/* utils.c:383-420.  chunklength = row length n+1; this rank owns (n+1)/numprocs rows. */
MPI_Settings *init_mpi_settings(int numprocs, int chunklength)
{
    MPI_Settings *s = (MPI_Settings *)calloc(1, sizeof(MPI_Settings));
    if (!s) { fprintf(stderr, "[ERROR] out of host memory\n"); exit(1); }
    int rank = 0;
#ifdef CG_USE_MPI
    MPI_Comm_rank(MPI_COMM_WORLD, &rank);
    s->cartcomm = MPI_COMM_WORLD;
#else
    if (numprocs != 1) {
        fprintf(stderr, "[ERROR] this build has no MPI: numprocs must be 1 (got %d)\n", numprocs);
        exit(1);
    }
#endif
    if (numprocs < 1 || chunklength % numprocs != 0) {
        fprintf(stderr, "[ERROR] Number of rows '%d' not divisible by number of processors '%d'.\n", chunklength, numprocs);
        exit(1);
    }
    int ndev = 0;
    CK(cg_device_count(&ndev), "cg_device_count");
    if (ndev < 1) { fprintf(stderr, "[ERROR] no GPU visible; this build has no CPU path\n"); exit(1); }
    s->cartsize = numprocs;
    s->rank = rank;
    s->numprocs = numprocs;
    s->device = rank % ndev;
    s->local_cols = chunklength;
    s->local_rows = chunklength / numprocs;
    s->row0 = (long)rank * s->local_rows;
    s->toprank = rank > 0 ? rank - 1 : -1;
    s->bottomrank = rank < numprocs - 1 ? rank + 1 : -1;
    s->leftrank = s->rightrank = -1;
    s->itags = (int *)malloc((size_t)numprocs * sizeof(int));
    s->jtags = (int *)malloc((size_t)numprocs * sizeof(int));
    s->irequests = (MPI_Request *)calloc((size_t)numprocs, sizeof(MPI_Request));
    s->jrequests = (MPI_Request *)calloc((size_t)numprocs, sizeof(MPI_Request));
    for (int i = 0; i < numprocs; ++i) s->itags[i] = s->jtags[i] = i;      /* utils.c:401-404 */
    s->host_row[0] = (double *)malloc((size_t)chunklength * sizeof(double));
    s->host_row[1] = (double *)malloc((size_t)chunklength * sizeof(double));
    const char *e = getenv("CG_STRICT");
    s->strict_dot = (e && *e && *e != '0') ? 1 : 0;
    g_settings = s;
    return s;
}

/* utils.c:276-340. */
d_struct *init_locald(const int n, const int chunklength, const int myrank, MPI_Settings *s)
{
    (void)myrank;
    if (chunklength != n + 1) {
        fprintf(stderr, "[ERROR] row-block build: chunklength must be the row length n+1 = %d (got %d)\n", n + 1, chunklength);
        exit(1);
    }
    d_struct *d = (d_struct *)calloc(1, sizeof(d_struct));
    if (!d) { fprintf(stderr, "[ERROR] out of host memory\n"); exit(1); }
    const long G = s->local_cols, rows = s->local_rows;
    CK(cg_malloc(s->device, rows * G, &d->locald), "cg_malloc(locald)");
    if (s->toprank >= 0) CK(cg_malloc(s->device, G, &d->top_pad), "cg_malloc(top_pad)");         /* utils.c:294-301 */
    if (s->bottomrank >= 0) CK(cg_malloc(s->device, G, &d->bottom_pad), "cg_malloc(bottom_pad)"); /* utils.c:290-293 */
    CK(cg_init_d(s->device, n, s->row0, rows, d->locald), "cg_init_d");
    return d;
}

/* utils.c:342-350.  The vector has local_rows * chunklength elements. */
double *init_localg(const int chunklength, double *locald)
{
    const long rows = g_settings ? g_settings->local_rows : chunklength;
    double *g = NULL;
    CK(cg_malloc(cur_device(), rows * chunklength, &g), "cg_malloc(localg)");
    CK(cg_init_g(cur_device(), rows * chunklength, locald, g), "cg_init_g");
    return g;
}

/* utils.c:8-36. */
void apply_stencil_serial(int n, stencil_struct *my_stencil, double *d, double *q)
{
    check_stencil(my_stencil);
    const long G = (long)n + 1;
    CK(cg_stencil9(cur_device(), my_stencil->stencil, G, 0, G, d, NULL, NULL, q, NULL), "cg_stencil9");
}

/* utils.c:221-274, rows only.  Device rows are staged through the host. */
void exchange_boundaries(const int chunklength, d_struct *ld, const int myrank, MPI_Settings *s)
{
    (void)myrank;
#ifdef CG_USE_MPI
    const long G = chunklength, rows = s->local_rows;
    MPI_Status st;
    if (ld->bottom_pad != NULL) {          /* last row down, neighbour's first row back (utils.c:234-239) */
        CK(cg_copy_to_host(s->device, s->host_row[0], ld->locald + (rows - 1) * G, G), "halo d2h");
        MPI_Sendrecv(s->host_row[0], (int)G, MPI_DOUBLE, s->bottomrank, s->itags[s->rank],
                     s->host_row[1], (int)G, MPI_DOUBLE, s->bottomrank, s->itags[s->bottomrank], s->cartcomm, &st);
        CK(cg_copy_to_device(s->device, ld->bottom_pad, s->host_row[1], G), "halo h2d");
    }
    if (ld->top_pad != NULL) {             /* first row up, neighbour's last row back (utils.c:243-248) */
        CK(cg_copy_to_host(s->device, s->host_row[0], ld->locald, G), "halo d2h");
        MPI_Sendrecv(s->host_row[0], (int)G, MPI_DOUBLE, s->toprank, s->itags[s->rank],
                     s->host_row[1], (int)G, MPI_DOUBLE, s->toprank, s->itags[s->toprank], s->cartcomm, &st);
        CK(cg_copy_to_device(s->device, ld->top_pad, s->host_row[1], G), "halo h2d");
    }
#else
    (void)chunklength; (void)ld; (void)s;  /* one rank: no neighbours */
#endif
}

/* utils.c:38-219, rows only: exchange, then the stencil over the whole block with its pad rows. */
void apply_stencil_parallel(const int chunklength, stencil_struct *my_stencil, d_struct *ld, double *localq,
                            const int myrank, MPI_Settings *s)
{
    check_stencil(my_stencil);
    exchange_boundaries(chunklength, ld, myrank, s);
    CK(cg_stencil9(s->device, my_stencil->stencil, chunklength, s->row0, s->local_rows, ld->locald, ld->top_pad, ld->bottom_pad, localq, NULL),
       "cg_stencil");
}

/* utils.c:368-381.  rows*cols is all that matters, as in the reference. */
void dot(const int rows, const int cols, double *localv, double *localw, MPI_Comm comm, double *result)
{
    (void)comm;
    MPI_Settings *s = g_settings;
    const long rr = (s && s->numprocs > 1) ? s->local_rows : rows;
    const long count = rr * (long)cols;
    const int strict = s ? s->strict_dot : 0;
    double local = 0.0;
#ifdef CG_USE_MPI
    if (s && s->numprocs > 1) {
        if (strict) {
            /* one accumulator handed from rank to rank: the np=1 summation order */
            double start = 0.0;
            MPI_Status st;
            if (s->rank > 0) MPI_Recv(&start, 1, MPI_DOUBLE, s->rank - 1, 77, s->cartcomm, &st);
            CK(cg_dot(s->device, count, localv, localw, 1, start, &local), "cg_dot");
            if (s->rank < s->numprocs - 1) MPI_Send(&local, 1, MPI_DOUBLE, s->rank + 1, 77, s->cartcomm);
            MPI_Bcast(&local, 1, MPI_DOUBLE, s->numprocs - 1, s->cartcomm);
            *result = local;
        } else {
            CK(cg_dot(s->device, count, localv, localw, 0, 0.0, &local), "cg_dot");
            MPI_Allreduce(&local, result, 1, MPI_DOUBLE, MPI_SUM, s->cartcomm);   /* utils.c:380 */
        }
        return;
    }
#endif
    CK(cg_dot(cur_device(), count, localv, localw, strict, 0.0, &local), "cg_dot");
    *result = local;
}

/* utils.c:352-366: debug dump of a local array, same text format; the block
 * coordinates are (rank, 0) in a row-block decomposition. */
void print_local2dmesh(const int rows, const int cols, double *mesh, const int myrank, MPI_Comm cartcomm)
{
    (void)cartcomm;
    if (mesh == NULL) { for (int i = 0; i < rows; ++i) putchar('\n'); return; }
    double *h = (double *)malloc((size_t)rows * cols * sizeof(double));
    if (!h) { fprintf(stderr, "[ERROR] out of host memory\n"); exit(1); }
    CK(cg_copy_to_host(cur_device(), h, mesh, (long)rows * cols), "cg_copy_to_host");
    for (int i = 0; i < rows; ++i) {
        for (int j = 0; j < cols; ++j) printf("([%d (%d,%d)] %lf) ", myrank, myrank, 0, h[(long)i * cols + j]);
        putchar('\n');
    }
    free(h);
}

/* utils.c:422-435: frees the members, not the containers. */
void free_struct_elems(stencil_struct *stencil, d_struct *ld, MPI_Settings *s)
{
    free(stencil->stencil);
    const int dev = s ? s->device : 0;
    if (ld->top_pad) cg_free(dev, ld->top_pad);
    if (ld->locald) cg_free(dev, ld->locald);
    if (ld->bottom_pad) cg_free(dev, ld->bottom_pad);
    free(s->itags); free(s->jtags); free(s->irequests); free(s->jrequests);
    free(s->host_row[0]); free(s->host_row[1]);
    if (g_settings == s) g_settings = NULL;
}
