/*
 * The reference's nine utils.h entry points (utils.h:34-42), implemented on top
 * of the HIP C ABI (include/cg_hip.h).  Plain C; links against libcg_hip.so.
 *
 * Every array argument is a device pointer.  Errors follow the reference's own
 * convention for bad input -- print "[ERROR] ..." and exit(1) (cg.c:15-18,23-26)
 * -- because the signatures have no way to return a status.
 *
 * Build with -DCG_USE_MPI (and an MPI compiler wrapper) for more than one rank:
 * one rank per GPU, halo rows (and columns) staged through the host with
 * MPI_Sendrecv and the dot products completed with MPI_Allreduce, exactly the
 * collective the reference uses (utils.c:380).  Two layouts:
 *   row blocks (default)   chunklength = n+1, rank r owns (n+1)/P whole rows;
 *   CG_LAYOUT=2d           the reference's own decomposition: P a perfect square,
 *                          chunklength = (n+1)/sqrt(P), rank r = carti*sqrt(P)+cartj owns
 *                          the chunk at (carti, cartj) (utils.c:408-417), with row AND
 *                          column pads, the strided column gathered on the device
 *                          (utils.c:392) and the corner points done with two pads each
 *                          (utils.c:156-218).  With CG_STRICT=1 every rank sums its chunk
 *                          in the reference's order and MPI_Allreduce combines the sums:
 *                          the bits of `mpirun -np P ./cg n` of the reference itself.
 * This call-by-call path exists for source compatibility; the fast multi-GPU
 * path is cg_solve with cg_comm_init (RCCL, device-resident scalars, hipGraph).
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

/* The rank's place in the decomposition: row blocks (chunklength = the row length n+1, this rank owns
 * (n+1)/numprocs rows), or the reference's side x side chunks (utils.c:408-417, chunklength = (n+1)/side). */
static void set_layout(MPI_Settings *s, int two_d, int chunklength)
{
    const int rank = s->rank, numprocs = s->numprocs;
    s->layout2d = two_d;
    if (two_d) {
        /* utils.c:408-417: a side x side non-periodic cartesian grid, rank = carti*side + cartj */
        int side = 1;
        while ((side + 1) * (side + 1) <= numprocs) ++side;
        if (side * side != numprocs) {
            fprintf(stderr, "[ERROR] Number of processors must be a perfect square.\n");          /* cg.c:15-18 */
            exit(1);
        }
        const int carti = rank / side, cartj = rank % side;
        s->cartsize = side;
        s->local_rows = s->local_cols = chunklength;
        s->row0 = (long)carti * chunklength;
        s->col0 = (long)cartj * chunklength;
        s->toprank = carti > 0 ? rank - side : -1;
        s->bottomrank = carti < side - 1 ? rank + side : -1;
        s->leftrank = cartj > 0 ? rank - 1 : -1;
        s->rightrank = cartj < side - 1 ? rank + 1 : -1;
        CK(cg_malloc(s->device, chunklength, &s->dev_col[0]), "cg_malloc(column staging)");
        CK(cg_malloc(s->device, chunklength, &s->dev_col[1]), "cg_malloc(column staging)");
    } else {
        if (numprocs < 1 || chunklength % numprocs != 0) {
            fprintf(stderr, "[ERROR] Number of rows '%d' not divisible by number of processors '%d'.\n", chunklength, numprocs);
            exit(1);
        }
        s->cartsize = numprocs;
        s->local_cols = chunklength;
        s->local_rows = chunklength / numprocs;
        s->row0 = (long)rank * s->local_rows;
        s->col0 = 0;
        s->toprank = rank > 0 ? rank - 1 : -1;
        s->bottomrank = rank < numprocs - 1 ? rank + 1 : -1;
        s->leftrank = s->rightrank = -1;
    }
}

/* utils.c:383-420. */
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
    int ndev = 0;
    CK(cg_device_count(&ndev), "cg_device_count");
    if (ndev < 1) { fprintf(stderr, "[ERROR] no GPU visible; this build has no CPU path\n"); exit(1); }
    s->rank = rank;
    s->numprocs = numprocs;
    s->device = rank % ndev;
    /* Which decomposition `chunklength` belongs to: CG_LAYOUT says it (2d / rows); unset, a single rank has only one,
     * and several ranks are decided in init_locald, the first call that also knows n: the reference's own main passes
     * (n+1)/sqrt(P) (cg.c:22), the row-block driver passes n+1. */
    const char *lay = getenv("CG_LAYOUT");
    if (numprocs == 1) set_layout(s, 0, chunklength);
    else if (lay && strcmp(lay, "2d") == 0) set_layout(s, 1, chunklength);
    else if (lay && *lay) set_layout(s, 0, chunklength);
    else s->layout2d = -1;
    s->itags = (int *)malloc((size_t)numprocs * sizeof(int));
    s->jtags = (int *)malloc((size_t)numprocs * sizeof(int));
    s->irequests = (MPI_Request *)calloc((size_t)numprocs, sizeof(MPI_Request));
    s->jrequests = (MPI_Request *)calloc((size_t)numprocs, sizeof(MPI_Request));
    for (int i = 0; i < numprocs; ++i) s->itags[i] = s->jtags[i] = i;      /* utils.c:401-404 */
    s->host_row[0] = (double *)malloc(((size_t)chunklength + 2) * sizeof(double));     /* a row, and a corner value at each end */
    s->host_row[1] = (double *)malloc(((size_t)chunklength + 2) * sizeof(double));
    const char *e = getenv("CG_STRICT");
    s->strict_dot = (e && *e && *e != '0') ? 1 : 0;
    g_settings = s;
    return s;
}

/* utils.c:276-340. */
d_struct *init_locald(const int n, const int chunklength, const int myrank, MPI_Settings *s)
{
    (void)myrank;
    if (s->layout2d < 0) {                       /* undecided so far: n tells which decomposition chunklength belongs to */
        int side = 1;
        while ((side + 1) * (side + 1) <= s->numprocs) ++side;
        const int square = side * side == s->numprocs;
        if (chunklength != n + 1 && square && (long)chunklength * side == (long)n + 1) set_layout(s, 1, chunklength);
        else set_layout(s, 0, chunklength);
    }
    if (!s->layout2d && chunklength != n + 1) {
        fprintf(stderr, "[ERROR] row-block build: chunklength must be the row length n+1 = %d (got %d); "
                        "CG_LAYOUT=2d selects the reference's sqrt(P) x sqrt(P) chunks\n", n + 1, chunklength);
        exit(1);
    }
    if (s->layout2d && (long)chunklength * s->cartsize != (long)n + 1) {
        fprintf(stderr, "[ERROR] Number of points '%d+1' not divisible by the %d x %d processor grid.\n", n, s->cartsize, s->cartsize);
        exit(1);
    }
    d_struct *d = (d_struct *)calloc(1, sizeof(d_struct));
    if (!d) { fprintf(stderr, "[ERROR] out of host memory\n"); exit(1); }
    const long cols = s->local_cols, rows = s->local_rows;
    CK(cg_malloc(s->device, rows * cols, &d->locald), "cg_malloc(locald)");
    if (s->toprank >= 0) CK(cg_malloc(s->device, cols, &d->top_pad), "cg_malloc(top_pad)");         /* utils.c:294-301 */
    if (s->bottomrank >= 0) CK(cg_malloc(s->device, cols, &d->bottom_pad), "cg_malloc(bottom_pad)"); /* utils.c:290-293 */
    if (s->leftrank >= 0) CK(cg_malloc(s->device, rows, &d->left_pad), "cg_malloc(left_pad)");       /* utils.c:310-316 */
    if (s->rightrank >= 0) CK(cg_malloc(s->device, rows, &d->right_pad), "cg_malloc(right_pad)");    /* utils.c:306-309 */
    /* the four values diagonally across the chunk's corners (nine-point coefficients; zero until the first exchange) */
    if (s->layout2d) CK(cg_malloc(s->device, 4, &d->corner_pad), "cg_malloc(corner_pad)");
    if (s->layout2d) CK(cg_init_d_block(s->device, n, s->row0, s->col0, rows, cols, d->locald), "cg_init_d_block");
    else CK(cg_init_d(s->device, n, s->row0, rows, d->locald), "cg_init_d");
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

/* utils.c:221-274.  Device rows (and, in the two-dimensional layout, columns) are staged through the host. */
#ifdef CG_USE_MPI
static void swap_with(MPI_Settings *s, int peer, const double *dev_send, double *dev_recv, long count)
{
    MPI_Status st;
    CK(cg_copy_to_host(s->device, s->host_row[0], dev_send, count), "halo d2h");
    MPI_Sendrecv(s->host_row[0], (int)count, MPI_DOUBLE, peer, s->itags[s->rank],
                 s->host_row[1], (int)count, MPI_DOUBLE, peer, s->itags[peer], s->cartcomm, &st);
    CK(cg_copy_to_device(s->device, dev_recv, s->host_row[1], count), "halo h2d");
}
#endif

#ifdef CG_USE_MPI
/* Two-dimensional layout, rows with the corner values riding along: the row going up or down is sent with the first
 * (last) element of this rank's left and right pads at its ends -- the points that lie diagonally across the
 * receiver's corners -- and received as [corner, row, corner].  The columns must have been exchanged before.
 * first = 1: the chunk's first row goes to `peer` above, its reply fills top_pad and corners 0, 1; first = 0: the
 * last row goes down, the reply fills bottom_pad and corners 2, 3. */
static void swap_row_with_corners(MPI_Settings *s, d_struct *ld, int peer, int first)
{
    const long cols = s->local_cols, rows = s->local_rows;
    const long r = first ? 0 : rows - 1;
    MPI_Status st;
    double *out = s->host_row[0], *in = s->host_row[1];
    out[0] = out[cols + 1] = 0.0;
    if (ld->left_pad) CK(cg_copy_to_host(s->device, &out[0], ld->left_pad + r, 1), "corner d2h");
    CK(cg_copy_to_host(s->device, out + 1, ld->locald + r * cols, cols), "halo d2h");
    if (ld->right_pad) CK(cg_copy_to_host(s->device, &out[cols + 1], ld->right_pad + r, 1), "corner d2h");
    MPI_Sendrecv(out, (int)cols + 2, MPI_DOUBLE, peer, s->itags[s->rank], in, (int)cols + 2, MPI_DOUBLE, peer, s->itags[peer],
                 s->cartcomm, &st);
    CK(cg_copy_to_device(s->device, first ? ld->top_pad : ld->bottom_pad, in + 1, cols), "halo h2d");
    const double pair[2] = {in[0], in[cols + 1]};
    CK(cg_copy_to_device(s->device, ld->corner_pad + (first ? 0 : 2), pair, 2), "corner h2d");
}
#endif

void exchange_boundaries(const int chunklength, d_struct *ld, const int myrank, MPI_Settings *s)
{
    (void)myrank; (void)chunklength;
#ifdef CG_USE_MPI
    const long cols = s->local_cols, rows = s->local_rows;
    if (s->layout2d) {
        /* columns first (utils.c:255-272), then the rows with the corner values at their ends: what the
         * reference's corner_pad (utils.h:15) would hold if its nine-point case had been finished */
        if (ld->right_pad != NULL) {
            CK(cg_pack_column(s->device, rows, cols, ld->locald, cols - 1, s->dev_col[0]), "cg_pack_column");
            swap_with(s, s->rightrank, s->dev_col[0], ld->right_pad, rows);
        }
        if (ld->left_pad != NULL) {
            CK(cg_pack_column(s->device, rows, cols, ld->locald, 0, s->dev_col[1]), "cg_pack_column");
            swap_with(s, s->leftrank, s->dev_col[1], ld->left_pad, rows);
        }
        if (ld->bottom_pad != NULL) swap_row_with_corners(s, ld, s->bottomrank, 0);
        if (ld->top_pad != NULL) swap_row_with_corners(s, ld, s->toprank, 1);
        return;
    }
    /* last row down, the lower neighbour's first row back (utils.c:234-239); first row up (utils.c:243-248) */
    if (ld->bottom_pad != NULL) swap_with(s, s->bottomrank, ld->locald + (rows - 1) * cols, ld->bottom_pad, cols);
    if (ld->top_pad != NULL) swap_with(s, s->toprank, ld->locald, ld->top_pad, cols);
    /* last column to the right, first column to the left (utils.c:255-272): the strided column is gathered on
     * the device (the reference's MPI_Type_vector, utils.c:392) and received contiguously, as there */
    if (ld->right_pad != NULL) {
        CK(cg_pack_column(s->device, rows, cols, ld->locald, cols - 1, s->dev_col[0]), "cg_pack_column");
        swap_with(s, s->rightrank, s->dev_col[0], ld->right_pad, rows);
    }
    if (ld->left_pad != NULL) {
        CK(cg_pack_column(s->device, rows, cols, ld->locald, 0, s->dev_col[1]), "cg_pack_column");
        swap_with(s, s->leftrank, s->dev_col[1], ld->left_pad, rows);
    }
#else
    (void)ld; (void)s;                     /* one rank: no neighbours */
#endif
}

/* utils.c:38-219: exchange, then the stencil over the whole chunk with its pads. */
void apply_stencil_parallel(const int chunklength, stencil_struct *my_stencil, d_struct *ld, double *localq,
                            const int myrank, MPI_Settings *s)
{
    check_stencil(my_stencil);
    exchange_boundaries(chunklength, ld, myrank, s);
    if (s->layout2d)
        CK(cg_stencil_block9(s->device, my_stencil->stencil, s->local_rows, s->local_cols, ld->locald, ld->top_pad, ld->bottom_pad,
                             ld->left_pad, ld->right_pad, ld->corner_pad, localq), "cg_stencil_block9");
    else
        CK(cg_stencil9(s->device, my_stencil->stencil, chunklength, s->row0, s->local_rows, ld->locald, ld->top_pad, ld->bottom_pad,
                       localq, NULL), "cg_stencil");
}

/* utils.c:368-381.  rows*cols is all that matters, as in the reference. */
void dot(const int rows, const int cols, double *localv, double *localw, MPI_Comm comm, double *result)
{
    (void)comm;
    MPI_Settings *s = g_settings;
    const long rr = (s && s->numprocs > 1) ? s->local_rows : rows;
    const long count = rr * (long)((s && s->layout2d) ? s->local_cols : cols);
    const int strict = s ? s->strict_dot : 0;
    double local = 0.0;
#ifdef CG_USE_MPI
    if (s && s->numprocs > 1 && s->layout2d) {
        /* the reference's own summation: every rank its chunk (in its exact order when strict), MPI_Allreduce (utils.c:380) */
        CK(cg_dot(s->device, count, localv, localw, strict, 0.0, &local), "cg_dot");
        MPI_Allreduce(&local, result, 1, MPI_DOUBLE, MPI_SUM, s->cartcomm);
        return;
    }
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
    if (ld->left_pad) cg_free(dev, ld->left_pad);
    if (ld->top_pad) cg_free(dev, ld->top_pad);
    if (ld->locald) cg_free(dev, ld->locald);
    if (ld->bottom_pad) cg_free(dev, ld->bottom_pad);
    if (ld->right_pad) cg_free(dev, ld->right_pad);
    if (ld->corner_pad) cg_free(dev, ld->corner_pad);
    if (s->dev_col[0]) cg_free(dev, s->dev_col[0]);
    if (s->dev_col[1]) cg_free(dev, s->dev_col[1]);
    free(s->itags); free(s->jtags); free(s->irequests); free(s->jrequests);
    free(s->host_row[0]); free(s->host_row[1]);
    if (g_settings == s) g_settings = NULL;
}
