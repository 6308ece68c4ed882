/*
 * Drop-in replacement for the reference's utils.h (omarmalik710/conjugate-gradient,
 * utils.h:1-42): the same two macros, three structs and nine prototypes, so the
 * call sites in the reference's cg.c (cg.c:30,41,43,55,58,61,79,97,146) compile
 * unchanged against it.  What changes is where the data lives:
 *
 *   every `double *` handed to or returned by these functions is a DEVICE pointer
 *   (HBM of the GPU this rank drives); host code must not dereference it.
 *
 * The functions are implemented in src/utils_hip.c on top of the C ABI of
 * include/cg_hip.h (libcg_hip.so, hand-written HIP for gfx950).  The three
 * update loops that the reference runs inline in main (cg.c:63-87) cannot touch
 * device memory from the host; cg_hip.h provides cg_axpy / cg_xpay for them, and
 * cg_solve for the whole loop (device-resident scalars, hipGraph), which is the
 * path to use for speed.
 *
 * Decomposition: the reference cuts the grid into sqrt(P) x sqrt(P) blocks
 * (utils.c:408-417); this build cuts it into P row blocks, one GPU per rank, so
 * only contiguous rows are exchanged.  `chunklength` therefore means the ROW
 * LENGTH (n+1) and the number of local rows is settings->local_rows.  For
 * numprocs == 1 both readings coincide with the reference's.
 *
 * Structs keep the reference's members in the reference's order; new members
 * are appended.  The header compiles with or without <mpi.h>: include <mpi.h>
 * first to get the real MPI types, otherwise integer stand-ins are declared
 * (the reference's header uses the MPI types without including <mpi.h>,
 * utils.h:26-31).
 */
#ifndef CG_COMPAT_UTILS_H
#define CG_COMPAT_UTILS_H

#ifndef MAX_ITERS
#define MAX_ITERS 200          /* utils.h:1 (now overridable with -DMAX_ITERS=...) */
#endif
#ifndef UNROLL_FACT
#define UNROLL_FACT 4          /* utils.h:2 (kept for source compatibility; unused on the GPU) */
#endif

#ifndef MPI_VERSION            /* <mpi.h> not included: single-process build */
#define CG_COMPAT_NO_MPI 1
typedef int MPI_Request;
typedef int MPI_Datatype;
typedef int MPI_Comm;
typedef struct { int unused; } MPI_Status;
#endif

#ifdef __cplusplus
extern "C" {
#endif

typedef struct my_stencil {    /* utils.h:3-7 */
    int size;
    int extent;
    int* stencil;              /* host array of size*size ints (size = 3), row-wise; all nine entries are honoured (the reference
                                  reads entries 1, 3, 4, 5, 7 only, utils.c:27-31) */
} stencil_struct;

typedef struct my_d {          /* utils.h:9-16; all DEVICE pointers */
    double* left_pad;          /* column left of the chunk (two-dimensional layout), NULL at the global boundary and in row blocks */
    double* top_pad;           /* row above the block, NULL at the global boundary (utils.c:290-301) */
    double* locald;
    double* bottom_pad;        /* row below the block, NULL at the global boundary */
    double* right_pad;         /* column right of the chunk, NULL at the global boundary and in row blocks */
    double* corner_pad;        /* two-dimensional layout: the four values diagonally across the chunk's corners (up-left,
                                  up-right, down-left, down-right), filled by exchange_boundaries for nine-point coefficients;
                                  declared and never used in the reference (utils.h:15) */
} d_struct;

typedef struct my_MPI_Settings {   /* utils.h:18-32 */
    int cartsize;              /* row blocks: their number (= numprocs); two-dimensional layout: sqrt(numprocs), as utils.c:408 */
    int leftrank;              /* neighbour ranks, -1 where there is none (utils.c:413-417) */
    int toprank;
    int bottomrank;
    int rightrank;
    int* itags;
    int* jtags;
    MPI_Request* irequests;
    MPI_Request* jrequests;
    MPI_Status status;
    MPI_Datatype rowtype;
    MPI_Datatype coltype;
    MPI_Comm cartcomm;
    /* ---- appended by this build ---- */
    int device;                /* HIP device of this rank */
    int rank, numprocs;
    long local_cols;           /* columns of this rank's chunk (row blocks: the row length n+1) */
    long local_rows;           /* rows of this rank's chunk */
    long row0;                 /* global index of the first owned row */
    long col0;                 /* global index of the first owned column (0 in row blocks) */
    int layout2d;              /* 1: the reference's sqrt(P) x sqrt(P) chunks (CG_LAYOUT=2d), 0: row blocks */
    double* dev_col[2];        /* device staging for the first / last column (two-dimensional layout) */
    int strict_dot;            /* 1: dot() sums in the reference's exact order (bit-identical, slow) */
    double* host_row[2];       /* pinned staging rows for the MPI halo exchange */
} MPI_Settings;

void apply_stencil_serial(int n, stencil_struct* my_stencil, double* d, double* q);
void apply_stencil_parallel(const int chunklength, stencil_struct* my_stencil, d_struct* locald, double* localq, const int myrank, MPI_Settings* mpi_settings);
void exchange_boundaries(const int chunklength, d_struct* locald, const int myrank, MPI_Settings* mpi_settings);
d_struct* init_locald(const int n, const int chunklength, const int myrank, MPI_Settings* mpi_settings);
double* init_localg(const int chunklength, double* d);
void print_local2dmesh(const int rows, const int cols, double* mesh, const int myrank, MPI_Comm cartcomm);
void dot(const int rows, const int cols, double* localv, double* localw, MPI_Comm comm, double* result);
MPI_Settings* init_mpi_settings(int numprocs, int chunklength);
void free_struct_elems(stencil_struct* stencil, d_struct* locald, MPI_Settings* mpi_settings);

#ifdef __cplusplus
}
#endif
#endif
