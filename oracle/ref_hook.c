/*
 * Hooks linked into the build of the UNMODIFIED reference sources that
 * oracle/Makefile produces in oracle/_ref/ -- TEST INFRASTRUCTURE ONLY.
 *
 * The reference fixes its iteration count at compile time (utils.h:1), prints
 * the squared residual with only ~7 significant digits (cg.c:139) and has no
 * stop rule.  The Makefile preprocesses cg.c where it lies, rewrites three
 * statements in the preprocessor's output stream (nothing is written to disk
 * but the object file) and links this file:
 *   `iter<200;`       -> `iter<ref_max_iters;`          run-time iteration count
 *   `q0 = q1;`        -> `q0 = q1; if (ref_should_stop(q1, iter)) break;`
 *   `free(localu);`   -> `ref_final(...); free(localu);` full-precision report
 *   `memcpy(stencil->stencil, my_stencil, sizeof(my_stencil));`
 *                     -> the same followed by `ref_stencil(stencil->stencil);`
 * Environment: REF_ITERS (default 200), REF_RTOL (default 0 = never stop
 * early), REF_DUMP (prefix for raw u/g/d dumps, rank 0 of a 1-rank run only),
 * REF_STENCIL="up,left,centre,right,down" (the five entries of the 3x3 coefficient
 * array that the reference's kernels read, utils.c:27-31; default: cg.c:33's own).
 */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

int ref_max_iters = 200;
static double ref_rtol = 0.0;
static double ref_qinit = -1.0;
static int ref_iters_done = 0;

__attribute__((constructor)) static void ref_hook_init(void)
{
    const char *s = getenv("REF_ITERS");
    if (s && *s) ref_max_iters = atoi(s);
    s = getenv("REF_RTOL");
    if (s && *s) ref_rtol = atof(s);
}

/* Overrides entries 1, 3, 4, 5, 7 of the 3x3 coefficient array from REF_STENCIL. */
void ref_stencil(int *stencil9)
{
    const char *s = getenv("REF_STENCIL");
    int v[5];
    if (s && *s && sscanf(s, "%d,%d,%d,%d,%d", &v[0], &v[1], &v[2], &v[3], &v[4]) == 5) {
        stencil9[1] = v[0]; stencil9[3] = v[1]; stencil9[4] = v[2]; stencil9[5] = v[3]; stencil9[7] = v[4];
    }
}

/* Called after q0 = q1 of every iteration, on every rank with the same q1
 * (the all-reduced value), so a break is collective-safe.  The first q1 seen
 * cannot give q_init, so ref_set_qinit is called before the loop. */
void ref_set_qinit(double q0) { ref_qinit = q0; }

int ref_should_stop(double q1, int iter)
{
    ref_iters_done = iter + 1;
    return ref_rtol > 0.0 && q1 <= ref_rtol * ref_rtol * ref_qinit;
}

static void dump(const char *prefix, const char *ext, const double *a, long count)
{
    char path[4096];
    snprintf(path, sizeof path, "%s.%s.f64", prefix, ext);
    FILE *f = fopen(path, "wb");
    if (!f) return;
    fwrite(a, sizeof(double), (size_t)count, f);
    fclose(f);
}

void ref_final(int rank, int numprocs, int n, int chunkarea,
               const double *u, const double *g, const double *d, double q1)
{
    if (rank != 0) return;
    if (ref_iters_done == 0) ref_iters_done = ref_max_iters;
    printf("[ORACLE] q1 = %.17e (%a) iters=%d np=%d qinit = %a", q1, q1, ref_iters_done, numprocs,
           ref_qinit);
    if (numprocs == 1) {
        const long G = (long)n + 1;
        const double h = (double)1 / n;
        double sum_u = 0.0, max_err = 0.0;
        for (long i = 0; i < G; ++i)
            for (long j = 0; j < G; ++j) {
                const double x = i * h, y = j * h;
                const double e = fabs(u[i * G + j] - x * (1 - x) * y * (1 - y));
                sum_u += u[i * G + j];
                if (e > max_err) max_err = e;
            }
        printf(" sum_u = %.17e max_err = %.6e", sum_u, max_err);
        const char *prefix = getenv("REF_DUMP");
        if (prefix && *prefix) {
            dump(prefix, "u", u, chunkarea);
            dump(prefix, "g", g, chunkarea);
            dump(prefix, "d", d, chunkarea);
        }
    }
    putchar('\n');
}
