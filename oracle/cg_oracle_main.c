/*
 * Command-line front end of the CPU oracle -- TEST INFRASTRUCTURE ONLY.
 *
 *   cg_oracle <n> [iters=200] [rtol=0] [dump_prefix]
 *
 * Prints the reference's two output lines (cg.c:139-140: squared residual
 * norm as %.16lf, then the wall time of the timed region in seconds) followed
 * by a full-precision line
 *   [ORACLE] q1 = <%.17e> (<%a>) iters=<k> sum_u = <%.17e> max_err = <%.6e>
 * The timed region is the reference's (cg.c:50-51,131-132): the initial g.g
 * and all iterations, set-up excluded.  With a dump prefix the final u, g, d
 * are written as raw little-endian doubles to <prefix>.{u,g,d}.f64.
 */
#define _POSIX_C_SOURCE 200809L
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>

#include "cg_oracle.h"

static double now_s(void)
{
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double)ts.tv_sec + 1e-9 * (double)ts.tv_nsec;
}

static int dump(const char *prefix, const char *ext, const double *a, long count)
{
    char path[4096];
    snprintf(path, sizeof path, "%s.%s.f64", prefix, ext);
    FILE *f = fopen(path, "wb");
    if (!f) return -1;
    const size_t w = fwrite(a, sizeof(double), (size_t)count, f);
    fclose(f);
    return w == (size_t)count ? 0 : -1;
}

int main(int argc, char **argv)
{
    if (argc < 2) {
        fprintf(stderr, "usage: %s <n> [iters] [rtol] [dump_prefix]\n", argv[0]);
        return 2;
    }
    const long n = atol(argv[1]);
    const long iters = argc > 2 ? atol(argv[2]) : 200;
    const double rtol = argc > 3 ? atof(argv[3]) : 0.0;
    const char *prefix = argc > 4 ? argv[4] : NULL;
    if (n < 2 || iters < 0) {
        fprintf(stderr, "[ERROR] need n >= 2 and iters >= 0\n");
        return 1;
    }
    const long G = n + 1, N = G * G;
    double *u = calloc((size_t)N, sizeof(double));
    double *q = calloc((size_t)N, sizeof(double));
    double *d = malloc((size_t)N * sizeof(double));
    double *g = malloc((size_t)N * sizeof(double));
    if (!u || !q || !d || !g) {
        fprintf(stderr, "[ERROR] out of memory for a %ldx%ld grid\n", G, G);
        return 1;
    }
    oracle_init_d(n, 0, G, d);
    oracle_init_g(N, d, g);

    const double t0 = now_s();
    double q0 = oracle_dot(N, g, g), q1 = q0;
    const double stop = rtol * rtol * q0;
    long it = 0;
    for (; it < iters; ++it) {
        oracle_stencil(G, 0, G, d, NULL, NULL, q);
        const double tau = q0 / oracle_dot(N, d, q);
        oracle_axpy(N, tau, d, u);
        oracle_axpy(N, tau, q, g);
        q1 = oracle_dot(N, g, g);
        const double beta = q1 / q0;
        oracle_xpay(N, beta, d, g);
        q0 = q1;
        if (rtol > 0 && q1 <= stop) { ++it; break; }
    }
    const double runtime = now_s() - t0;

    double sum_u = 0.0, max_err = 0.0;
    const double h = (double)1 / (double)n;
    for (long i = 0; i < G; ++i) {
        const double x = (double)i * h;
        for (long j = 0; j < G; ++j) {
            const double y = (double)j * h;
            const double e = fabs(u[i * G + j] - x * (1 - x) * y * (1 - y));
            sum_u += u[i * G + j];
            if (e > max_err) max_err = e;
        }
    }
    printf("[INFO] norm_g^2 = %.16lf\n", q1);
    printf("%.16lf\n", runtime);
    printf("[ORACLE] q1 = %.17e (%a) iters=%ld sum_u = %.17e max_err = %.6e\n",
           q1, q1, it, sum_u, max_err);
    int rc = 0;
    if (prefix)
        rc = dump(prefix, "u", u, N) | dump(prefix, "g", g, N) | dump(prefix, "d", d, N);
    free(u); free(q); free(d); free(g);
    return rc ? 1 : 0;
}
