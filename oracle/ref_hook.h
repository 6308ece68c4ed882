/* Declarations force-included into the reference driver when oracle/Makefile
 * builds oracle/_ref/cg_ref; see ref_hook.c.  TEST INFRASTRUCTURE ONLY. */
#ifndef REF_HOOK_H
#define REF_HOOK_H
extern int ref_max_iters;
void ref_set_qinit(double q0);
void ref_stencil(int *stencil9);
int ref_should_stop(double q1, int iter);
void ref_final(int rank, int numprocs, int n, int chunkarea,
               const double *u, const double *g, const double *d, double q1);
#endif
