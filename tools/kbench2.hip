// Address-pattern microbenchmark (not part of the product): a 2R1W / 3R2W stream over G x G fp64
// arrays where each workgroup marches down a band of rows of a column strip, as the stencil kernels
// do.  Varies strip width, band height (= number of workgroups) and rows in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));

__device__ __forceinline__ long xcd_remap(long b, long n) { const long q = n >> 3, r = n & 7, x = b & 7, i = b >> 3; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }

// SEG = double2 per thread per row (strip width = 512*SEG columns); PF = rows in flight
template <int MODE, int SEG, int PF>
__global__ __launch_bounds__(256) void k_march(v2d *a, v2d *b, v2d *c, v2d *d2, v2d *e2, double *sink, long G2, long rows_per_band, int nstrips, int remap, int order)
{
    long tile = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    long strip, band;
    if (order == 0) { strip = tile % nstrips; band = tile / nstrips; }      // strips fastest
    else { const long nb = gridDim.x / nstrips; band = tile % nb; strip = tile / nb; }   // bands fastest
    const long c0 = strip * 256 * SEG + threadIdx.x;
    const long r0 = band * rows_per_band;
    const double t = 0.5;
    v2d accv = {0.0, 0.0};
    for (long r = r0; r < r0 + rows_per_band; r += PF) {
        v2d va[PF][SEG], vb[PF][SEG], vc[PF][SEG];
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int s = 0; s < SEG; ++s) {
                const long k = (r + p) * G2 + c0 + s * 256;
                va[p][s] = __builtin_nontemporal_load(a + k);
                vb[p][s] = __builtin_nontemporal_load(b + k);
                if (MODE == 1 || MODE == 2) vc[p][s] = __builtin_nontemporal_load(c + k);
            }
#pragma unroll
        for (int p = 0; p < PF; ++p)
#pragma unroll
            for (int s = 0; s < SEG; ++s) {
                const long k = (r + p) * G2 + c0 + s * 256;
                if (MODE == 0) { __builtin_nontemporal_store(t * va[p][s] - vb[p][s], c + k); }      // 2R1W to a third array
                else if (MODE == 3) { accv += va[p][s] * vb[p][s]; }
                else if (MODE == 2) { __builtin_nontemporal_store(t * va[p][s] - vb[p][s], d2 + k); __builtin_nontemporal_store(vb[p][s] + t * vc[p][s], e2 + k); __builtin_nontemporal_store(vc[p][s] + t * va[p][s], c + k); }
                else { __builtin_nontemporal_store(va[p][s] + t * vc[p][s], a + k); __builtin_nontemporal_store(vb[p][s] + t * vc[p][s], b + k); }
            }
    }
    if (MODE == 3 && accv.x + accv.y == 123.456) sink[0] = accv.x;
}

static v2d *g_d2, *g_e2; static double *g_sink;
template <int MODE, int SEG, int PF>
void run(v2d *a, v2d *b, v2d *c, long G, int nblocks, int remap, int order)
{
    const long G2 = G / 2;
    const int nstrips = (int)(G2 / (256 * SEG));
    const int nbands = nblocks / nstrips;
    if (nbands < 1 || G % nbands) return;
    const long rpb = G / nbands;
    if (rpb % PF) return;
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL((k_march<MODE, SEG, PF>), dim3(nstrips * nbands), dim3(256), 0, 0, a, b, c, g_d2, g_e2, g_sink, G2, rpb, nstrips, remap, order); };
    launch(); launch();
    CHECK(hipEventRecord(e0, 0));
    for (int i = 0; i < 8; ++i) launch();
    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
    const double gb = (MODE == 0 ? 3.0 : MODE == 1 ? 5.0 : MODE == 2 ? 6.0 : 2.0) * G * G * 8 / 1e9;
    printf("%s seg=%d(width %5d) pf=%d blocks=%5d (strips %3d x bands %4d) remap=%d order=%d : %7.3f ms %6.0f GB/s\n", MODE == 0 ? "2r1w" : MODE == 1 ? "3r2w" : MODE == 2 ? "3r3w" : "2r0w", SEG, 512 * SEG, PF, nstrips * nbands, nstrips, nbands, remap, order, ms, gb / ms / 1e-3 / 1e0 * 1e-0);
    fflush(stdout);
}

template <int MODE, int SEG, int PF>
void sweep(v2d *a, v2d *b, v2d *c, long G)
{
    for (int nb : {256, 512, 1024, 2048, 4096, 8192})
        for (int remap : {1, 0}) run<MODE, SEG, PF>(a, b, c, G, nb, remap, 0);
    run<MODE, SEG, PF>(a, b, c, G, 256, 0, 1);
    run<MODE, SEG, PF>(a, b, c, G, 1024, 0, 1);
    run<MODE, SEG, PF>(a, b, c, G, 1024, 1, 1);
}

int main(int argc, char **argv)
{
    const long G = argc > 1 ? atol(argv[1]) : 16384;
    v2d *a, *b, *c; const size_t bytes = (size_t)G * G * 8;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes));
    CHECK(hipMalloc(&g_d2, bytes)); CHECK(hipMalloc(&g_e2, bytes)); CHECK(hipMalloc(&g_sink, 64));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
    sweep<2, 1, 1>(a, b, c, G); sweep<2, 1, 2>(a, b, c, G); sweep<2, 4, 1>(a, b, c, G);
    sweep<3, 1, 1>(a, b, c, G); sweep<3, 1, 4>(a, b, c, G); sweep<3, 4, 2>(a, b, c, G);
    return 0;
}
