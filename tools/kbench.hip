// Stand-alone HBM streaming microbenchmark used to choose the launch geometry of the
// CG kernels (not part of the product).  Build: hipcc --offload-arch=gfx950 -O3 tools/kbench.hip -o tools/kbench
// Prints GB/s of algorithmic bytes for read / copy / 2R1W / 4R2W streams over 2 GiB arrays
// under different grid sizes, unroll factors, block->address mappings and cache policies.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT> __device__ __forceinline__ double2 ld(const double2 *p)
{
    if (NT) { double2 r; r.x = __builtin_nontemporal_load(&p->x); r.y = __builtin_nontemporal_load(&p->y); return r; }
    return *p;
}
template <bool NT> __device__ __forceinline__ void st(double2 *p, double2 v)
{
    if (NT) { __builtin_nontemporal_store(v.x, &p->x); __builtin_nontemporal_store(v.y, &p->y); }
    else *p = v;
}

// MODE 0: read-only sum; 1: copy; 2: d = b*d - g (2R1W); 3: u += t*d, g += t*q (4R2W)
template <int MODE, int UNROLL, bool NTL, bool NTS, int BLOCK>
__global__ __launch_bounds__(BLOCK) void k_stream(double2 *a, double2 *b, const double2 *c, const double2 *d,
                                                  long nvec, int chunked, double *sink)
{
    const long per_iter = (long)BLOCK * UNROLL;
    long start, step, end;
    if (chunked) {
        const long chunk = ((nvec + gridDim.x - 1) / gridDim.x + per_iter - 1) / per_iter * per_iter;
        start = (long)blockIdx.x * chunk; end = start + chunk; if (end > nvec) end = nvec; step = per_iter;
    } else {
        start = (long)blockIdx.x * per_iter; end = nvec; step = (long)gridDim.x * per_iter;
    }
    double acc = 0.0;
    const double t = 0.5;
    for (long base = start; base < end; base += step) {
        double2 va[UNROLL], vb[UNROLL], vc[UNROLL], vd[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long k = base + (long)u * BLOCK + threadIdx.x;
            if (k < end) {
                va[u] = ld<NTL>(a + k);
                if (MODE >= 2) vb[u] = ld<NTL>(b + k);
                if (MODE == 3) { vc[u] = ld<NTL>(c + k); vd[u] = ld<NTL>(d + k); }
            }
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            const long k = base + (long)u * BLOCK + threadIdx.x;
            if (k < end) {
                if (MODE == 0) acc += va[u].x + va[u].y;
                if (MODE == 1) st<NTS>(b + k, va[u]);
                if (MODE == 2) { va[u].x = t * va[u].x - vb[u].x; va[u].y = t * va[u].y - vb[u].y; st<NTS>(a + k, va[u]); }
                if (MODE == 3) {
                    va[u].x += t * vc[u].x; va[u].y += t * vc[u].y;
                    vb[u].x += t * vd[u].x; vb[u].y += t * vd[u].y;
                    acc += vb[u].x * vb[u].x + vb[u].y * vb[u].y;
                    st<NTS>(a + k, va[u]); st<NTS>(b + k, vb[u]);
                }
            }
        }
    }
    if (MODE == 0 || MODE == 3) if (acc == 123.456) sink[0] = acc;
}

struct Bufs { double2 *a, *b, *c, *d; double *sink; long nvec; };

template <int MODE, int UNROLL, bool NTL, bool NTS, int BLOCK>
void run(const Bufs &B, int grid, int chunked, const char *tag)
{
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    auto launch = [&]() { hipLaunchKernelGGL((k_stream<MODE, UNROLL, NTL, NTS, BLOCK>), dim3(grid), dim3(BLOCK), 0, 0, B.a, B.b, B.c, B.d, B.nvec, chunked, B.sink); };
    launch(); launch();
    CHECK(hipEventRecord(e0, 0));
    const int reps = 8;
    for (int i = 0; i < reps; ++i) launch();
    CHECK(hipEventRecord(e1, 0));
    CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
    const double streams[4] = {1, 2, 3, 6};
    const double gb = streams[MODE] * B.nvec * 16.0 / 1e9;
    printf("%-8s mode=%d unroll=%d ntl=%d nts=%d block=%4d grid=%6d chunked=%d : %7.3f ms %7.0f GB/s\n", tag, MODE, UNROLL, (int)NTL, (int)NTS, BLOCK, grid, chunked, ms, gb / (ms * 1e-3));
    fflush(stdout);
    CHECK(hipEventDestroy(e0)); CHECK(hipEventDestroy(e1));
}

template <int MODE, int UNROLL, bool NTL, bool NTS>
void sweep_grid(const Bufs &B, const char *tag)
{
    for (int grid : {256, 512, 1024, 2048, 4096, 16384}) {
        run<MODE, UNROLL, NTL, NTS, 256>(B, grid, 0, tag);
    }
    run<MODE, UNROLL, NTL, NTS, 256>(B, 512, 1, tag);
    run<MODE, UNROLL, NTL, NTS, 256>(B, 2048, 1, tag);
    run<MODE, UNROLL, NTL, NTS, 512>(B, 512, 0, tag);
    run<MODE, UNROLL, NTL, NTS, 1024>(B, 256, 0, tag);
    run<MODE, UNROLL, NTL, NTS, 1024>(B, 512, 0, tag);
}

int main(int argc, char **argv)
{
    const long G = argc > 1 ? atol(argv[1]) : 16384;
    Bufs B; B.nvec = G * G / 2;
    const size_t bytes = (size_t)B.nvec * 16;
    CHECK(hipMalloc(&B.a, bytes)); CHECK(hipMalloc(&B.b, bytes)); CHECK(hipMalloc(&B.c, bytes)); CHECK(hipMalloc(&B.d, bytes));
    CHECK(hipMalloc(&B.sink, 64));
    CHECK(hipMemset(B.a, 0, bytes)); CHECK(hipMemset(B.b, 0, bytes)); CHECK(hipMemset(B.c, 0, bytes)); CHECK(hipMemset(B.d, 0, bytes));
    printf("== read ==\n");
    sweep_grid<0, 1, false, false>(B, "read");
    sweep_grid<0, 4, false, false>(B, "read");
    sweep_grid<0, 4, true, false>(B, "read");
    printf("== copy ==\n");
    sweep_grid<1, 1, false, false>(B, "copy");
    sweep_grid<1, 4, false, false>(B, "copy");
    sweep_grid<1, 4, true, true>(B, "copy");
    sweep_grid<1, 4, false, true>(B, "copy");
    printf("== 2R1W ==\n");
    sweep_grid<2, 1, false, false>(B, "2r1w");
    sweep_grid<2, 2, false, false>(B, "2r1w");
    sweep_grid<2, 4, false, false>(B, "2r1w");
    sweep_grid<2, 4, true, true>(B, "2r1w");
    printf("== 4R2W ==\n");
    sweep_grid<3, 1, false, false>(B, "4r2w");
    sweep_grid<3, 2, false, false>(B, "4r2w");
    sweep_grid<3, 4, false, false>(B, "4r2w");
    sweep_grid<3, 2, true, true>(B, "4r2w");
    sweep_grid<3, 2, false, true>(B, "4r2w");
    return 0;
}
