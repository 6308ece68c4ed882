// Does the chip-wide access window matter?  2R1W streams, 16384^2 doubles (not part of the product):
//   E  flat grid-stride kernel (all workgroups sweep one compact window through memory)
//   F  strip march, 256 workgroups, each walking chunks of R rows and then jumping ahead 8*R rows
//      (R = 2048 is the plain one-band-per-workgroup march; small R keeps the 256 workgroups within
//      8*R consecutive rows of each array at any time)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ long xcd_remap(long b, long n) { const long q = n >> 3, r = n & 7, x = b & 7, i = b >> 3; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
__device__ __forceinline__ v2d ld(const double *p) { return __builtin_nontemporal_load((const v2d *)p); }
__device__ __forceinline__ void st(double *p, v2d o) { __builtin_nontemporal_store(o, (v2d *)p); }

__global__ __launch_bounds__(256) void k_flat(const double *a, const double *b, double *c, long nvec)
{
    const long stride = (long)gridDim.x * 256;
    for (long k = (long)blockIdx.x * 256 + threadIdx.x; k < nvec; k += stride) st(c + 2 * k, 0.5 * ld(a + 2 * k) - ld(b + 2 * k));
}
template <int PF>
__global__ __launch_bounds__(256) void k_chunk(const double *a, const double *b, double *c, long G, int R, int remap)
{
    const long tile = remap ? xcd_remap(blockIdx.x, gridDim.x) : blockIdx.x;
    const long strip = tile % 32, k = tile / 32, nb = gridDim.x / 32;
    const long c0 = strip * 512 + threadIdx.x * 2;
    for (long base = k * R; base < G; base += nb * R) {
        const long r0 = base, r1 = base + R;
        v2d ra[PF], rb[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) { ra[p] = ld(a + (r0 + p) * G + c0); rb[p] = ld(b + (r0 + p) * G + c0); }
        for (long rb0 = r0; rb0 < r1; rb0 += PF) {
#pragma unroll
            for (int p = 0; p < PF; ++p) {
                const long r = rb0 + p;
                const v2d va = ra[p], vb = rb[p];
                const long rn = (r + PF < r1) ? r + PF : r;
                ra[p] = ld(a + rn * G + c0); rb[p] = ld(b + rn * G + c0);
                st(c + r * G + c0, 0.5 * va - vb);
            }
        }
    }
}
static hipEvent_t e0, e1;
template <class F> void timeit(const char *tag, long G, F launch)
{
    launch(); launch(); CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 8; ++i) launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
    printf("%-44s : %7.3f ms %6.0f GB/s\n", tag, ms, 3.0 * G * G * 8 / 1e9 / (ms * 1e-3)); fflush(stdout);
}
int main()
{
    const long G = 16384; const size_t bytes = (size_t)G * G * 8;
    double *a, *b, *c; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    char tag[128];
    for (int rep = 0; rep < 2; ++rep) {
        for (int blocks : {256, 512, 1024, 2048, 4096, 8192}) {
            snprintf(tag, sizeof tag, "E flat blocks=%d", blocks);
            timeit(tag, G, [&]() { hipLaunchKernelGGL(k_flat, dim3(blocks), dim3(256), 0, 0, a, b, c, G * G / 2); });
        }
        for (int nblk : {256, 512}) for (int remap : {0, 1}) for (int R : {2048 * 256 / nblk, 512, 128, 32, 8}) {
            snprintf(tag, sizeof tag, "F chunk march pf=2 blocks=%d remap=%d R=%d", nblk, remap, R);
            timeit(tag, G, [&]() { hipLaunchKernelGGL(k_chunk<2>, dim3(nblk), dim3(256), 0, 0, a, b, c, G, R, remap); });
        }
        for (int R : {2048, 128, 32}) {
            snprintf(tag, sizeof tag, "F chunk march pf=4 blocks=256 remap=1 R=%d", R);
            timeit(tag, G, [&]() { hipLaunchKernelGGL(k_chunk<4>, dim3(256), dim3(256), 0, 0, a, b, c, G, R, 1); });
        }
    }
    return 0;
}
