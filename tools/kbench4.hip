// global_load vs buffer_load (per-row descriptors) on the same 2R1W strip-march stream (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long xcd_remap(long b, long n) { const long q = n >> 3, r = n & 7, x = b & 7, i = b >> 3; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000); }
template <int KIND, int PF, int AUX>   // KIND 0 global nt, 1 buffer (AUX = cache policy bits), 2 global plain
__global__ __launch_bounds__(256) void k_march(const double *a, const double *b, double *c, long G, long rpb, int nstrips)
{
    long tile = xcd_remap(blockIdx.x, gridDim.x);
    const long strip = tile % nstrips, band = tile / nstrips;
    const long c0 = strip * 512 + threadIdx.x * 2, r0 = band * rpb;
    const int off = (int)(c0 * 8);
    const double t = 0.5;
    for (long r = r0; r < r0 + rpb; r += PF) {
        v2d va[PF], vb[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const long k = (r + p) * G;
            if (KIND == 1) { va[p] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc(a + k, (int)(G * 8)), off, 0, AUX));
                             vb[p] = __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc(b + k, (int)(G * 8)), off, 0, AUX)); }
            else if (KIND == 0) { va[p] = __builtin_nontemporal_load((const v2d *)(a + k + c0)); vb[p] = __builtin_nontemporal_load((const v2d *)(b + k + c0)); }
            else { va[p] = *(const v2d *)(a + k + c0); vb[p] = *(const v2d *)(b + k + c0); }
        }
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const long k = (r + p) * G;
            const v2d o = t * va[p] - vb[p];
            if (KIND == 1) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rsrc(c + k, (int)(G * 8)), off, 0, AUX);
            else if (KIND == 0) __builtin_nontemporal_store(o, (v2d *)(c + k + c0));
            else *(v2d *)(c + k + c0) = o;
        }
    }
}
template <int KIND, int PF, int AUX> void run(const char *tag, double *a, double *b, double *c, long G, int nbands)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nstrips = 32; const long rpb = G / nbands;
    auto launch = [&]() { hipLaunchKernelGGL((k_march<KIND, PF, AUX>), dim3(nstrips * nbands), dim3(256), 0, 0, a, b, c, G, rpb, nstrips); };
    launch(); launch(); CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 8; ++i) launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
    printf("%-22s pf=%d bands=%3d : %7.3f ms %6.0f GB/s\n", tag, PF, nbands, ms, 3.0 * G * G * 8 / 1e9 / (ms * 1e-3)); fflush(stdout);
}
int main()
{
    const long G = 16384; const size_t bytes = (size_t)G * G * 8;
    double *a, *b, *c; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
    for (int rep = 0; rep < 2; ++rep) for (int nb : {8, 16}) {
        run<0, 4, 0>("global nt", a, b, c, G, nb);
        run<2, 4, 0>("global plain", a, b, c, G, nb);
        run<1, 4, 2>("buffer nt(aux=2)", a, b, c, G, nb);
        run<1, 4, 0>("buffer plain(aux=0)", a, b, c, G, nb);
        run<1, 4, 1>("buffer sc0(aux=1)", a, b, c, G, nb);
        run<1, 4, 3>("buffer sc0+nt(aux=3)", a, b, c, G, nb);
        run<1, 4, 16>("buffer sc1(aux=16)", a, b, c, G, nb);
        run<1, 4, 18>("buffer sc1+nt(aux=18)", a, b, c, G, nb);
        run<0, 1, 0>("global nt", a, b, c, G, nb);
        run<1, 1, 2>("buffer nt(aux=2)", a, b, c, G, nb);
    }
    return 0;
}
