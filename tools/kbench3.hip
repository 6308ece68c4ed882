// Does the relative placement of the streamed arrays matter?  2R1W / 3R2W strip-march streams with the
// arrays carved out of one slab at (2 GiB + pad) spacing (not part of the product).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ long xcd_remap(long b, long n) { const long q = n >> 3, r = n & 7, x = b & 7, i = b >> 3; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
template <int MODE, int PF>
__global__ __launch_bounds__(256) void k_march(v2d *a, v2d *b, v2d *c, v2d *d, v2d *e, long G2, long rpb, int nstrips)
{
    long tile = xcd_remap(blockIdx.x, gridDim.x);
    const long strip = tile % nstrips, band = tile / nstrips;
    const long c0 = strip * 256 + threadIdx.x, r0 = band * rpb;
    const double t = 0.5;
    for (long r = r0; r < r0 + rpb; r += PF) {
        v2d va[PF], vb[PF], vc[PF];
#pragma unroll
        for (int p = 0; p < PF; ++p) { const long k = (r + p) * G2 + c0; va[p] = __builtin_nontemporal_load(a + k); vb[p] = __builtin_nontemporal_load(b + k); if (MODE == 1) vc[p] = __builtin_nontemporal_load(c + k); }
#pragma unroll
        for (int p = 0; p < PF; ++p) { const long k = (r + p) * G2 + c0;
            if (MODE == 0) __builtin_nontemporal_store(t * va[p] - vb[p], c + k);
            else { __builtin_nontemporal_store(va[p] + t * vc[p], d + k); __builtin_nontemporal_store(vb[p] + t * vc[p], e + k); } }
    }
}
int main()
{
    const long G = 16384; const size_t bytes = (size_t)G * G * 8;
    char *slab; CHECK(hipMalloc(&slab, 5 * bytes + (64u << 20))); CHECK(hipMemset(slab, 0, 5 * bytes + (64u << 20)));
    const long pads[] = {0, 256, 4096, 65536, 1 << 20, (1 << 20) + 4096, (2 << 20) + 65536 + 4096, 3 * 1048576 + 3 * 4096 + 768, 5 * 1048576 + 7 * 65536};
    for (long pad : pads) for (int mode = 0; mode < 2; ++mode) {
        v2d *p[5]; for (int i = 0; i < 5; ++i) p[i] = (v2d *)(slab + i * (bytes + pad));
        hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
        const int nstrips = 32, nbands = 8; const long rpb = G / nbands;
        auto launch = [&]() { if (mode == 0) hipLaunchKernelGGL((k_march<0, 4>), dim3(256), dim3(256), 0, 0, p[0], p[1], p[2], p[3], p[4], G / 2, rpb, nstrips);
                              else hipLaunchKernelGGL((k_march<1, 4>), dim3(256), dim3(256), 0, 0, p[0], p[1], p[2], p[0], p[1], G / 2, rpb, nstrips); };
        launch(); launch(); CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 8; ++i) launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
        float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
        printf("%s pad=%9ld : %7.3f ms %6.0f GB/s\n", mode == 0 ? "2r1w" : "3r2w(in place)", pad, ms, (mode == 0 ? 3.0 : 5.0) * bytes / 1e9 / (ms * 1e-3)); fflush(stdout);
    }
    return 0;
}
