// Does the NUMBER of concurrent streams matter at equal bytes?  Strip march, 256 workgroups, 16384^2 points
// (not part of the product).  Per point 16 B are read and 16 B written, either as
//   S  two read arrays + two write arrays of doubles (2R2W, 16 B per lane per array), or
//   I  one read array + one write array of interleaved pairs (1R1W, 32 B per lane),
// and the same with a third read+write pair (3R3W against 1R1W of 24 B... here: 48 B per lane as 3 x 16).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
__device__ __forceinline__ long xcd_remap(long b, long n) { const long q = n >> 3, r = n & 7, x = b & 7, i = b >> 3; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
__device__ __forceinline__ v2d ld(const double *p) { return __builtin_nontemporal_load((const v2d *)p); }
__device__ __forceinline__ void st(double *p, v2d o) { __builtin_nontemporal_store(o, (v2d *)p); }

// NS separate streams of G doubles per row
template <int NS, int PF>
__global__ __launch_bounds__(256) void k_sep(const double *const *in, double *const *out, long G, long rpb)
{
    const long tile = xcd_remap(blockIdx.x, gridDim.x);
    const long strip = tile % 32, band = tile / 32;
    const long c0 = strip * 512 + threadIdx.x * 2, r0 = band * rpb, r1 = r0 + rpb;
    const double *a[NS]; double *o[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) { a[s] = in[s]; o[s] = out[s]; }
    v2d ring[PF][NS];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < NS; ++s) ring[p][s] = ld(a[s] + (r0 + p) * G + c0);
    for (long rb = r0; rb < r1; rb += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const long r = rb + p, rn = (r + PF < r1) ? r + PF : r;
            v2d v[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { v[s] = ring[p][s]; ring[p][s] = ld(a[s] + rn * G + c0); }
#pragma unroll
            for (int s = 0; s < NS; ++s) st(o[s] + r * G + c0, 0.5 * v[s] - v[(s + 1) % NS]);
        }
    }
}
// one stream of NS*G doubles per row, point-interleaved (lane owns 2 points = 2*NS doubles, contiguous)
template <int NS, int PF>
__global__ __launch_bounds__(256) void k_int(const double *in, double *out, long G, long rpb)
{
    const long tile = xcd_remap(blockIdx.x, gridDim.x);
    const long strip = tile % 32, band = tile / 32;
    const long c0 = (strip * 512 + threadIdx.x * 2) * NS, r0 = band * rpb, r1 = r0 + rpb, RG = G * NS;
    v2d ring[PF][NS];
#pragma unroll
    for (int p = 0; p < PF; ++p)
#pragma unroll
        for (int s = 0; s < NS; ++s) ring[p][s] = ld(in + (r0 + p) * RG + c0 + 2 * s);
    for (long rb = r0; rb < r1; rb += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const long r = rb + p, rn = (r + PF < r1) ? r + PF : r;
            v2d v[NS];
#pragma unroll
            for (int s = 0; s < NS; ++s) { v[s] = ring[p][s]; ring[p][s] = ld(in + rn * RG + c0 + 2 * s); }
#pragma unroll
            for (int s = 0; s < NS; ++s) st(out + r * RG + c0 + 2 * s, 0.5 * v[s] - v[(s + 1) % NS]);
        }
    }
}
static hipEvent_t e0, e1;
template <class F> void timeit(const char *tag, double gb, F launch)
{
    launch(); launch(); CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 8; ++i) launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
    printf("%-40s : %7.3f ms %6.0f GB/s\n", tag, ms, gb / (ms * 1e-3)); fflush(stdout);
}
int main()
{
    const long G = 16384; const size_t bytes = (size_t)G * G * 8;
    double *buf[6];
    for (auto &b : buf) { CHECK(hipMalloc(&b, bytes)); CHECK(hipMemset(b, 0, bytes)); }
    double *big_in, *big_out; CHECK(hipMalloc(&big_in, 3 * bytes)); CHECK(hipMalloc(&big_out, 3 * bytes));
    CHECK(hipMemset(big_in, 0, 3 * bytes)); CHECK(hipMemset(big_out, 0, 3 * bytes));
    const double **din; double **dout; CHECK(hipMalloc(&din, 3 * sizeof(void *))); CHECK(hipMalloc(&dout, 3 * sizeof(void *)));
    const double *hin[3] = {buf[0], buf[1], buf[2]}; double *hout[3] = {buf[3], buf[4], buf[5]};
    CHECK(hipMemcpy(din, hin, sizeof hin, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dout, hout, sizeof hout, hipMemcpyHostToDevice));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const long rpb = G / 8; const double pt = (double)G * G * 8 / 1e9;
    for (int rep = 0; rep < 2; ++rep) {
        timeit("S 1R1W  8 B/pt  pf2", 2 * pt, [&]() { hipLaunchKernelGGL((k_sep<1, 2>), dim3(256), dim3(256), 0, 0, din, dout, G, rpb); });
        timeit("S 1R1W  8 B/pt  pf4", 2 * pt, [&]() { hipLaunchKernelGGL((k_sep<1, 4>), dim3(256), dim3(256), 0, 0, din, dout, G, rpb); });
        timeit("S 2R2W 16 B/pt  pf1", 4 * pt, [&]() { hipLaunchKernelGGL((k_sep<2, 1>), dim3(256), dim3(256), 0, 0, din, dout, G, rpb); });
        timeit("S 2R2W 16 B/pt  pf2", 4 * pt, [&]() { hipLaunchKernelGGL((k_sep<2, 2>), dim3(256), dim3(256), 0, 0, din, dout, G, rpb); });
        timeit("I 1R1W 16 B/pt  pf1", 4 * pt, [&]() { hipLaunchKernelGGL((k_int<2, 1>), dim3(256), dim3(256), 0, 0, big_in, big_out, G, rpb); });
        timeit("I 1R1W 16 B/pt  pf2", 4 * pt, [&]() { hipLaunchKernelGGL((k_int<2, 2>), dim3(256), dim3(256), 0, 0, big_in, big_out, G, rpb); });
        timeit("S 3R3W 24 B/pt  pf1", 6 * pt, [&]() { hipLaunchKernelGGL((k_sep<3, 1>), dim3(256), dim3(256), 0, 0, din, dout, G, rpb); });
        timeit("S 3R3W 24 B/pt  pf2", 6 * pt, [&]() { hipLaunchKernelGGL((k_sep<3, 2>), dim3(256), dim3(256), 0, 0, din, dout, G, rpb); });
        timeit("I 1R1W 24 B/pt  pf1", 6 * pt, [&]() { hipLaunchKernelGGL((k_int<3, 1>), dim3(256), dim3(256), 0, 0, big_in, big_out, G, rpb); });
        timeit("I 1R1W 24 B/pt  pf2", 6 * pt, [&]() { hipLaunchKernelGGL((k_int<3, 2>), dim3(256), dim3(256), 0, 0, big_in, big_out, G, rpb); });
    }
    return 0;
}
