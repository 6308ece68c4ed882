// Which feature of the stencil kernel's loop structure costs HBM efficiency?  2R1W strip-march variants
// (not part of the product): A plain load/store rows; B software-pipelined (row r+PF loaded before row r is
// stored); C = B + one LDS exchange and workgroup barrier per row; D = C with buffer loads/stores.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef double v2d __attribute__((ext_vector_type(2)));
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
__device__ __forceinline__ long xcd_remap(long b, long n) { const long q = n >> 3, r = n & 7, x = b & 7, i = b >> 3; return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + i; }
__device__ __forceinline__ __amdgpu_buffer_rsrc_t rsrc(const void *p, int bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ void lds_barrier() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local"); __builtin_amdgcn_s_barrier(); __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local"); }
template <bool BUF> __device__ __forceinline__ v2d ld(const double *row, long c0, int G8) {
    if (BUF) return __builtin_bit_cast(v2d, __builtin_amdgcn_raw_buffer_load_b128(rsrc(row, G8), (int)(c0 * 8), 0, 2));
    return __builtin_nontemporal_load((const v2d *)(row + c0)); }
template <bool BUF> __device__ __forceinline__ void st(double *row, long c0, int G8, v2d o) {
    if (BUF) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, o), rsrc(row, G8), (int)(c0 * 8), 0, 2);
    else __builtin_nontemporal_store(o, (v2d *)(row + c0)); }

template <int VAR, int PF>
__global__ __launch_bounds__(256) void k_march(const double *a, const double *b, double *c, long G, long rpb, int nstrips)
{
    constexpr bool BUF = (VAR == 3);
    __shared__ double L[2][514];
    long tile = xcd_remap(blockIdx.x, gridDim.x);
    const long strip = tile % nstrips, band = tile / nstrips;
    const long c0 = strip * 512 + threadIdx.x * 2, r0 = band * rpb, r1 = r0 + rpb;
    const int G8 = (int)(G * 8);
    const double t = 0.5;
    if (VAR == 0) {
        for (long r = r0; r < r1; r += PF) {
            v2d va[PF], vb[PF];
#pragma unroll
            for (int p = 0; p < PF; ++p) { va[p] = ld<false>(a + (r + p) * G, c0, G8); vb[p] = ld<false>(b + (r + p) * G, c0, G8); }
#pragma unroll
            for (int p = 0; p < PF; ++p) st<false>(c + (r + p) * G, c0, G8, t * va[p] - vb[p]);
        }
        return;
    }
    v2d ra[PF], rb[PF];
#pragma unroll
    for (int p = 0; p < PF; ++p) { ra[p] = ld<BUF>(a + (r0 + p) * G, c0, G8); rb[p] = ld<BUF>(b + (r0 + p) * G, c0, G8); }
    int buf = 0;
    for (long rb0 = r0; rb0 < r1; rb0 += PF) {
#pragma unroll
        for (int p = 0; p < PF; ++p) {
            const long r = rb0 + p;
            const v2d va = ra[p], vb = rb[p];
            const long rn = (r + PF < r1) ? r + PF : r;       // refill (last rows re-read themselves: harmless)
            ra[p] = ld<BUF>(a + rn * G, c0, G8); rb[p] = ld<BUF>(b + rn * G, c0, G8);
            v2d o = t * va - vb;
            if (VAR >= 2) {
                double *row = L[buf];
                row[1 + threadIdx.x * 2] = o.x; row[2 + threadIdx.x * 2] = o.y;
                lds_barrier();
                o.x += 1e-30 * row[threadIdx.x * 2]; o.y += 1e-30 * row[threadIdx.x * 2 + 3];
                buf ^= 1;
            }
            st<BUF>(c + r * G, c0, G8, o);
        }
    }
}
template <int VAR, int PF> void run(const char *tag, double *a, double *b, double *c, long G, int nbands)
{
    hipEvent_t e0, e1; CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int nstrips = 32; const long rpb = G / nbands;
    auto launch = [&]() { hipLaunchKernelGGL((k_march<VAR, PF>), dim3(nstrips * nbands), dim3(256), 0, 0, a, b, c, G, rpb, nstrips); };
    launch(); launch(); CHECK(hipEventRecord(e0, 0)); for (int i = 0; i < 8; ++i) launch(); CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
    float ms; CHECK(hipEventElapsedTime(&ms, e0, e1)); ms /= 8;
    printf("%-34s pf=%d blocks=%4d : %7.3f ms %6.0f GB/s\n", tag, PF, nstrips * nbands, ms, 3.0 * G * G * 8 / 1e9 / (ms * 1e-3)); fflush(stdout);
}
__global__ void k_fill(double *p, size_t n, unsigned seed)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long x = (i + seed) * 0x9E3779B97F4A7C15ull; x ^= x >> 29; x *= 0xBF58476D1CE4E5B9ull; x ^= x >> 32;
        p[i] = (double)(x >> 11) * (1.0 / 9007199254740992.0) - 0.5;      // random mantissa bits
    }
}
int main(int argc, char **argv)
{
    const bool rnd = argc > 1 && argv[1][0] == 'r';      // "rand": random operand bits instead of zeros
    const long G = 16384; const size_t bytes = (size_t)G * G * 8;
    double *a, *b, *c; CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&c, bytes));
    CHECK(hipMemset(a, 0, bytes)); CHECK(hipMemset(b, 0, bytes)); CHECK(hipMemset(c, 0, bytes));
    if (rnd) { k_fill<<<4096, 256>>>(a, bytes / 8, 1u); k_fill<<<4096, 256>>>(b, bytes / 8, 77u); CHECK(hipDeviceSynchronize()); }
    printf("operands: %s\n", rnd ? "random" : "zeros");
    for (int nb : {8}) {
        run<0, 1>("A batch load/store", a, b, c, G, nb); run<0, 2>("A batch load/store", a, b, c, G, nb); run<0, 4>("A batch load/store", a, b, c, G, nb);
        run<0, 8>("A batch load/store", a, b, c, G, nb); run<0, 16>("A batch load/store", a, b, c, G, nb); run<0, 32>("A batch load/store", a, b, c, G, nb);
        run<1, 1>("B pipelined", a, b, c, G, nb); run<1, 2>("B pipelined", a, b, c, G, nb); run<1, 4>("B pipelined", a, b, c, G, nb);
        run<2, 1>("C pipelined+LDS+barrier", a, b, c, G, nb); run<2, 2>("C pipelined+LDS+barrier", a, b, c, G, nb); run<2, 4>("C pipelined+LDS+barrier", a, b, c, G, nb);
        run<3, 1>("D = C with buffer ops", a, b, c, G, nb); run<3, 2>("D = C with buffer ops", a, b, c, G, nb); run<3, 4>("D = C with buffer ops", a, b, c, G, nb);
    }
    return 0;
}
