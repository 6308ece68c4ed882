// Does a region of HBM that was just read come back faster (the 256 MiB memory-side cache)?  Stand-alone microbenchmark
// (not part of the product).  Build: hipcc --offload-arch=gfx950 -O3 tools/kbench8.hip -o tools/kbench8
// A reader kernel sums a region of a 8 GiB buffer; timed: the same region again (warm) against a region far away (cold),
// for region sizes from 32 MiB to 1 GiB, with plain and with non-temporal loads in either kernel.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

template <bool NT>
__global__ __launch_bounds__(256) void k_read(const double2 *a, long nvec, double *sink)
{
    double acc = 0.0;
    const long step = (long)gridDim.x * 256 * 4;
    for (long base = (long)blockIdx.x * 256 * 4; base < nvec; base += step) {
        double2 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const long k = base + u * 256 + threadIdx.x;
            if (k < nvec) {
                if (NT) { v[u].x = __builtin_nontemporal_load(&a[k].x); v[u].y = __builtin_nontemporal_load(&a[k].y); }
                else v[u] = a[k];
            } else v[u] = make_double2(0.0, 0.0);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc += v[u].x + v[u].y;
    }
    if (acc == 1.2345e300) *sink = acc;
}

int main()
{
    const size_t total = 8ull << 30;
    double2 *buf; double *sink;
    CHECK(hipMalloc(&buf, total));
    CHECK(hipMalloc(&sink, 8));
    CHECK(hipMemset(buf, 0, total));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const int grid = 2048;
    for (int warm_nt = 0; warm_nt < 2; ++warm_nt)
        for (int timed_nt = 0; timed_nt < 2; ++timed_nt)
            for (size_t mb = 32; mb <= 1024; mb *= 2) {
                const long nvec = (long)(mb << 20) / 16;
                const double2 *r0 = buf, *far = buf + (4ull << 30) / 16;
                float ms_warm = 0.f, ms_cold = 0.f;
                for (int rep = 0; rep < 5; ++rep) {
                    // warm case: read r0 (untimed), then r0 again (timed)
                    if (warm_nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, r0, nvec, sink);
                    else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, r0, nvec, sink);
                    CHECK(hipEventRecord(e0, 0));
                    if (timed_nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, r0, nvec, sink);
                    else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, r0, nvec, sink);
                    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                    float t; CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep == 0 || t < ms_warm) ms_warm = t;
                    // cold case: read a far region of 1 GiB (evicts), then r0 (timed)
                    hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, far, (long)(1ull << 30) / 16, sink);
                    CHECK(hipEventRecord(e0, 0));
                    if (timed_nt) hipLaunchKernelGGL(k_read<true>, dim3(grid), dim3(256), 0, 0, r0, nvec, sink);
                    else hipLaunchKernelGGL(k_read<false>, dim3(grid), dim3(256), 0, 0, r0, nvec, sink);
                    CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1));
                    CHECK(hipEventElapsedTime(&t, e0, e1)); if (rep == 0 || t < ms_cold) ms_cold = t;
                }
                printf("first read %s, timed read %s, %4zu MiB: warm %7.1f us (%6.0f GB/s)   cold %7.1f us (%6.0f GB/s)\n", warm_nt ? "nt   " : "plain",
                       timed_nt ? "nt   " : "plain", mb, ms_warm * 1e3, (mb << 20) / (ms_warm * 1e6), ms_cold * 1e3, (mb << 20) / (ms_cold * 1e6));
                fflush(stdout);
            }
    return 0;
}
