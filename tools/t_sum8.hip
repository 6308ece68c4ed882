// micro test of cg::wave_sum8: lane l holds v[p] = 1000*p + l; expected sum over lanes = 64000*p + 2016 in lanes 8p..8p+7
#include <hip/hip_runtime.h>
#include <cstdio>
#include "../conjugate-gradient_amd/csrc/cg_kernels.hpp"
__global__ void k(double *out, float *outf)
{
    double v[8]; float f[8];
    for (int p = 0; p < 8; ++p) { v[p] = 1000.0 * p + threadIdx.x; f[p] = 4096.0f * p + (float)(threadIdx.x * threadIdx.x); }
    out[threadIdx.x] = cg::wave_sum8(v);
    double fd[8]; for (int p = 0; p < 8; ++p) fd[p] = f[p]; outf[threadIdx.x] = (float)cg::wave_sum8(fd);
}
int main()
{
    double *d, h[64]; float *df, hf[64];
    hipMalloc(&d, sizeof h); hipMalloc(&df, sizeof hf);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, df);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost); hipMemcpy(hf, df, sizeof hf, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const int p = l >> 3;
        if (h[l] != 64000.0 * p + 2016.0 || hf[l] != 4096.0f * 64 * p + 85344.0f) { ++bad; printf("lane %d: %g %g (row %g?)\n", l, h[l], hf[l], (h[l] - 2016.0) / 64000.0); }
    }
    printf(bad ? "FAILED\n" : "wave_sum8 ok\n");
    return bad != 0;
}
