// Issue cost of single vector instructions on gfx950, in cycles per wave-instruction per SIMD:
//   hipcc --offload-arch=gfx950 -O3 tools/ibench.hip -o tools/ibench && tools/ibench
// Every wavefront runs eight independent chains of the instruction under test; with W wavefronts per SIMD
// resident the SIMD's issue rate is what is measured, not a chain's latency.  Cycles are taken from
// s_memtime inside the kernel (shader clock), so the chip's clock under load does not enter.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)

template <int OP>
__global__ __launch_bounds__(256) void k_bench(double *out, long *cyc, int iters, double seed)
{
    double v[8], w[8];
    float f[8];
    int n[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { v[i] = seed + threadIdx.x * 1e-3 + i; w[i] = 1.0 + 1e-9 * i; f[i] = (float)v[i]; n[i] = threadIdx.x + i; }
    __shared__ double lds[256 * 9];
    double *L = lds + threadIdx.x;
    const long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 3) asm volatile("v_rndne_f64 %0, %0" : "+v"(v[i]));
                if (OP == 4) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(v[i]) : "v"(n[i]));
                if (OP == 5) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(v[i]));
                if (OP == 6) asm volatile("v_cmp_eq_f64 vcc, %0, %1" : : "v"(v[i]), "v"(w[i]) : "vcc");
                if (OP == 7) asm volatile("v_max_f64 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 8) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (OP == 9) asm volatile("v_lshl_or_b32 %0, %0, 12, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if (OP == 10) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(v[i]) : "v"(f[i]));
                if (OP == 11) asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if (OP == 12) asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(n[i]), "+v"(n[(i + 1) & 7]));
                if (OP == 13) asm volatile("ds_write_b64 %0, %1" : : "v"((int)(threadIdx.x * 8 + i * 2048)), "v"(v[i]) : "memory");
                if (OP == 14) asm volatile("ds_read_b64 %0, %1" : "=v"(v[i]) : "v"((int)(threadIdx.x * 8 + i * 2048)) : "memory");
                if (OP == 15) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : "vcc");
                if (OP == 16) asm volatile("v_fract_f64 %0, %0" : "+v"(v[i]));
                if (OP == 17) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(n[i]) : "v"(v[i]));
                if (OP == 18) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(v[i]));
                if (OP == 19) asm volatile("v_add_f64 %0, %0, |%1|" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 20) asm volatile("v_min_u32 %0, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
                if (OP == 21) asm volatile("v_cmp_eq_u32 vcc, %0, %1" : : "v"(n[i]), "v"(n[(i + 1) & 7]) : "vcc");
                if (OP == 22) asm volatile("v_trunc_f64 %0, %0" : "+v"(v[i]));
                if (OP == 23) asm volatile("v_cmp_class_f64 vcc, %0, %1" : : "v"(v[i]), "v"(n[i]) : "vcc");
                if (OP == 24) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(w[i]));
                if (OP == 25) asm volatile("v_mul_f64 %0, %0, 0.5" : "+v"(v[i]));
                if (OP == 26) asm volatile("v_lshlrev_b64 %0, 1, %0" : "+v"(v[i]));
                if (OP == 27) asm volatile("v_add_co_u32 %0, vcc, %0, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]) : "vcc");
                if (OP == 28) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(v[i]) : "v"(n[i]), "v"(n[(i + 1) & 7]) : "vcc");
                if (OP == 29) asm volatile("v_mov_b64 %0, %1" : "=v"(v[i]) : "v"(w[i]));
            }
        }
        if (OP == 13 || OP == 14) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const long t1 = __builtin_amdgcn_s_memtime();
    double s = L[0] * 0.0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i] + f[i] + n[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
void run(const char *name, int wg_per_cu)
{
    const int nwg = 256 * wg_per_cu, iters = 2000;
    double *out;
    long *cyc;
    hipMalloc(&out, nwg * 256 * sizeof(double));
    hipMalloc(&cyc, nwg * sizeof(long));
    hipLaunchKernelGGL(k_bench<OP>, dim3(nwg), dim3(256), 0, 0, out, cyc, 10, 1.5);
    hipLaunchKernelGGL(k_bench<OP>, dim3(nwg), dim3(256), 0, 0, out, cyc, iters, 1.5);
    hipDeviceSynchronize();
    std::vector<long> h(nwg);
    hipMemcpy(h.data(), cyc, nwg * sizeof(long), hipMemcpyDeviceToHost);
    double mean = 0;
    for (long c : h) mean += (double)c;
    mean /= nwg;
    // wg_per_cu workgroups of 4 wavefronts per CU = wg_per_cu wavefronts per SIMD, each issuing iters*32 instructions
    printf("%-28s %d waves/SIMD: %6.2f cycles per wave-instruction per SIMD\n", name, wg_per_cu, mean / ((double)iters * 32 * wg_per_cu));
    hipFree(out); hipFree(cyc);
}

int main()
{
#define RUN(OP, NAME) run<OP>(NAME, 1); run<OP>(NAME, 2); run<OP>(NAME, 4);
    RUN(0, "v_add_f64") RUN(1, "v_mul_f64") RUN(2, "v_fma_f64") RUN(3, "v_rndne_f64") RUN(4, "v_ldexp_f64")
    RUN(5, "v_cvt_f32_f64") RUN(6, "v_cmp_eq_f64") RUN(7, "v_max_f64") RUN(8, "v_add_f32") RUN(9, "v_lshl_or_b32")
    RUN(10, "v_cvt_f64_f32") RUN(11, "v_mov_b32_dpp row_shr") RUN(12, "v_permlane32_swap_b32") RUN(13, "ds_write_b64")
    RUN(14, "ds_read_b64") RUN(15, "v_cndmask_b32") RUN(16, "v_fract_f64") RUN(17, "v_frexp_exp_i32_f64")
    RUN(18, "v_cvt_i32_f64") RUN(19, "v_add_f64 |abs|") RUN(20, "v_min_u32") RUN(21, "v_cmp_eq_u32") RUN(22, "v_trunc_f64")
    RUN(23, "v_cmp_class_f64") RUN(24, "v_pk_add_f32") RUN(25, "v_mul_f64 x0.5") RUN(26, "v_lshlrev_b64") RUN(27, "v_add_co_u32")
    RUN(28, "v_mad_u64_u32") RUN(29, "v_mov_b64")
    return 0;
}
