// VALU issue-rate micro-benchmark for gfx950: cycles per wave64 instruction for the operations a softmax is built from,
// measured with 1 and 2 waves per SIMD (4 / 8 waves per workgroup, one workgroup per CU).
//   hipcc --offload-arch=gfx950 -O3 -o valu_rates valu_rates.hip && ./valu_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <string.h>

#define REP8(x) x x x x x x x x
#define BODY64(INS) REP8(REP8(INS))

// each instruction instance works on its own destination register (8 rotating registers): independent streams
#define KERNEL(name, asm8)                                                                                         \
    __global__ __launch_bounds__(512) void name(float* out, int iters) {                                          \
        float r0 = threadIdx.x, r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4, r5 = r0 + 5, r6 = r0 + 6, r7 = r0 + 7; \
        float s0 = 0.5f, s1 = 0.25f;                                                                               \
        for (int i = 0; i < iters; ++i) {                                                                          \
            asm volatile(REP8(asm8) : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(r7) : "v"(s0), "v"(s1)); \
        }                                                                                                          \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;                       \
    }

KERNEL(k_fma, "v_fma_f32 %0, %8, %9, %0\n v_fma_f32 %1, %8, %9, %1\n v_fma_f32 %2, %8, %9, %2\n v_fma_f32 %3, %8, %9, %3\n v_fma_f32 %4, %8, %9, %4\n v_fma_f32 %5, %8, %9, %5\n v_fma_f32 %6, %8, %9, %6\n v_fma_f32 %7, %8, %9, %7\n")
KERNEL(k_exp, "v_exp_f32 %0, %0\n v_exp_f32 %1, %1\n v_exp_f32 %2, %2\n v_exp_f32 %3, %3\n v_exp_f32 %4, %4\n v_exp_f32 %5, %5\n v_exp_f32 %6, %6\n v_exp_f32 %7, %7\n")
KERNEL(k_exp16, "v_exp_f16 %0, %0\n v_exp_f16 %1, %1\n v_exp_f16 %2, %2\n v_exp_f16 %3, %3\n v_exp_f16 %4, %4\n v_exp_f16 %5, %5\n v_exp_f16 %6, %6\n v_exp_f16 %7, %7\n")
KERNEL(k_max3, "v_max3_f32 %0, %0, %8, %9\n v_max3_f32 %1, %1, %8, %9\n v_max3_f32 %2, %2, %8, %9\n v_max3_f32 %3, %3, %8, %9\n v_max3_f32 %4, %4, %8, %9\n v_max3_f32 %5, %5, %8, %9\n v_max3_f32 %6, %6, %8, %9\n v_max3_f32 %7, %7, %8, %9\n")
KERNEL(k_cvtpk, "v_cvt_pk_f16_f32 %0, %0, %8\n v_cvt_pk_f16_f32 %1, %1, %8\n v_cvt_pk_f16_f32 %2, %2, %8\n v_cvt_pk_f16_f32 %3, %3, %8\n v_cvt_pk_f16_f32 %4, %4, %8\n v_cvt_pk_f16_f32 %5, %5, %8\n v_cvt_pk_f16_f32 %6, %6, %8\n v_cvt_pk_f16_f32 %7, %7, %8\n")
KERNEL(k_dot2c, "v_dot2c_f32_f16 %0, %8, %9\n v_dot2c_f32_f16 %1, %8, %9\n v_dot2c_f32_f16 %2, %8, %9\n v_dot2c_f32_f16 %3, %8, %9\n v_dot2c_f32_f16 %4, %8, %9\n v_dot2c_f32_f16 %5, %8, %9\n v_dot2c_f32_f16 %6, %8, %9\n v_dot2c_f32_f16 %7, %8, %9\n")
KERNEL(k_add, "v_add_f32 %0, %0, %8\n v_add_f32 %1, %1, %8\n v_add_f32 %2, %2, %8\n v_add_f32 %3, %3, %8\n v_add_f32 %4, %4, %8\n v_add_f32 %5, %5, %8\n v_add_f32 %6, %6, %8\n v_add_f32 %7, %7, %8\n")
KERNEL(k_pkfma16, "v_pk_fma_f16 %0, %8, %9, %0\n v_pk_fma_f16 %1, %8, %9, %1\n v_pk_fma_f16 %2, %8, %9, %2\n v_pk_fma_f16 %3, %8, %9, %3\n v_pk_fma_f16 %4, %8, %9, %4\n v_pk_fma_f16 %5, %8, %9, %5\n v_pk_fma_f16 %6, %8, %9, %6\n v_pk_fma_f16 %7, %8, %9, %7\n")
KERNEL(k_pkmax16, "v_pk_max_f16 %0, %0, %8\n v_pk_max_f16 %1, %1, %8\n v_pk_max_f16 %2, %2, %8\n v_pk_max_f16 %3, %3, %8\n v_pk_max_f16 %4, %4, %8\n v_pk_max_f16 %5, %5, %8\n v_pk_max_f16 %6, %6, %8\n v_pk_max_f16 %7, %7, %8\n")
KERNEL(k_pkadd16, "v_pk_add_f16 %0, %0, %8\n v_pk_add_f16 %1, %1, %8\n v_pk_add_f16 %2, %2, %8\n v_pk_add_f16 %3, %3, %8\n v_pk_add_f16 %4, %4, %8\n v_pk_add_f16 %5, %5, %8\n v_pk_add_f16 %6, %6, %8\n v_pk_add_f16 %7, %7, %8\n")
KERNEL(k_ldexp, "v_ldexp_f32 %0, %0, %8\n v_ldexp_f32 %1, %1, %8\n v_ldexp_f32 %2, %2, %8\n v_ldexp_f32 %3, %3, %8\n v_ldexp_f32 %4, %4, %8\n v_ldexp_f32 %5, %5, %8\n v_ldexp_f32 %6, %6, %8\n v_ldexp_f32 %7, %7, %8\n")
KERNEL(k_mov, "v_mov_b32 %0, %8\n v_mov_b32 %1, %8\n v_mov_b32 %2, %8\n v_mov_b32 %3, %8\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8\n")

typedef void (*kern_t)(float*, int);

static double run(kern_t k, int threads, int iters, float* d) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, 16);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    const double ghz_nominal = prop.clockRate * 1e-6;
    struct { const char* name; kern_t k; int regs; } ks[] = {
        {"v_fma_f32", k_fma}, {"v_add_f32", k_add}, {"v_mov_b32", k_mov}, {"v_max3_f32", k_max3}, {"v_exp_f32", k_exp}, {"v_exp_f16", k_exp16},
        {"v_cvt_pk_f16_f32", k_cvtpk}, {"v_dot2c_f32_f16", k_dot2c}, {"v_ldexp_f32", k_ldexp}, {"v_pk_fma_f16", k_pkfma16}, {"v_pk_max_f16", k_pkmax16},
        {"v_pk_add_f16", k_pkadd16}};
    const int iters = 20000;
    printf("nominal clock %.2f GHz; cycles per wave64 instruction per SIMD assuming that clock (64 instr per iteration, %d iterations)\n", ghz_nominal, iters);
    for (auto& e : ks) {
        for (int threads : {256, 512}) {
            const double ms = run(e.k, threads, iters, d);
            const double instr_per_simd = (double)iters * 64 * (threads / 256);
            printf("  %-20s %d wave(s)/SIMD: %8.3f ms  -> %6.2f cycles/instr/SIMD\n", e.name, threads / 256, ms, ms * 1e-3 * ghz_nominal * 1e9 / instr_per_simd);
        }
    }
    return 0;
}
