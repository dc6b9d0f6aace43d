// How many VALU instructions hide behind one v_mfma_f32_32x32x16_f16 on gfx950?  Streams of "1 MFMA + K fillers", K = 0..12,
// fillers of one kind (v_exp_f32 / v_max3_f32 / v_cvt_pk_f16_f32 / v_add_f32 / a softmax-like mix), 1 and 2 waves per SIMD,
// plus the split case: one wave all MFMA, its SIMD partner all VALU.   hipcc --offload-arch=gfx950 -O3 -o mfma_valu_mix mfma_valu_mix.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;

#define FILL_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define FILL_MAX(i) "v_max3_f32 %" #i ", %" #i ", %12, %13\n"
#define FILL_CVT(i) "v_cvt_pk_f16_f32 %" #i ", %" #i ", %12\n"
#define FILL_ADD(i) "v_add_f32 %" #i ", %" #i ", %12\n"

template <int KIND, int K, int ROLE>   // ROLE 0: every wave runs MFMA + K fillers; 1: waves 0-3 MFMA only, waves 4-7 fillers only (K per slot)
__global__ __launch_bounds__(512) void mix(float* out, int iters) {
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    f16v c0, c1;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    float r[12];
    for (int i = 0; i < 12; ++i) r[i] = threadIdx.x * 0.01f + i;
    float s0 = 0.5f, s1 = 0.25f;
    const bool mfma_role = ROLE == 0 || threadIdx.x < 256;
    const bool fill_role = ROLE == 0 || threadIdx.x >= 256;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (mfma_role) {
                if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            }
            if (fill_role) {
#define DO(i) if (K > i) { \
                if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])); \
                else if (KIND == 1) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s0), "v"(s1)); \
                else if (KIND == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s0)); \
                else if (KIND == 3) asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s0)); \
                else { /* softmax mix: of every 6 fillers 2 exp, 1 cvt, 1 dot2c, 1 max3, 1 add */ \
                    if ((i % 6) < 2) asm volatile("v_exp_f32 %0, %0" : "+v"(r[i])); \
                    else if ((i % 6) == 2) asm volatile("v_cvt_pk_f16_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s0)); \
                    else if ((i % 6) == 3) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(r[i]) : "v"(s0), "v"(s1)); \
                    else if ((i % 6) == 4) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(r[i]) : "v"(s0), "v"(s1)); \
                    else asm volatile("v_add_f32 %0, %0, %1" : "+v"(r[i]) : "v"(s0)); } }
                DO(0) DO(1) DO(2) DO(3) DO(4) DO(5) DO(6) DO(7) DO(8) DO(9) DO(10) DO(11)
#undef DO
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float acc = 0;
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    for (int i = 0; i < 12; ++i) acc += r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int KIND, int K, int ROLE>
static void run(const char* kind, float* d, int threads, double ghz) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((mix<KIND, K, ROLE>), dim3(256), dim3(threads), 0, 0, d, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mix<KIND, K, ROLE>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)iters * 8;            // MFMA slots per wave
    const int mfma_waves_per_simd = ROLE == 1 ? 1 : threads / 256;
    const double tf = (double)256 * 4 * mfma_waves_per_simd * slots * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("  %-8s K=%2d role=%d %d wave(s)/SIMD: %7.3f ms  %6.1f ns per MFMA slot per wave (%5.1f cycles @%.1f GHz)  MFMA rate %6.0f TF\n", kind, K, ROLE,
           threads / 256, ms, ms * 1e6 / slots, ms * 1e6 / slots * ghz, ghz, tf);
}

template <int KIND>
static void sweep(const char* kind, float* d, double ghz) {
    run<KIND, 0, 0>(kind, d, 256, ghz); run<KIND, 2, 0>(kind, d, 256, ghz); run<KIND, 4, 0>(kind, d, 256, ghz); run<KIND, 5, 0>(kind, d, 256, ghz);
    run<KIND, 6, 0>(kind, d, 256, ghz); run<KIND, 8, 0>(kind, d, 256, ghz); run<KIND, 12, 0>(kind, d, 256, ghz);
    run<KIND, 0, 0>(kind, d, 512, ghz); run<KIND, 2, 0>(kind, d, 512, ghz); run<KIND, 4, 0>(kind, d, 512, ghz); run<KIND, 5, 0>(kind, d, 512, ghz);
    run<KIND, 6, 0>(kind, d, 512, ghz); run<KIND, 8, 0>(kind, d, 512, ghz); run<KIND, 12, 0>(kind, d, 512, ghz);
    run<KIND, 4, 1>(kind, d, 512, ghz); run<KIND, 8, 1>(kind, d, 512, ghz); run<KIND, 12, 1>(kind, d, 512, ghz);
}

int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    const double ghz = 2.4;
    printf("1 MFMA (32x32x16 f16) + K fillers per slot; cycles quoted at the nominal %.1f GHz (the chip clocks lower under MFMA load)\n", ghz);
    sweep<0>("exp", d, ghz);
    sweep<1>("max3", d, ghz);
    sweep<3>("add", d, ghz);
    sweep<4>("softmax", d, ghz);
    return 0;
}
