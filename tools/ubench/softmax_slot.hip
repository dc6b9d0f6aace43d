// Cost of one "softmax slot" beside an MFMA on gfx950: which instruction sequence turns two scores into two packed P values
// (+ row sum + row max) at the lowest cost in MFMA-pipe time?  One slot = 1 x v_mfma_f32_32x32x16_f16 (or 2 x 16x16x32) + the sequence.
//   hipcc --offload-arch=gfx950 -O3 -o softmax_slot softmax_slot.hip
#include <hip/hip_runtime.h>
#include <stdio.h>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(16))) float f16v;
typedef __attribute__((ext_vector_type(4))) float f4v;

#define EXP(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x))
#define CVT(d, a, b) asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define DOT2C(acc, p, ones) asm volatile("v_dot2c_f32_f16 %0, %1, %2" : "+v"(acc) : "v"(p), "v"(ones))
#define ADD(acc, x) asm volatile("v_add_f32 %0, %0, %1" : "+v"(acc) : "v"(x))
#define MAX3(m, a, b) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(m) : "v"(a), "v"(b))
#define PKADD(acc, p) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(acc) : "v"(p))
#define LDSR(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr))

template <int SEQ, int MF>   // MF 0: 32x32x16, 1: two 16x16x32 per slot
__global__ __launch_bounds__(512) void slot(float* out, int iters) {
    __shared__ float lds[4096];
    lds[threadIdx.x] = threadIdx.x;
    __syncthreads();
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    f16v c0, c1;
    f4v d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    for (int i = 0; i < 16; ++i) { c0[i] = 0.f; c1[i] = 0.f; }
    float x0 = threadIdx.x * 0.01f, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, acc0 = 0, acc1 = 0, mx = 0;
    unsigned p0 = 0, p1 = 0, ones = 0x3c003c00u, pacc = 0;
    f4v ld = d0;
    const unsigned laddr = (unsigned)(size_t)lds + (threadIdx.x & 63) * 16;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (MF == 0) {
                if (u & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0);
            } else {
                if (u & 1) { d2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d2, 0, 0, 0); }
                else { d0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d0, 0, 0, 0); }
            }
            float& e0 = (u & 1) ? x2 : x0;
            float& e1 = (u & 1) ? x3 : x1;
            unsigned& pp = (u & 1) ? p1 : p0;
            if (SEQ == 1) { EXP(e0); EXP(e1); CVT(pp, e0, e1); DOT2C(acc0, pp, ones); MAX3(mx, e0, e1); }
            if (SEQ == 2) { EXP(e0); EXP(e1); CVT(pp, e0, e1); ADD(acc0, e0); ADD(acc1, e1); MAX3(mx, e0, e1); }
            if (SEQ == 3) { EXP(e0); EXP(e1); CVT(pp, e0, e1); MAX3(mx, e0, e1); }
            if (SEQ == 4) { EXP(e0); EXP(e1); CVT(pp, e0, e1); PKADD(pacc, pp); MAX3(mx, e0, e1); }
            if (SEQ == 5) { EXP(e0); EXP(e1); CVT(pp, e0, e1); }
            if (SEQ == 6) { EXP(e0); EXP(e1); }
            if (SEQ == 7) { CVT(pp, e0, e1); ADD(acc0, e0); ADD(acc1, e1); MAX3(mx, e0, e1); }
            if (SEQ == 8) { EXP(e0); EXP(e1); CVT(pp, e0, e1); ADD(acc0, e0); ADD(acc1, e1); MAX3(mx, e0, e1); LDSR(ld, laddr); }
            if (SEQ == 9) { EXP(e0); EXP(e1); CVT(pp, e0, e1); ADD(acc0, e0); ADD(acc1, e1); MAX3(mx, e0, e1); LDSR(ld, laddr); LDSR(ld, laddr); }
            if (MF == 1) {   // the second 16x16x32 of the slot sits behind the first half of the fillers' issue
                if (u & 1) d3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d3, 0, 0, 0);
                else d1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, d1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ld));
    }
    float acc = acc0 + acc1 + mx + x0 + x1 + x2 + x3 + (float)p0 + (float)p1 + (float)pacc + ld[0];
    for (int i = 0; i < 16; ++i) acc += c0[i] + c1[i];
    for (int i = 0; i < 4; ++i) acc += d0[i] + d1[i] + d2[i] + d3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

template <int SEQ, int MF>
static void run(const char* what, float* d, int threads) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((slot<SEQ, MF>), dim3(256), dim3(threads), 0, 0, d, 8);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((slot<SEQ, MF>), dim3(256), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double slots = (double)iters * 8;
    const double tf = (double)256 * 4 * (threads / 256) * slots * 2.0 * 32 * 32 * 16 / (ms * 1e-3) / 1e12;
    printf("  %-58s %s %d wave(s)/SIMD: %6.1f ns per slot per wave   MFMA rate %6.0f TF\n", what, MF ? "2x16x16x32" : "32x32x16  ", threads / 256, ms * 1e6 / slots, tf);
}

#define BOTH(SEQ, what) run<SEQ, 0>(what, d, 256); run<SEQ, 0>(what, d, 512); run<SEQ, 1>(what, d, 256); run<SEQ, 1>(what, d, 512);
int main() {
    float* d;
    hipMalloc(&d, 256 * 512 * 4);
    BOTH(0, "MFMA only")
    BOTH(6, "exp exp")
    BOTH(5, "exp exp cvt_pk")
    BOTH(3, "exp exp cvt_pk max3")
    BOTH(1, "exp exp cvt_pk dot2c max3              (round-1 softmax)")
    BOTH(2, "exp exp cvt_pk add add max3")
    BOTH(4, "exp exp cvt_pk pk_add_f16 max3")
    BOTH(7, "cvt_pk add add max3                    (no exp)")
    BOTH(8, "exp exp cvt_pk add add max3 + 1 ds_read_b128")
    BOTH(9, "exp exp cvt_pk add add max3 + 2 ds_read_b128")
    return 0;
}
