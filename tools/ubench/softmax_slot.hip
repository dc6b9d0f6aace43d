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
// Round 6: exp2 of TWO scores per issue on the packed-fp16 pipe instead of two quarter-rate v_exp_f32 -- P is rounded to fp16 for the PV product anyway.
//   t = packed fp16 of the two (score - max) <= 0;  k = (t + 1536) - 1536 = round(t) (ulp 1 at 1536);  f = t - k in [-0.5, 0.5];
//   2^f by a degree-3 polynomial (3 v_pk_fma_f16, ~1e-4 relative);  2^k by adding k << 10 to the exponent field (v_pk_lshlrev_b16 of the magic sum's bits + v_pk_add_u16).
// 8 packed operations for two values (9 with the clamp that keeps the result normal) against 2 x v_exp_f32 + v_cvt_pk_f16_f32.
#define PKEXP2(pp, e0, e1, tmp, kf, magic, q3, q2, q1, one)                                                              \
    asm volatile("v_cvt_pk_f16_f32 %0, %2, %3\n\t"                                                                     \
                 "v_pk_add_f16 %1, %0, %4\n\t"          /* t + 1536 */                                                 \
                 "v_pk_lshlrev_b16 %2, 10, %1\n\t"      /* k << 10 (mod 2^16) from the mantissa of the magic sum */     \
                 "v_pk_add_f16 %1, %1, %4 neg_lo:[0,1] neg_hi:[0,1]\n\t"   /* k as fp16 */                              \
                 "v_pk_add_f16 %0, %0, %1 neg_lo:[0,1] neg_hi:[0,1]\n\t"   /* f = t - k */                              \
                 "v_pk_fma_f16 %1, %5, %0, %6\n\t"                                                                     \
                 "v_pk_fma_f16 %1, %1, %0, %7\n\t"                                                                     \
                 "v_pk_fma_f16 %1, %1, %0, %8\n\t"                                                                     \
                 "v_pk_add_u16 %0, %1, %2"                                                                               \
                 : "=&v"(pp), "=&v"(tmp), "+v"(e0)                                                                       \
                 : "v"(e1), "v"(magic), "v"(q3), "v"(q2), "v"(q1), "v"(one))

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
    unsigned magic = 0x66006600u, q3 = 0x2b1b2b1bu, q2 = 0x33af33afu, q1 = 0x398c398cu, one = 0x3c003c00u, kf = 0;   // 1536, 0.0555, 0.2402, 0.6931, 1.0 in packed fp16
    (void)kf;
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
            if (SEQ == 10) { unsigned tmp; PKEXP2(pp, e0, e1, tmp, kf, magic, q3, q2, q1, one); }
            if (SEQ == 11) { unsigned tmp; PKEXP2(pp, e0, e1, tmp, kf, magic, q3, q2, q1, one); ADD(acc0, e0); ADD(acc1, e1); MAX3(mx, e0, e1); }
            if (SEQ == 12) { EXP(e0); unsigned tmp; PKEXP2(pp, e0, e1, tmp, kf, magic, q3, q2, q1, one); }   // half of the scores on each path
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
    BOTH(10, "packed-fp16 polynomial exp2 of both scores (9 VALU ops)")
    BOTH(11, "packed-fp16 polynomial exp2 + add add max3")
    BOTH(12, "one v_exp_f32 + the packed polynomial (mixed)")
    return 0;
}
