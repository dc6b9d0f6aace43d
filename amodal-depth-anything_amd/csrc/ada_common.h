// Shared device/host helpers for libada_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include "../../include/ada_hip.h"

// ---- operand type -----------------------------------------------------------------------
#ifdef ADA_OPERAND_BF16
typedef __bf16 op_t;
#define ADA_OP_DTYPE ADA_DT_BF16
#define ADA_OP_SATURATION 3.0e38f   /* bf16 conversions do not saturate: only inf / NaN register */
#else
typedef _Float16 op_t;
#define ADA_OP_DTYPE ADA_DT_F16
#define ADA_OP_SATURATION 65504.0f  /* to_op clamps to the largest finite fp16 */
#endif

typedef __attribute__((ext_vector_type(8))) op_t opx8;
typedef __attribute__((ext_vector_type(4))) op_t opx4;
typedef __attribute__((ext_vector_type(2))) op_t opx2;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;
typedef __attribute__((ext_vector_type(2))) uint32_t u32x2;

#define ADA_DEV __device__ __forceinline__

// D = A(32x16) * B(16x32) + C, fp32 accumulate.  Lane l holds A[l&31][8*(l>>5)+j], B[8*(l>>5)+j][l&31],
// D[(r&3)+8*(r>>2)+4*(l>>5)][l&31] for r in [0,16).
ADA_DEV f32x16 mfma32(opx8 a, opx8 b, f32x16 c) {
#ifdef ADA_OPERAND_BF16
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
#endif
}

// D = A(16x32) * B(32x16) + C.  Lane l holds A[l&15][8*(l>>4)+j], B[8*(l>>4)+j][l&15],
// D[4*(l>>4)+r][l&15] for r in [0,4).
ADA_DEV f32x4 mfma16(opx8 a, opx8 b, f32x4 c) {
#ifdef ADA_OPERAND_BF16
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
#else
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
#endif
}

// fp8 correction terms of a split-precision product (ada_igemm_args.f8_from): D += 2^(sa - 127) 2^(sb - 127) A(16x128, e5m2) * B(128x16, e4m3) on the
// block-scaled fp8 instruction -- 128 k per issue at twice the fp16 rate (tools/ubench/mfma_f8_corr.hip: 4555 against 1936 TFLOP/s, registers only).
// Lane l supplies 32 bytes of row / column l & 15; A and B share the (lane group, byte) -> k map, so any k order the two operands agree on
// contracts correctly.  sa / sb: E8M0 scale in byte 0, one value for the whole wave.
typedef __attribute__((ext_vector_type(8))) int i32x8;
// Inline assembly with the accumulator tied in place: the builtin (ROCm 7.2) selects an untied form of the scaled instruction -- a fresh destination
// for every issue -- and the 128 accumulators of the 256x256 tile then spill (74-82 VGPRs).  The compiler's hazard recogniser does not look into the
// statement: callers issue each accumulator once per k-step and put f8_hazard_fence() behind the last issue of a step before anything else reads them.
ADA_DEV f32x4 mfma16_f8(i32x8 a, i32x8 b, f32x4 c, int sa, int sb) {
    asm volatile("v_mfma_scale_f32_16x16x128_f8f6f4 %0, %1, %2, %0, %3, %4 op_sel_hi:[0,0,0] cbsz:1" : "+v"(c) : "v"(a), "v"(b), "v"(sa), "v"(sb));
    return c;
}
// 8-pass matrix result -> any reader: at most 11 wait states (cdna4_isa.md, MFMA hazards); 20 to be safe
ADA_DEV void f8_hazard_fence() { asm volatile("s_nop 15\n\ts_nop 3" ::: "memory"); }
// four fp32 -> four e5m2 bytes (round to nearest even; the conversion does not saturate -- 70000 becomes inf -- so clamp to the largest finite code)
ADA_DEV uint32_t bf8x4(float a, float b, float c, float d) {
    const float m = 57344.0f;
    a = __builtin_fminf(__builtin_fmaxf(a, -m), m); b = __builtin_fminf(__builtin_fmaxf(b, -m), m);
    c = __builtin_fminf(__builtin_fmaxf(c, -m), m); d = __builtin_fminf(__builtin_fmaxf(d, -m), m);
    int r = __builtin_amdgcn_cvt_pk_bf8_f32(a, b, 0, false);
    r = __builtin_amdgcn_cvt_pk_bf8_f32(c, d, r, true);
    return (uint32_t)r;
}
// The [hi | lo8 | hi8] form of a split-precision activation (split_seg < 0 in the producers' arguments): hi = round(v) operand-typed at column n,
// then per row seg BYTES lo8 = e5m2((v - hi) * 2^10) and seg bytes hi8 = e5m2(v) -- the same 2 * seg operand slots as [hi | lo].
#define ADA_F8_LO_SHIFT 1024.0f
#define ADA_F8_LO_SCALE_BYTE 117   /* E8M0 of 2^-10 */

// row of accumulator register r inside a 32x32 MFMA tile, for lane half hi = lane>>5
ADA_DEV int crow32(int r, int hi) { return (r & 3) + 8 * (r >> 2) + 4 * hi; }

// fp32 -> operand type, saturating instead of overflowing to inf (fp16 max = 65504)
ADA_DEV op_t to_op(float v) {
#ifdef ADA_OPERAND_BF16
    return (op_t)v;
#else
    v = __builtin_fminf(__builtin_fmaxf(v, -65504.0f), 65504.0f);
    return (op_t)v;
#endif
}

// unsigned division by a runtime constant through its float reciprocal; exact for n < 2^24
struct FastDiv {
    uint32_t d;
    float inv;
};
static inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d ? d : 1;
    f.inv = 1.0f / (float)f.d;
    return f;
}
ADA_DEV void fast_divmod(uint32_t n, FastDiv f, uint32_t& q, uint32_t& r) {
    uint32_t qq = (uint32_t)((float)n * f.inv);
    int32_t rr = (int32_t)(n - qq * f.d);
    if (rr < 0) {
        qq -= 1;
        rr += (int32_t)f.d;
    } else if (rr >= (int32_t)f.d) {
        qq += 1;
        rr -= (int32_t)f.d;
    }
    q = qq;
    r = (uint32_t)rr;
}

// ---- host-side error plumbing -------------------------------------------------------------
void ada_set_error(const char* fmt, ...);
int ada_check_launch(const char* what);

#define ADA_REQUIRE(cond, code, ...)   \
    do {                               \
        if (!(cond)) {                 \
            ada_set_error(__VA_ARGS__); \
            return (code);             \
        }                              \
    } while (0)
