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
