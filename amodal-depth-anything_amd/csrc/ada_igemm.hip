// Implicit-GEMM contraction with fused epilogues (see include/ada_hip.h: ada_igemm).
//
// Tiling (gfx950).  A workgroup computes a BM x BN output tile in k-steps of BK; each wave owns TI x TJ blocks of 32x32, each
// block four 16x16 sub-tiles of v_mfma_f32_16x16x32 (fp32 accumulate) -- the MFMA flavour that sustains the higher rate under the
// chip's power cap.  Tile shapes:
//     256 x 256 x 64, 8 waves (2 x 4), wave tile 128 x 64  -- the default: 1 workgroup / CU, 128 FLOP per LDS byte staged
//     128 x 128 x 64, 4 waves, two workgroups per CU        -- N <= 128 (output_conv1, ViT-B head) and mid-size problems
//     256 x 128 x 64, 8 waves                              -- mid-size problems (picked by the quantised time estimate)
//     128 x 64 x 64 / 256 x 32 x 64, 4 waves                 -- small problems (single images, ViT-S/B at small batch: chosen by a
//                                                            quantised time estimate) and narrow outputs (32-channel tail conv)
//     (512 x 128 x 64 and a 128 x 256 x 32 tile with co-resident workgroups were A/B-ed in rounds 1-3 -- 10-15 % slower on every
//      ViT-L shape, profiles/r03_a_tile_ab.txt -- and are no longer built.)
// A and W k-slabs (rows of BK operands = 128 or 64 B) go HBM/L2 -> LDS with 16-byte buffer loads to LDS (buffer_load_dwordx4 ... lds,
// no VGPR round trip), two LDS stages, one barrier per k-step; the loads of slab t+1 are in flight during the MFMAs of slab t.
// LDS rows are stored linearly (the LDS-DMA writes wave base + lane*16) but each lane *fetches* chunk
// c ^ key(row) of its row (key = (row>>1)&7 for 128-byte rows, (row>>2)&3 for 64-byte rows) and the fragment reads
// apply the same XOR, so every ds_read_b128 lane group hits 16 distinct 16-byte bank slots (0 conflicts measured;
// cdna_hip_programming.md T2 / rule 21).
// For a 3x3 convolution the A slab of k-step (tap, kc) is the same 128-byte row segment shifted by
// (dy*Wp + dx) pixels in the zero-bordered NHWC input: the im2col gather costs one scalar add per k-step.
//
// Epilogue.  Accumulators are transposed through LDS (wave-private 32-row slabs, fp32) so that every lane then
// owns 4 consecutive columns of a row: bias / LayerScale / residual / activation run on float4, and the
// fp32 (16 B) and operand-typed (8 B) stores are contiguous 128-256 B row segments.
#include <math.h>
#include <stdarg.h>
#include <stdlib.h>
#include <atomic>
#include <mutex>
#include <type_traits>
#include "ada_common.h"

namespace {


enum { EPI_STD = 0, EPI_GELU = 1, EPI_SHUFFLE = 2, EPI_SWIGLU = 3, EPI_TAIL = 4 };

// Non-temporal hints on the streaming traffic of the epilogues (round 3 A/B) -- bit 0: the fp32 residual loads, bit 1: the fp32 stores (both
// neutral end to end: off), bit 2: the 16-byte operand-typed stores (+1.0 % end to end: ON).
#ifndef ADA_EPI_NT
#define ADA_EPI_NT 4
#endif
ADA_DEV float4 ld_res4(const float* ptr) {
#if ADA_EPI_NT & 1
    const f32x4 v = __builtin_nontemporal_load((const f32x4*)ptr);
    return make_float4(v[0], v[1], v[2], v[3]);
#else
    return *(const float4*)ptr;
#endif
}
ADA_DEV void st_f32x4(float* ptr, float4 v) {
#if ADA_EPI_NT & 2
    const f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, (f32x4*)ptr);
#else
    *(float4*)ptr = v;
#endif
}

struct IgemmDev {
    int M, N, K, a_mode;
    const op_t* A;
    long lda;
    int Ho, Wo, Hp, Wp, stride;
    FastDiv dWo, dHoWo;
    const op_t* W;
    const float* bias;
    const float* gamma;
    const float* res;
    long ldr;
    int res_row_mod, res_row_off;
    FastDiv dResMod;
    int flags;
    float* out_f32;
    long ldo_f32;
    int map_f32;
    op_t* out_op;
    long ldo_op;
    int map_op;
    int map_h, map_w;
    FastDiv dMapW, dMapHW;
    int shuffle_s, shuffle_c;
    int split_seg;   // > 0: the op-typed output is written as [hi | lo] in two column segments of this width (split precision)
    int a_dup_seg;   // > 0: the A operand is a [hi | lo] split tensor contracted as (hi, lo, hi) against [w_hi | w_hi | w_lo] weights
    int split_f8;    // the split output is [hi | lo8 | hi8] (bytes behind the hi segment) instead of [hi | lo]
    int f8_from, f8_mid;   // k-steps (inside a period of the k-walk: all of K, or one conv tap) from which the operands are fp8 bytes / the scale pair changes; 0 = off
    unsigned f8_scales;    // E8M0 scale bytes: A, W of [f8_from, f8_mid) in bits 0-15, A, W of [f8_mid, period) in bits 16-31
    int bias_row_mod;   // > 0: the bias vector depends on the row: row m uses bias[(m / bias_row_mod) * N + n] (one vector per group of rows)
    FastDiv dBiasMod;
    int a_wrap;      // PLAIN, > 0: the A row is a_wrap elements long and the k-walk wraps around once: K = 2 * a_wrap against [w_hi | w_lo] weights
    int tap_cols;    // CONV3, > 0: the N columns come in blocks of tap_cols ("phases" of a sub-pixel convolution) that use only some of the 9 taps
    unsigned long long tap_bits[3];   // 9-bit tap masks of up to 16 blocks, seven per word
    FastDiv dShC, dShS;
    const float* tail_w;
    float tail_b;
    int tail_act;
    int tiles_m, tiles_n;
    int cps;  // k-steps per conv tap = lda / 64
    int variant;  // main-loop variant for A/B runs (ada_debug_set_variant)
    int group_n;  // tile order: N-tiles are walked in column groups of this width (== tiles_n: plain row-major)
    unsigned long long* dbg;  // optional per-block timestamps (ada_debug_set_timestamps)
};

// exact-erf GELU (nn.GELU default, reference mlp.py:23): gelu(x) = x * Phi(x) = max(x, 0) - |x| * Phi(-|x|), with the lower tail
// Phi(-a) = exp2(-q(a)), q = -log2(Phi(-a)) a smooth, nearly quadratic function fitted by a degree-6 polynomial on a in [0, 6]
// (weighted for the error of the product; tools/fit_gelu.py).  fp32 evaluation: |error| <= 2.5e-7 over all x -- the class of the
// Abramowitz-Stegun 7.1.26 form used before -- in 8 full-rate VALU operations + one exp2 instead of 15 + rcp + exp.
// a is clamped to 12 (Phi(-12) ~ 2^-98 flushes the product to 0): the fitted q turns over far outside its interval.
ADA_DEV float gelu_erf(float x) {
    const float a = __builtin_fminf(__builtin_fabsf(x), 12.0f);
    float q = -3.2904290173e-05f;
    q = __builtin_fmaf(q, a, 7.6214928455e-04f);
    q = __builtin_fmaf(q, a, -8.0387993652e-03f);
    q = __builtin_fmaf(q, a, 5.3315321524e-02f);
    q = __builtin_fmaf(q, a, 4.5887145819e-01f);
    q = __builtin_fmaf(q, a, 1.1511568259e+00f);
    q = __builtin_fmaf(q, a, 9.9999958888e-01f);
    const float tail = __builtin_amdgcn_exp2f(-q);   // Phi(-|x|)
    return __builtin_fmaf(-a, tail, __builtin_fmaxf(x, 0.0f));
}

// (Round 5: the polynomial on packed fp32 -- v_pk_fma_f32 on pairs of elements, half the FMA issue slots, bit-identical -- measured NEUTRAL: fc1 + GELU
// 405.6 us against 402-412 scalar; with the GELU compiled out the launch takes 361 us, so the activation costs 44 us of VALU work that is not bound
// by FMA issue.  profiles/r05_f_ln_quarter_wave_and_packed_gelu.txt)
ADA_DEV void gelu_erf4(float4& v) {
    v.x = gelu_erf(v.x);
    v.y = gelu_erf(v.y);
    v.z = gelu_erf(v.z);
    v.w = gelu_erf(v.w);
}

// SiLU of the SwiGLU gate (reference swiglu_ffn.py:31): t * sigmoid(t), the division as one v_rcp (1 ulp) instead of the IEEE sequence
ADA_DEV float silu(float t) { return t * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896340736f * t)); }

// GEMM row -> row of an output buffer
ADA_DEV long map_row(const IgemmDev& p, int map, uint32_t m) {
    if (map == ADA_MAP_PLAIN) return (long)m;
    if (map == ADA_MAP_TOKEN) {
        uint32_t b, r;
        fast_divmod(m, p.dMapHW, b, r);
        return (long)m + b + 1;
    }
    // PAD: interior of [B, map_h+2, map_w+2]
    uint32_t b, rem, y, x;
    fast_divmod(m, p.dMapHW, b, rem);
    fast_divmod(rem, p.dMapW, y, x);
    return ((long)b * (p.map_h + 2) + (y + 1)) * (p.map_w + 2) + (x + 1);
}

// Walks GEMM rows m, m+step, m+2*step ... through the interior of a zero-bordered [B, map_h+2, map_w+2] grid without a
// division per row (valid for step <= map_w).
struct PadWalk {
    int x, y;
    long prow;
};
ADA_DEV PadWalk pad_start(const IgemmDev& p, uint32_t m) {
    uint32_t b, rem, y, x;
    fast_divmod(m, p.dMapHW, b, rem);
    fast_divmod(rem, p.dMapW, y, x);
    PadWalk w;
    w.x = (int)x; w.y = (int)y;
    w.prow = ((long)b * (p.map_h + 2) + (y + 1)) * (p.map_w + 2) + (x + 1);
    return w;
}
ADA_DEV void pad_step(const IgemmDev& p, PadWalk& w, int step) {
    w.x += step;
    w.prow += step;
    if (w.x >= p.map_w) {
        w.x -= p.map_w;
        w.y += 1;
        w.prow += 2;
        if (w.y >= p.map_h) {
            w.y = 0;
            w.prow += 2 * (p.map_w + 2);
        }
    }
}

ADA_DEV opx4 pack4(float4 v) {
    opx4 o;
    o[0] = to_op(v.x); o[1] = to_op(v.y); o[2] = to_op(v.z); o[3] = to_op(v.w);
    return o;
}

// Operand-typed stores.  With split_seg > 0 the value is written in split precision, hi = round(v) at column n, lo = round(v - hi) at
// n + seg: a following contraction over the THREE k segments (hi, lo, hi) -- the third re-reads the first, ada_igemm_args.a_dup_seg --
// against weights packed [w_hi | w_hi | w_lo] evaluates x_hi w_hi + x_lo w_hi + x_hi w_lo, i.e. the product to ~fp32 accuracy on the fp16
// matrix cores (used for selected contractions of the DPT head and the patch embedding).
// (col = dst's column inside its row: the byte segments of the [hi | lo8 | hi8] form start at row + seg, one byte per column)
template <bool F8OK = true>
ADA_DEV void store_op4(const IgemmDev& p, op_t* dst, int col, float4 v) {
    const opx4 h = pack4(v);
    *(opx4*)dst = h;
    if (p.split_seg > 0) {
        float4 r;
        r.x = v.x - (float)h[0]; r.y = v.y - (float)h[1]; r.z = v.z - (float)h[2]; r.w = v.w - (float)h[3];
        if (F8OK && p.split_f8) {
            char* b = (char*)dst + (2 * p.split_seg - col);        // = (char*)(row + seg) + col
            *(uint32_t*)b = bf8x4(r.x * ADA_F8_LO_SHIFT, r.y * ADA_F8_LO_SHIFT, r.z * ADA_F8_LO_SHIFT, r.w * ADA_F8_LO_SHIFT);
            *(uint32_t*)(b + p.split_seg) = bf8x4(v.x, v.y, v.z, v.w);
        } else {
            *(opx4*)(dst + p.split_seg) = pack4(r);
        }
    }
}
template <bool F8OK = true>
ADA_DEV void store_op8(const IgemmDev& p, op_t* dst, int col, float4 v0, float4 v1) {
    const opx4 lo = pack4(v0), hi4 = pack4(v1);
    opx8 o;
    o[0] = lo[0]; o[1] = lo[1]; o[2] = lo[2]; o[3] = lo[3];
    o[4] = hi4[0]; o[5] = hi4[1]; o[6] = hi4[2]; o[7] = hi4[3];
    // The 16-byte operand-typed stores are non-temporal: the output of a linear layer / conv is written once here and read by the NEXT
    // kernel; streamed past the caches it does not push this GEMM's weight slabs and A panels out of L2 (fc1 + GELU 427 -> 390 us per
    // launch, +1.0 % end to end: profiles/r03_o_nontemporal_stores_ab.txt).  Compile-time on purpose: behind a runtime branch the
    // optimiser merges the two stores and drops the hint (checked in the .s).  The same hint on the 8-byte stores, on the attention output,
    // on LayerNorm / bilinear outputs and on the fp32 residual traffic measured neutral or negative.
#if ADA_EPI_NT & 4
    __builtin_nontemporal_store(o, (opx8*)dst);
#else
    *(opx8*)dst = o;
#endif
    if (p.split_seg > 0) {
        float4 r0, r1;
        r0.x = v0.x - (float)lo[0]; r0.y = v0.y - (float)lo[1]; r0.z = v0.z - (float)lo[2]; r0.w = v0.w - (float)lo[3];
        r1.x = v1.x - (float)hi4[0]; r1.y = v1.y - (float)hi4[1]; r1.z = v1.z - (float)hi4[2]; r1.w = v1.w - (float)hi4[3];
        if (F8OK && p.split_f8) {
            char* b8 = (char*)dst + (2 * p.split_seg - col);       // = (char*)(row + seg) + col
            u32x2 l8, h8;
            l8[0] = bf8x4(r0.x * ADA_F8_LO_SHIFT, r0.y * ADA_F8_LO_SHIFT, r0.z * ADA_F8_LO_SHIFT, r0.w * ADA_F8_LO_SHIFT);
            l8[1] = bf8x4(r1.x * ADA_F8_LO_SHIFT, r1.y * ADA_F8_LO_SHIFT, r1.z * ADA_F8_LO_SHIFT, r1.w * ADA_F8_LO_SHIFT);
            h8[0] = bf8x4(v0.x, v0.y, v0.z, v0.w);
            h8[1] = bf8x4(v1.x, v1.y, v1.z, v1.w);
            *(u32x2*)b8 = l8;
            *(u32x2*)(b8 + p.split_seg) = h8;
            return;
        }
        const opx4 a = pack4(r0), b = pack4(r1);
        opx8 l;
        l[0] = a[0]; l[1] = a[1]; l[2] = a[2]; l[3] = a[3];
        l[4] = b[0]; l[5] = b[1]; l[6] = b[2]; l[7] = b[3];
        *(opx8*)(dst + p.split_seg) = l;
    }
}

#include "ada_igemm_pipe4.inc"

// LOOP: main loop of the kernel -- 0 single-barrier loop (every tile shape),
// 2 hand-scheduled one-wave-per-SIMD loop (256x256x64 tile with 2 x 2 waves, wave tile 128 x 128, 256 accumulators in AGPRs -- the
// library kernel's shape, profiles/r03_b_vendor_gemm_kernel_anatomy.txt: 1.5x fewer LDS fragment bytes per FLOP than the 8-wave tile).
// Its instruction stream is generated assembly (tools/gen_pipe4_asm.py -> ada_igemm_pipe4.inc); the accumulators never exist as C++ values:
// the epilogue's dump() fetches them from the AGPRs.  Measured (profiles/r03_i_gemm_4wave_asm_loop.txt): the main loop ties the 8-wave loop at
// K = 1024 and beats it -- and the library -- on long k-loops (fc2 shape with a bias epilogue: 1203 vs 1111 vs 1096 TF/s), but with one wave per
// SIMD the prologue and the VALU / latency-bound fused epilogues run slower; net it wins from K >= 8192 (the 9216-deep head convs: -7 ... -12 %),
// which is where launch_epi selects it.  (Round 2's phased 8-wave ping-pong loop measured the same as the single-barrier loop and left the build.)
// (A third loop that loaded the weight fragments straight to registers -- LDS traffic 256 -> 160 KB per k-tile -- passed every test and
// ran 30 % slower: fragment-shaped loads are expensive on the texture path.  profiles/r02_g_gemm_b_direct_ab.txt)
template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int EPI, int LOOP = 0>
__global__ __launch_bounds__(WAVES_M * WAVES_N * 64, (WAVES_M * WAVES_N == 4 && BM * BN == 256 * 256) ? 1 : 2) void igemm_kernel(IgemmDev p) {
    constexpr bool PIPE4 = LOOP == 2;
    static_assert(LOOP == 0 || LOOP == 2, "main loops: 0 single-barrier, 2 hand-scheduled 4-wave");
    static_assert(LOOP != 2 || (BM == 256 && BN == 256 && BK == 64 && WAVES_M == 2 && WAVES_N == 2), "the hand-scheduled main loop is written for the 256x256x64 tile, 2 x 2 waves");
    constexpr int NWAVES = WAVES_M * WAVES_N;
    constexpr int NT = NWAVES * 64;
    constexpr int TI = BM / (WAVES_M * 32);
    constexpr int TJ = BN / (WAVES_N * 32);
    constexpr int RB = BK * 2;                  // bytes per LDS row
    constexpr int CHUNKS = RB / 16;             // 16-byte chunks per row (8 or 4)
    constexpr int NSUB = BK / 16;               // MFMA k-sub-steps per k-step (4 or 2)
    constexpr int A_BYTES = BM * RB;
    constexpr int B_BYTES = BN * RB;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int ROWS_PER_PASS = NT / CHUNKS;  // rows copied by one workgroup-wide LDS-DMA pass
    constexpr int A_IT = BM / ROWS_PER_PASS;
    constexpr int B_IT = BN / ROWS_PER_PASS;
    static_assert(BK == 64 || BK == 32, "BK is 64 or 32");
    static_assert(A_IT >= 1 && B_IT >= 1 && BM % ROWS_PER_PASS == 0 && BN % ROWS_PER_PASS == 0, "tile vs staging pass");
    static_assert(TJ == 1 || TJ == 2 || TJ == 4, "wave tile is 32, 64 or 128 columns wide");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);

    // XCD-aware (bijective) block remap: each XCD's L2 sees a contiguous run of tiles that share A panels.
    int tm, tn;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        // column-group-major order: all M-panels of a group of `group_n` N-tiles before the next group, so the group's
        // weight slabs stay resident in the XCD's 4 MiB L2 instead of being re-streamed every round of tiles
        const int gn = p.group_n;
        const int full = p.tiles_n / gn;              // number of full-width groups
        const int per_group = p.tiles_m * gn;
        if (logical < full * per_group) {
            const int g = logical / per_group, rr = logical - g * per_group;
            tm = rr / gn;
            tn = g * gn + (rr - tm * gn);
        } else {
            const int gl = p.tiles_n - full * gn;     // width of the last, narrower group
            const int rr = logical - full * per_group;
            tm = rr / gl;
            tn = full * gn + (rr - tm * gl);
        }
    }
    const int m0 = tm * BM, n0 = tn * BN;
    unsigned long long t_entry = 0, t_first = 0, t_loop = 0, t_vm = 0, t_bar = 0;
    if (p.dbg) t_entry = __builtin_amdgcn_s_memtime();

    // ---- per-thread staging addresses --------------------------------------------------
    const int srow = tid / CHUNKS;                                  // row inside a staging pass
    const int gchunk = (tid % CHUNKS) ^ ((tid >> 4) & (CHUNKS - 1));  // swizzled source chunk; key(row) == (tid>>4)&(CHUNKS-1)
    // Every copy address is (workgroup-uniform tile base + k-step offset) + a 32-bit per-lane byte offset that never changes.
    // The copies are buffer loads to LDS (buffer_load_dwordx4 ... offen lds): tile base in a buffer resource (SGPRs), k-step
    // offset in the scalar offset, lane offset in one VGPR -- a copy is s_mov m0 + the load, no 64-bit VALU add per copy.
    auto a_row_base = [&](uint32_t m) -> long {   // element offset of GEMM row m's first operand element
        if (p.a_mode == ADA_A_PLAIN) return (long)m * p.lda;
        uint32_t b, rem, y, x;
        fast_divmod(m, p.dHoWo, b, rem);
        fast_divmod(rem, p.dWo, y, x);
        return (((long)b * p.Hp + (long)y * p.stride) * p.Wp + (long)x * p.stride) * p.lda;
    };
    const uint32_t m_first = (uint32_t)m0 < (uint32_t)p.M ? (uint32_t)m0 : (uint32_t)p.M - 1;
    const long a_tile_el = a_row_base(m_first);                          // uniform: rows of the tile only go up from here
    const op_t* a_tile = p.A + a_tile_el;
    const op_t* b_tile = p.W + (long)(n0 < p.N ? n0 : p.N - 1) * p.K;
    // raw buffer resources: stride 0, 2 GiB window above the tile base (nothing relies on out-of-range behaviour), dword 3 = 0x20000
    const __amdgpu_buffer_rsrc_t a_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)a_tile, 0, 0x7fffffff, 0x20000);
    const __amdgpu_buffer_rsrc_t b_rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)b_tile, 0, 0x7fffffff, 0x20000);
    uint32_t a_off[A_IT], b_off[B_IT];                                    // bytes
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        uint32_t m = (uint32_t)(m0 + it * ROWS_PER_PASS + srow);
        if (m >= (uint32_t)p.M) m = (uint32_t)p.M - 1;
        a_off[it] = (uint32_t)((a_row_base(m) - a_tile_el + gchunk * 8) * 2);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int n = n0 + it * ROWS_PER_PASS + srow;
        if (n >= p.N) n = p.N - 1;
        b_off[it] = (uint32_t)(((long)(n - (n0 < p.N ? n0 : p.N - 1)) * p.K + gchunk * 8) * 2);
    }

    // k-step kt -> element offsets of its A and W slabs (wave-uniform scalars)
    // a split A operand ([hi | lo] per row / per tap) is contracted as three k segments (hi, lo, hi): segment 2 re-reads segment 0
    const int sps = p.a_dup_seg / BK;                                   // k-steps per segment (0: plain operand)
    const int wrap = sps > 0 ? 2 * sps : p.a_wrap / BK;                 // PLAIN: k-step at which the A walk starts over (0: never)
    const int cps = sps > 0 ? 3 * sps : (int)(p.lda / BK);              // k-steps per conv tap
    auto slab_offsets = [&](int kt, long& aoff, long& boff) {
        if (p.a_mode == ADA_A_PLAIN) {
            aoff = (long)((wrap > 0 && kt >= wrap) ? kt - wrap : kt) * BK;
        } else {
            const int tap = kt / cps;
            int kc = kt - tap * cps;
            if (sps > 0 && kc >= 2 * sps) kc -= 2 * sps;
            const int dy = tap / 3, dx = tap - dy * 3;
            aoff = ((long)dy * p.Wp + dx) * p.lda + (long)kc * BK;
        }
        boff = (long)kt * BK;
    };
    // issue the global->LDS copies of part `part` (of NSUB) of a stage; part < 0 issues everything
    auto stage_part = [&](int buf, long aoff, long boff, int part) {
        char* sa = smem + buf * STAGE_BYTES + wave * 1024;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            if (part < 0 || (it % NSUB) == part)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(a_rsrc, (__attribute__((address_space(3))) void*)(sa + it * (NT * 16)), 16, (int)a_off[it],
                                                         (int)(aoff * 2), 0, 0);
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            if (part < 0 || ((it + NSUB / 2) % NSUB) == part)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(b_rsrc, (__attribute__((address_space(3))) void*)(sb + it * (NT * 16)), 16, (int)b_off[it],
                                                         (int)(boff * 2), 0, 0);
        }
    };

    // ---- main loop -----------------------------------------------------------------------
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    // v_mfma_f32_16x16x32: four 16x16 sub-tiles per 32x32 block (index 2a+b = row half a, column half b).  Same FLOPs per
    // LDS byte as the 32x32x16 instruction but a better rate under the power cap: +7-9 % on every GEMM shape of the model
    // (profiles/r01_j_gemm_mfma16_ab.txt; registers-only streams: 1.79 vs 1.72 PF, profiles/r01_g_mfma_power_ceiling.txt).
    f32x4 acc[TI][TJ][4];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int ab = 0; ab < 4; ++ab)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][ab][r] = 0.0f;

    // Sub-pixel convolutions (a stride-s transposed conv merged with the 3x3 conv that follows it, see ada_igemm_args.tap_cols): each block of
    // tap_cols output columns is one output phase and touches only 1, 2 or 4 of the 9 taps of the coarse grid -- the other weight blocks are
    // structurally zero.  The k-walk of this N-tile visits the union of its phases' taps only.
    unsigned long long taps_packed = 0x876543210ull;   // active taps in walk order, 4 bits each
    int ntaps = 9;
    if (p.a_mode != ADA_A_PLAIN && p.tap_cols > 0) {
        const int first = n0 / p.tap_cols, lastn = (n0 + BN < p.N ? n0 + BN : p.N) - 1;
        const int last = lastn / p.tap_cols;
        unsigned mask = 0;
        for (int ph = first; ph <= last; ++ph) {
            const int wi = ph >= 14 ? 2 : (ph >= 7 ? 1 : 0);
            mask |= (unsigned)(p.tap_bits[wi] >> (9 * (ph - 7 * wi))) & 0x1ffu;
        }
        mask = (unsigned)__builtin_amdgcn_readfirstlane((int)mask);
        taps_packed = 0;
        ntaps = 0;
        for (int t = 0; t < 9; ++t)
            if ((mask >> t) & 1u) taps_packed |= (unsigned long long)t << (4 * ntaps++);
    }
    const int nk = p.a_mode == ADA_A_PLAIN ? p.K / BK : ntaps * cps;
    if constexpr (PIPE4) {
        // everything between here and the epilogue's first dump() is the generated asm: fragments in v[0:127], accumulators in a[0:255]
        const int l15 = lane & 15, q4 = lane >> 4;
        const unsigned lds0 = (unsigned)(size_t)smem;
        const unsigned coff0 = (unsigned)((q4 ^ ((l15 >> 1) & 7)) * 16);
        const unsigned abase = lds0 + (unsigned)((wm * 128 + l15) * RB) + coff0;
        const unsigned bbase = lds0 + (unsigned)(A_BYTES + (wn * 128 + l15) * RB) + coff0;
        const unsigned m0s0 = (unsigned)__builtin_amdgcn_readfirstlane((int)(lds0 + (unsigned)wave * 1024u));
        constexpr unsigned OOB = 0x7fffffffu;     // a scalar offset beyond num_records: the copy zero-fills without fetching
        long a0o, b0o, a1o = 0, b1o = 0, a2o = 0, b2o = 0;
        slab_offsets(0, a0o, b0o);
        if (nk > 1) slab_offsets(1, a1o, b1o);
        if (nk > 2) slab_offsets(2, a2o, b2o);
        unsigned period = 0x7fffffffu, cnt = 0, jump = 0;
        if (p.a_mode != ADA_A_PLAIN) {   // 3x3 conv: consecutive k-tiles of an input row (three taps) are contiguous; every 3 * cps k-tiles the
            period = (unsigned)(3 * cps);   // A window moves down one padded input row
            cnt = 2u % period;
            jump = (unsigned)(((long)(p.Wp - 3) * p.lda) * 2);
        }
        // the asm needs the two buffer resources in SGPRs: rebuild them from readfirstlane'd pointer halves so that their uniformity is
        // provable (the tile bases come out of float-reciprocal divisions, i.e. VALU registers -- cdna_hip_programming.md T20)
        auto uniform_rsrc = [](const void* ptr) {
            const unsigned long long v = (unsigned long long)ptr;
            const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)v), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
            return __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)hi << 32) | lo), 0, 0x7fffffff, 0x20000);
        };
        const __amdgpu_buffer_rsrc_t a_rs = uniform_rsrc(a_tile), b_rs = uniform_rsrc(b_tile);
        pipe4_main_loop(a_rs, b_rs, a_off, b_off, abase, bbase, m0s0, (unsigned)(a0o * 2), nk > 1 ? (unsigned)(a1o * 2) : OOB, nk > 1 ? 128u : OOB,
                        (unsigned)(a2o * 2), (unsigned)nk, period, cnt, jump);
    } else {
    // Offsets of the next slab to stage.  Plain operands walk k-step j directly; a 3x3 conv walks (active tap, k-step inside the tap) with two
    // scalar counters (no division per k-step), the taps taken from taps_packed.  The walk state is passed and returned BY VALUE: captured by
    // reference it has its address taken, and the "memory" clobbers of the loop's waits then pin it to scratch (12 bytes, reloaded every k-step).
    struct Walk { int j, ti, kc; };
    auto next_offsets = [=](Walk w, long& aoff, long& boff) -> Walk {
        if (p.a_mode == ADA_A_PLAIN) {
            aoff = (long)((wrap > 0 && w.j >= wrap) ? w.j - wrap : w.j) * BK;
            boff = (long)w.j * BK;
            w.j += 1;
            return w;
        }
        const int tap = (int)((taps_packed >> (4 * w.ti)) & 15ull);
        const int kca = (sps > 0 && w.kc >= 2 * sps) ? w.kc - 2 * sps : w.kc;
        const int dy = (tap * 11) >> 5, dx = tap - dy * 3;
        aoff = ((long)dy * p.Wp + dx) * p.lda + (long)kca * BK;
        boff = ((long)tap * cps + w.kc) * BK;
        if (++w.kc == cps) {
            w.kc = 0;
            ++w.ti;
        }
        return w;
    };
    Walk walk{0, 0, 0};
    {
        long aoff, boff;
        walk = next_offsets(walk, aoff, boff);
        stage_part(0, aoff, boff, -1);
    }
    {
        // All copies of slab t+1 are issued right after the barrier; fragment reads are scheduled by the compiler.
        // (Hand-counted lgkmcnt pipelines and a ping-pong split of the two waves per SIMD were tried and measured: fewer
        // cycles per k-step but no wall-clock gain on this power-limited kernel -- profiles/r01_c_gemm_sched{4,5}_ab.txt.)
        // One k-step; F8: the slab's 128-byte rows hold 128 e5m2 (A) / e4m3 (W) codes (ada_igemm_args.f8_from) and go to the fp8 instruction.  The two
        // kinds of step live in two loops, not behind a branch in one: with both in one body the register allocator gives up tying the accumulators.
        auto k_step = [&](int kt, Walk w, unsigned sc, auto f8_tag) __attribute__((always_inline)) -> Walk {
            constexpr bool F8 = decltype(f8_tag)::value;
            const int cur = kt & 1;
            unsigned long long tw0 = 0, tw1 = 0;
            if (p.dbg) tw0 = __builtin_amdgcn_s_memtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (p.dbg) tw1 = __builtin_amdgcn_s_memtime();
            __syncthreads();
            if (p.dbg) {
                const unsigned long long tw2 = __builtin_amdgcn_s_memtime();
                t_vm += tw1 - tw0;
                t_bar += tw2 - tw1;
            }
            if (p.dbg && kt == 0) t_first = __builtin_amdgcn_s_memtime();
            // The two waves that share a SIMD (w and w+4 of an 8-wave workgroup) leave the barrier together; if both issued
            // their global->LDS copies first, neither would have MFMAs in flight for a few hundred cycles.  One group therefore
            // issues its copies after the first 32-wide half of the slab (they still have half a k-step to land).  Waves 0-3 are
            // the late group (waves 4-7 late: qkv +1.7 %, fc2 +3 % slower; nobody: -3 %; everybody: -5 % on long K --
            // profiles/r01_l_gemm_copy_stagger_variants.txt).
            const bool late = (NWAVES == 8) && wave < 4;
            const bool more = kt + 1 < nk;
            long aoff = 0, boff = 0;
            if (more) w = next_offsets(w, aoff, boff);
            // (Spreading the 8 copies of a wave over the MFMAs of "its" k half -- two behind every 8 MFMAs, order pinned with sched_barrier, the
            // thing that was worth 20 % in the 4-wave loop -- makes THIS loop slower: fc1 +10 %, fc2 +14 %, 8192^3 +20 %; the partner wave on the
            // SIMD already covers a burst, and the pins cost the compiler its own schedule.  profiles/r03_i_gemm_4wave_asm_loop.txt)
            if (more && !late) stage_part(cur ^ 1, aoff, boff, -1);
            const char* sbase = smem + cur * STAGE_BYTES;
            const int l15 = lane & 15, q4 = lane >> 4;
            const int a16_off = (wm * TI * 32 + l15) * RB, b16_off = A_BYTES + (wn * TJ * 32 + l15) * RB;
            if constexpr (F8) {
                // a lane's two 16-byte chunks q4 and 4 + q4 -- the addresses of the fp16 step's two k halves -- are its 32 bytes of one 16x16x128 issue
                // (A and W split k the same way, which is all the contraction needs)
                const int sa = (int)(sc & 255u), sb = (int)((sc >> 8) & 255u);
                const int key = (l15 >> 1) & 7;
                const int c0 = ((q4 ^ key) * 16), c1 = (((4 + q4) ^ key) * 16);
                auto frag8 = [&](int off) -> i32x8 {
                    const u32x4 x = *(const u32x4*)(sbase + off + c0), y = *(const u32x4*)(sbase + off + c1);
                    i32x8 r;
                    r[0] = (int)x[0]; r[1] = (int)x[1]; r[2] = (int)x[2]; r[3] = (int)x[3];
                    r[4] = (int)y[0]; r[5] = (int)y[1]; r[6] = (int)y[2]; r[7] = (int)y[3];
                    return r;
                };
                i32x8 bf8[TJ][2];
#pragma unroll
                for (int j = 0; j < TJ; ++j)
#pragma unroll
                    for (int b2 = 0; b2 < 2; ++b2) bf8[j][b2] = frag8(b16_off + (j * 32 + b2 * 16) * RB);
#pragma unroll
                for (int i = 0; i < TI; ++i) {
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const i32x8 af8 = frag8(a16_off + (i * 32 + a * 16) * RB);
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
#pragma unroll
                            for (int b2 = 0; b2 < 2; ++b2) acc[i][j][2 * a + b2] = mfma16_f8(af8, bf8[j][b2], acc[i][j][2 * a + b2], sa, sb);
                    }
                    // (the A fragments of one 32-row block at a time: hoisted together, the eight of the 256x256 tile do not fit beside its 128 accumulators)
                    __builtin_amdgcn_sched_barrier(0);
                    if (i == (TI - 1) / 2) {
                        if (more && late) stage_part(cur ^ 1, aoff, boff, -1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
                f8_hazard_fence();
            } else {
#pragma unroll
                for (int s = 0; s < BK / 32; ++s) {   // 32-wide k halves
                    const int coff = ((4 * s + q4) ^ (CHUNKS == 8 ? ((l15 >> 1) & 7) : ((l15 >> 2) & 3))) * 16;
                    opx8 af[TI][2], bf[TJ][2];
#pragma unroll
                    for (int i = 0; i < TI; ++i)
#pragma unroll
                        for (int a = 0; a < 2; ++a) af[i][a] = *(const opx8*)(sbase + a16_off + (i * 32 + a * 16) * RB + coff);
#pragma unroll
                    for (int j = 0; j < TJ; ++j)
#pragma unroll
                        for (int b2 = 0; b2 < 2; ++b2) bf[j][b2] = *(const opx8*)(sbase + b16_off + (j * 32 + b2 * 16) * RB + coff);
#pragma unroll
                    for (int i = 0; i < TI; ++i) {
#pragma unroll
                        for (int j = 0; j < TJ; ++j)
#pragma unroll
                            for (int a = 0; a < 2; ++a)
#pragma unroll
                                for (int b2 = 0; b2 < 2; ++b2) acc[i][j][2 * a + b2] = mfma16(af[i][a], bf[j][b2], acc[i][j][2 * a + b2]);
                    }
                    if (s == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (more && late) stage_part(cur ^ 1, aoff, boff, -1);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
            return w;
        };
        if (BK != 64 || p.f8_from == 0) {
            for (int kt = 0; kt < nk; ++kt) walk = k_step(kt, walk, 0u, std::false_type{});
        } else {
            // periods of the k-walk (all of K; one tap of a 3x3 conv): f8_from fp16 steps, then fp8 steps whose scale pair changes at f8_mid
            const int per = p.a_mode == ADA_A_PLAIN ? nk : cps;
            for (int kt = 0; kt < nk;) {
                for (int r = 0; r < p.f8_from; ++r, ++kt) walk = k_step(kt, walk, 0u, std::false_type{});
                for (int r = p.f8_from; r < p.f8_mid; ++r, ++kt) walk = k_step(kt, walk, p.f8_scales, std::true_type{});
                for (int r = p.f8_mid; r < per; ++r, ++kt) walk = k_step(kt, walk, p.f8_scales >> 16, std::true_type{});
            }
        }
    }
    }  // single-barrier loop

    // ---- epilogue: transpose through a wave-private LDS slab, then float4 per lane ----------------
    // The wave tile is walked in 32-row x GW-column groups (GW = 64, or 32 for the narrow tiles).
    __syncthreads();  // every wave is done reading the last stage before the slabs overwrite it
    if (p.dbg) t_loop = __builtin_amdgcn_s_memtime();
    constexpr int GW = TJ >= 2 ? 64 : 32;      // columns per epilogue group
    constexpr int GJ = GW / 32;                // MFMA tiles per group
    constexpr int NG = TJ / GJ;                // groups per wave-tile row block
    constexpr int SW = GW + 4;                 // slab row stride in floats: the four 16-lane quarters of a 16x16 dump (rows 4 apart) hit disjoint banks
    static_assert(NWAVES * 32 * SW * 4 <= 2 * STAGE_BYTES, "epilogue slabs must fit in the stage buffers");
    float* slab = (float*)(smem + wave * (32 * SW * 4));
    const int flags = p.flags;
    const int mbase = m0 + wm * TI * 32;
    const int nwave = n0 + wn * TJ * 32;

    // D[4*(lane>>4)+rr][lane&15] of sub-tile (a, b) -> slab row 16a + 4*(lane>>4) + rr, column 32jj + 16b + (lane&15): four rows per
    // lane and sub-tile, written as two ds_write2_b32 (rows rr, rr+1) off one base address per row half a (hipcc pairs only a
    // third of the stores on its own)
    const unsigned slab_lds = (unsigned)(size_t)slab + (unsigned)((4 * (lane >> 4) * SW + (lane & 15)) * 4);
    auto dump = [&](int i, int g) {
#pragma unroll
        for (int jj = 0; jj < GJ; ++jj)
#pragma unroll
            for (int ab = 0; ab < 4; ++ab) {
                f32x4 v;
                if constexpr (PIPE4) {   // accumulator registers of sub-tile (i, j, ab): a[16 (4 i + j) + 4 ab ...] (tools/gen_pipe4_asm.py areg)
                    const int ar = 16 * (4 * i + (g * GJ + jj)) + 4 * ab;
                    float v0, v1, v2, v3;
                    asm volatile("v_accvgpr_read_b32 %0, a[%4]\n\tv_accvgpr_read_b32 %1, a[%5]\n\tv_accvgpr_read_b32 %2, a[%6]\n\tv_accvgpr_read_b32 %3, a[%7]"
                                 : "=v"(v0), "=v"(v1), "=v"(v2), "=v"(v3) : "i"(ar), "i"(ar + 1), "i"(ar + 2), "i"(ar + 3));
                    v[0] = v0; v[1] = v1; v[2] = v2; v[3] = v3;
                } else {
                    v = acc[i][g * GJ + jj][ab];
                }
                const unsigned base = slab_lds + (unsigned)((ab >> 1) * 16 * SW * 4);
                const int col = jj * 32 + 16 * (ab & 1);   // in floats; the two offsets of ds_write2_b32 count 4-byte units (< 256)
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(base), "v"(v[0]), "v"(v[1]), "i"(col), "i"(col + SW) : "memory");
                asm volatile("ds_write2_b32 %0, %1, %2 offset0:%3 offset1:%4" ::"v"(base), "v"(v[2]), "v"(v[3]), "i"(col + 2 * SW), "i"(col + 3 * SW) : "memory");
            }
    };
    static_assert((GJ - 1) * 32 + 16 + 3 * SW < 256, "ds_write2_b32 offsets are 8 bits");

    if constexpr (EPI == EPI_SWIGLU) {
        // packer interleaved the w12 rows in 32-wide groups: columns [0,32) of a 64-column group = x1, [32,64) = x2
        static_assert(GW == 64 || EPI != EPI_SWIGLU, "SwiGLU needs 64-column groups");
        const int c8 = lane & 7, rsub = lane >> 3;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int nbase = nwave + g * GW;
            const int n1 = nbase + 4 * c8;
            const bool nval = n1 + 32 < p.N;
            float4 b1 = make_float4(0, 0, 0, 0), b2 = b1;
            if (nval && (flags & ADA_EP_BIAS)) {
                b1 = *(const float4*)(p.bias + n1);
                b2 = *(const float4*)(p.bias + n1 + 32);
            }
            const int nh = (nbase >> 1) + 4 * c8;  // hidden column
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                dump(i, g);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int row = k * 8 + rsub;
                    const int m = mbase + i * 32 + row;
                    const float4 x1 = *(const float4*)(slab + row * SW + 4 * c8);
                    const float4 x2 = *(const float4*)(slab + row * SW + 32 + 4 * c8);
                    if (m < p.M && nval) {
                        float4 gt;
                        gt.x = silu(x1.x + b1.x) * (x2.x + b2.x);
                        gt.y = silu(x1.y + b1.y) * (x2.y + b2.y);
                        gt.z = silu(x1.z + b1.z) * (x2.z + b2.z);
                        gt.w = silu(x1.w + b1.w) * (x2.w + b2.w);
                        // (split forms, round 6: the gated hidden feeds mlp.w3 -- swiglu_ffn.py:33 -- and exists in the operand type only; in a split-precision block it
                        //  is written [hi | lo] / [hi | lo8 | hi8] like every other split activation.  Never in the generated 4-wave loop's kernel: use_pipe4)
                        if constexpr (PIPE4) *(opx4*)(p.out_op + (long)m * p.ldo_op + nh) = pack4(gt);
                        else store_op4(p, p.out_op + (long)m * p.ldo_op + nh, nh, gt);
                    }
                }
            }
        }
    } else if constexpr (EPI == EPI_TAIL) {
        constexpr int CG = GW / 4;
        constexpr int RPI = 64 / CG;
        static_assert(NG == 1 || EPI != EPI_TAIL, "tail epilogue: one column group per wave");
        const int cg = lane % CG, rsub = lane / CG;
        const int n = nwave + 4 * cg;
        const bool nval = n < p.N;
        float4 bias4 = make_float4(0, 0, 0, 0), tail4 = make_float4(0, 0, 0, 0);
        if (nval) {
            if (flags & ADA_EP_BIAS) bias4 = *(const float4*)(p.bias + n);
            tail4 = *(const float4*)(p.tail_w + n);
        }
#pragma unroll
        for (int i = 0; i < TI; ++i) {
            dump(i, 0);
#pragma unroll
            for (int k = 0; k < 32 / RPI; ++k) {
                const int row = k * RPI + rsub;
                const int m = mbase + i * 32 + row;
                const float4 v = *(const float4*)(slab + row * SW + 4 * cg);
                float part = __builtin_fmaxf(v.x + bias4.x, 0.f) * tail4.x + __builtin_fmaxf(v.y + bias4.y, 0.f) * tail4.y +
                             __builtin_fmaxf(v.z + bias4.z, 0.f) * tail4.z + __builtin_fmaxf(v.w + bias4.w, 0.f) * tail4.w;
#pragma unroll
                for (int o = 1; o < CG; o <<= 1) part += __shfl_xor(part, o);
                if (cg == 0 && m < p.M) {
                    float d = part + p.tail_b;
                    if (p.tail_act == ADA_ACT_SIGMOID) d = 1.0f / (1.0f + __expf(-d));
                    else if (p.tail_act == ADA_ACT_RELU) d = __builtin_fmaxf(d, 0.0f);
                    p.out_f32[m] = d;
                }
            }
        }
    } else if (EPI != EPI_SHUFFLE && m0 + BM <= p.M && n0 + BN <= p.N && p.res_row_mod == 0 && p.bias_row_mod == 0 &&
               (!p.out_op || p.map_op == ADA_MAP_PLAIN || (p.map_op == ADA_MAP_PAD && p.map_w >= 8)) &&
               (!p.out_f32 || p.map_f32 == ADA_MAP_PLAIN) &&
               (p.out_f32 || (flags & ADA_EP_RESIDUAL) || (p.ldo_op & 7) == 0)) {
        // ---- interior tiles with plain (or zero-bordered NHWC) row maps -- every linear layer of the encoder, the 3x3 convs of
        //      the head: no bounds checks, no per-row division, row pointers advance by constant strides / a PadWalk.  The general code below spends most of its issue slots on exactly that
        //      bookkeeping: 10.7 k cycles per 256x256 tile for fp16 output, 35 k for the fp32 residual update, against 4-8 k
        //      for this path (profiles/r01_g_gemm_tile_anatomy.txt) -- the epilogue is VALU-issue-bound, not memory-bound.
        const bool has_bias = (flags & ADA_EP_BIAS) != 0, has_gamma = (flags & ADA_EP_GAMMA) != 0;
        if (!p.out_f32 && !(flags & ADA_EP_RESIDUAL)) {
            constexpr int CG = GW / 8, RPI = 64 / CG;
            // Lane -> (slab row rsub, column group cg).  The slab reads are ds_read_b128, serviced in the lane groups {0-3, 12-15, 20-27},
            // {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): with cg = lane % CG two lanes of every group met on one 16-byte bank slot
            // (rows are SW = GW + 4 floats apart), i.e. every read took twice its LDS cycles -- 4-6 % of ALL LDS cycles of the kernel
            // (SQ_LDS_BANK_CONFLICT).  Rotating the column group by the row makes the 16 lanes of a group hit 16 distinct slots.
            const int rsub = lane / CG, cg = CG == 8 ? ((lane - (rsub >> 1)) & 7) : lane % CG;
            const bool relu = (flags & ADA_EP_RELU_OP) != 0;
            const long ld = p.ldo_op;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const int n = nwave + g * GW + 8 * cg;
                float4 b0 = make_float4(0, 0, 0, 0), b1 = b0, g0 = make_float4(1, 1, 1, 1), g1 = g0;
                if (has_bias) { b0 = *(const float4*)(p.bias + n); b1 = *(const float4*)(p.bias + n + 4); }
                if (has_gamma) { g0 = *(const float4*)(p.gamma + n); g1 = *(const float4*)(p.gamma + n + 4); }
                op_t* dst = p.out_op + (long)(mbase + rsub) * ld + n;
                const bool pad = p.map_op == ADA_MAP_PAD;
#pragma unroll
                for (int i = 0; i < TI; ++i) {
                    dump(i, g);
                    PadWalk walk;
                    if (pad) walk = pad_start(p, (uint32_t)(mbase + i * 32 + rsub));
#pragma unroll
                    for (int k = 0; k < 32 / RPI; ++k) {
                        const int row = k * RPI + rsub;
                        float4 v0 = *(const float4*)(slab + row * SW + 8 * cg);
                        float4 v1 = *(const float4*)(slab + row * SW + 8 * cg + 4);
                        v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w;
                        v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
                        if constexpr (EPI == EPI_GELU) {
                            gelu_erf4(v0);
                            gelu_erf4(v1);
                        }
                        if (has_gamma) {
                            v0.x *= g0.x; v0.y *= g0.y; v0.z *= g0.z; v0.w *= g0.w;
                            v1.x *= g1.x; v1.y *= g1.y; v1.z *= g1.z; v1.w *= g1.w;
                        }
                        if (relu) {
                            v0.x = __builtin_fmaxf(v0.x, 0.f); v0.y = __builtin_fmaxf(v0.y, 0.f); v0.z = __builtin_fmaxf(v0.z, 0.f); v0.w = __builtin_fmaxf(v0.w, 0.f);
                            v1.x = __builtin_fmaxf(v1.x, 0.f); v1.y = __builtin_fmaxf(v1.y, 0.f); v1.z = __builtin_fmaxf(v1.z, 0.f); v1.w = __builtin_fmaxf(v1.w, 0.f);
                        }
                        if (pad) {
                            store_op8(p, p.out_op + walk.prow * ld + n, n, v0, v1);
                            pad_step(p, walk, RPI);
                        } else {
                            store_op8(p, dst + (long)(i * 32 + k * RPI) * ld, n, v0, v1);
                        }
                    }
                }
            }
        } else {
            constexpr int CG = GW / 4, RPI = 64 / CG, NKI = 32 / RPI, NPASS = NG * TI;
            const int rsub = lane / CG, cg = CG == 16 ? ((lane - rsub) & 15) : lane % CG;   // column group rotated by the row: see above
            const bool has_res = (flags & ADA_EP_RESIDUAL) != 0;
            const bool relu_f = (flags & ADA_EP_RELU_F32) != 0, relu_o = (flags & ADA_EP_RELU_OP) != 0;
            const long ldr = p.ldr, ldf = p.ldo_f32, ldo = p.ldo_op;
            auto rptr = [&](int q) -> const float* {
                const int g = q / TI, i = q - g * TI;
                return p.res + (long)(mbase + i * 32 + rsub) * ldr + (nwave + g * GW + 4 * cg);
            };
            // residual rows of the next 32-row pass are requested before this pass is transposed (8-wave tiles: the registers
            // are there; with co-resident workgroups the neighbours' MFMAs cover the latency instead)
            constexpr bool AHEAD = (NWAVES == 8 && TJ <= 2) || PIPE4;
            float4 rcur[NKI], rnext[AHEAD ? NKI : 1];
#pragma unroll
            for (int k = 0; k < NKI; ++k) rcur[k] = make_float4(0, 0, 0, 0);
            if (AHEAD && has_res) {
                const float* r0 = rptr(0);
#pragma unroll
                for (int k = 0; k < NKI; ++k) rcur[k] = ld_res4(r0 + (long)(k * RPI) * ldr);
            }
#pragma unroll
            for (int q = 0; q < NPASS; ++q) {
                const int g = q / TI, i = q - g * TI;
                const int n = nwave + g * GW + 4 * cg;
                float4 bias4 = make_float4(0, 0, 0, 0), gamma4 = make_float4(1, 1, 1, 1);
                if (has_bias) bias4 = *(const float4*)(p.bias + n);
                if (has_gamma) gamma4 = *(const float4*)(p.gamma + n);
                if (has_res) {
                    if constexpr (AHEAD) {
                        if (q + 1 < NPASS) {
                            const float* r1 = rptr(q + 1);
#pragma unroll
                            for (int k = 0; k < NKI; ++k) rnext[k] = ld_res4(r1 + (long)(k * RPI) * ldr);
                        }
                    } else {
                        const float* r0 = rptr(q);
#pragma unroll
                        for (int k = 0; k < NKI; ++k) rcur[k] = ld_res4(r0 + (long)(k * RPI) * ldr);
                    }
                }
                dump(i, g);
                const long mrow = mbase + i * 32 + rsub;
                const bool pad = p.out_op && p.map_op == ADA_MAP_PAD;
                PadWalk walk;
                if (pad) walk = pad_start(p, (uint32_t)mrow);
#pragma unroll
                for (int k = 0; k < NKI; ++k) {
                    float4 v = *(const float4*)(slab + (k * RPI + rsub) * SW + 4 * cg);
                    v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                    if constexpr (EPI == EPI_GELU) {
                        gelu_erf4(v);
                    }
                    v.x = v.x * gamma4.x + rcur[k].x; v.y = v.y * gamma4.y + rcur[k].y;
                    v.z = v.z * gamma4.z + rcur[k].z; v.w = v.w * gamma4.w + rcur[k].w;
                    if (p.out_f32) {
                        float4 w = v;
                        if (relu_f) {
                            w.x = __builtin_fmaxf(w.x, 0.f); w.y = __builtin_fmaxf(w.y, 0.f);
                            w.z = __builtin_fmaxf(w.z, 0.f); w.w = __builtin_fmaxf(w.w, 0.f);
                        }
                        st_f32x4(p.out_f32 + (mrow + k * RPI) * ldf + n, w);
                    }
                    if (p.out_op) {
                        if (relu_o) {
                            v.x = __builtin_fmaxf(v.x, 0.f); v.y = __builtin_fmaxf(v.y, 0.f);
                            v.z = __builtin_fmaxf(v.z, 0.f); v.w = __builtin_fmaxf(v.w, 0.f);
                        }
                        if (pad) {
                            store_op4(p, p.out_op + walk.prow * ldo + n, n, v);
                            pad_step(p, walk, RPI);
                        } else {
                            store_op4(p, p.out_op + (mrow + k * RPI) * ldo + n, n, v);
                        }
                    }
                }
                if constexpr (AHEAD) {
                    if (has_res) {
#pragma unroll
                        for (int k = 0; k < NKI; ++k) rcur[k] = rnext[k];
                    }
                }
            }
        }
    } else if (!p.out_f32 && !(flags & ADA_EP_RESIDUAL) && (p.ldo_op & 7) == 0 && (EPI != EPI_SHUFFLE || (p.shuffle_c & 7) == 0)) {
        // ---- operand-only output: 8 columns per lane -> one 16-byte store per row segment ----------------
        constexpr int CG = GW / 8;       // 8-column groups per row (4 or 8)
        constexpr int RPI = 64 / CG;     // rows per wave-wide access (16 or 8)
        const int cg = lane % CG, rsub = lane / CG;
        const bool relu = (flags & ADA_EP_RELU_OP) != 0;
#pragma unroll
        for (int g = 0; g < NG; ++g) {
            const int n = nwave + g * GW + 8 * cg;
            const bool nval = n < p.N;
            const bool nval2 = n + 4 < p.N;
            float4 b0 = make_float4(0, 0, 0, 0), b1 = b0, g0 = make_float4(1, 1, 1, 1), g1 = g0;
            if (flags & ADA_EP_BIAS) {
                if (nval) b0 = *(const float4*)(p.bias + n);
                if (nval2) b1 = *(const float4*)(p.bias + n + 4);
            }
            if (flags & ADA_EP_GAMMA) {
                if (nval) g0 = *(const float4*)(p.gamma + n);
                if (nval2) g1 = *(const float4*)(p.gamma + n + 4);
            }
            uint32_t sh_i = 0, sh_j = 0, sh_co = 0;
            if constexpr (EPI == EPI_SHUFFLE) {
                uint32_t ij;
                fast_divmod((uint32_t)(nval ? n : 0), p.dShC, ij, sh_co);
                fast_divmod(ij, p.dShS, sh_i, sh_j);
            }
#pragma unroll
            for (int i = 0; i < TI; ++i) {
                dump(i, g);
#pragma unroll
                for (int k = 0; k < 32 / RPI; ++k) {
                    const int row = k * RPI + rsub;
                    const int m = mbase + i * 32 + row;
                    float4 v0 = *(const float4*)(slab + row * SW + 8 * cg);
                    float4 v1 = *(const float4*)(slab + row * SW + 8 * cg + 4);
                    if (p.bias_row_mod > 0) {   // one bias vector per group of rows (the class-token read-out: a per-image bias, DA2/dpt.py:164-167)
                        uint32_t grp, rr_;
                        fast_divmod((uint32_t)(m < p.M ? m : p.M - 1), p.dBiasMod, grp, rr_);
                        const float* brow = p.bias + (long)grp * p.N;
                        b0 = nval ? *(const float4*)(brow + n) : make_float4(0, 0, 0, 0);
                        b1 = nval2 ? *(const float4*)(brow + n + 4) : make_float4(0, 0, 0, 0);
                    }
                    v0.x += b0.x; v0.y += b0.y; v0.z += b0.z; v0.w += b0.w;
                    v1.x += b1.x; v1.y += b1.y; v1.z += b1.z; v1.w += b1.w;
                    if constexpr (EPI == EPI_GELU) {
                        gelu_erf4(v0);
                        gelu_erf4(v1);
                    }
                    v0.x *= g0.x; v0.y *= g0.y; v0.z *= g0.z; v0.w *= g0.w;
                    v1.x *= g1.x; v1.y *= g1.y; v1.z *= g1.z; v1.w *= g1.w;
                    if (relu) {
                        v0.x = __builtin_fmaxf(v0.x, 0.f); v0.y = __builtin_fmaxf(v0.y, 0.f); v0.z = __builtin_fmaxf(v0.z, 0.f); v0.w = __builtin_fmaxf(v0.w, 0.f);
                        v1.x = __builtin_fmaxf(v1.x, 0.f); v1.y = __builtin_fmaxf(v1.y, 0.f); v1.z = __builtin_fmaxf(v1.z, 0.f); v1.w = __builtin_fmaxf(v1.w, 0.f);
                    }
                    if (m < p.M && nval) {
                        long orow;
                        int ocol = n;
                        if constexpr (EPI == EPI_SHUFFLE) {
                            uint32_t sb, rem, sy, sx;
                            fast_divmod((uint32_t)m, p.dMapHW, sb, rem);
                            fast_divmod(rem, p.dMapW, sy, sx);
                            orow = ((long)sb * (p.shuffle_s * p.map_h + 2) + (p.shuffle_s * sy + sh_i + 1)) * (p.shuffle_s * p.map_w + 2) +
                                   (p.shuffle_s * sx + sh_j + 1);
                            ocol = (int)sh_co;
                        } else {
                            orow = map_row(p, p.map_op, (uint32_t)m);
                        }
                        op_t* dst = p.out_op + orow * p.ldo_op + ocol;
                        if (nval2) store_op8(p, dst, ocol, v0, v1);
                        else store_op4<EPI != EPI_SHUFFLE>(p, dst, ocol, v0);   // (behind a shuffle the fp8 form needs shuffle_c % 8 == 0 -- validated -- so it never gets here:
                                                                                 //  with its code in this branch too the 256x256 shuffle kernel spills two registers)
                    }
                }
            }
        }
    } else {
        // ---- fp32 output and/or residual: 4 columns per lane; the residual loads of a whole 32-row pass are issued
        //      together (clamped, unconditional) one pass ahead, so their HBM latency hides behind the LDS transpose ----
        constexpr int CG = GW / 4;       // float4 column groups per row (8 or 16)
        constexpr int RPI = 64 / CG;     // rows covered by one wave-wide float4 read (8 or 4)
        constexpr int NKI = 32 / RPI;
        constexpr int NPASS = NG * TI;   // pass index q = g * TI + i
        const int cg = lane % CG, rsub = lane / CG;
        const bool has_res = (flags & ADA_EP_RESIDUAL) != 0;
        auto col_of = [&](int g) { return nwave + g * GW + 4 * cg; };
        auto res_ptr = [&](int q, int k) -> const float* {
            const int g = q / TI, i = q - g * TI;
            int m = mbase + i * 32 + k * RPI + rsub;
            if (m >= p.M) m = p.M - 1;
            int nc = col_of(g);
            if (nc >= p.N) nc = 0;
            long rrow;
            if (p.res_row_mod > 0) {
                uint32_t qq, rr;
                fast_divmod((uint32_t)m, p.dResMod, qq, rr);
                rrow = (long)rr + p.res_row_off;
            } else {
                rrow = map_row(p, p.map_f32, (uint32_t)m);
            }
            return p.res + rrow * p.ldr + nc;
        };
        // one-pass-ahead prefetch only where a single workgroup owns the CU; with 2-3 co-resident workgroups the other
        // workgroups' MFMAs already cover the latency and the 32 extra VGPRs would spill
        constexpr bool AHEAD = (NWAVES == 8 && TJ <= 2) || PIPE4;
        float4 rcur[NKI], rnext[AHEAD ? NKI : 1];
#pragma unroll
        for (int k = 0; k < NKI; ++k) rcur[k] = make_float4(0, 0, 0, 0);
        if (AHEAD && has_res) {
#pragma unroll
            for (int k = 0; k < NKI; ++k) rcur[k] = *(const float4*)res_ptr(0, k);
        }
#pragma unroll
        for (int q = 0; q < NPASS; ++q) {
            const int g = q / TI, i = q - g * TI;
            const int n = col_of(g);
            const bool nval = n < p.N;       // N % 4 == 0 is checked on the host
            const int nc = nval ? n : 0;
            float4 bias4 = make_float4(0, 0, 0, 0), gamma4 = make_float4(1, 1, 1, 1);
            if (flags & ADA_EP_BIAS) bias4 = *(const float4*)(p.bias + nc);
            if (flags & ADA_EP_GAMMA) gamma4 = *(const float4*)(p.gamma + nc);
            uint32_t sh_i = 0, sh_j = 0, sh_co = 0;
            if constexpr (EPI == EPI_SHUFFLE) {
                uint32_t ij;
                fast_divmod((uint32_t)nc, p.dShC, ij, sh_co);
                fast_divmod(ij, p.dShS, sh_i, sh_j);
            }
            if constexpr (AHEAD) {
                if (has_res && q + 1 < NPASS) {
#pragma unroll
                    for (int k = 0; k < NKI; ++k) rnext[k] = *(const float4*)res_ptr(q + 1, k);
                }
            } else {
                if (has_res) {
#pragma unroll
                    for (int k = 0; k < NKI; ++k) rcur[k] = *(const float4*)res_ptr(q, k);
                }
            }
            dump(i, g);
#pragma unroll
            for (int k = 0; k < NKI; ++k) {
                const int row = k * RPI + rsub;
                const int m = mbase + i * 32 + row;
                float4 v = *(const float4*)(slab + row * SW + 4 * cg);
                v.x += bias4.x; v.y += bias4.y; v.z += bias4.z; v.w += bias4.w;
                if constexpr (EPI == EPI_GELU) {
                    gelu_erf4(v);
                }
                v.x = v.x * gamma4.x + rcur[k].x; v.y = v.y * gamma4.y + rcur[k].y;
                v.z = v.z * gamma4.z + rcur[k].z; v.w = v.w * gamma4.w + rcur[k].w;
                if (m < p.M && nval) {
                    if (p.out_f32) {
                        const long frow = map_row(p, p.map_f32, (uint32_t)m);
                        float4 w = v;
                        if (flags & ADA_EP_RELU_F32) {
                            w.x = __builtin_fmaxf(w.x, 0.f); w.y = __builtin_fmaxf(w.y, 0.f);
                            w.z = __builtin_fmaxf(w.z, 0.f); w.w = __builtin_fmaxf(w.w, 0.f);
                        }
                        *(float4*)(p.out_f32 + frow * p.ldo_f32 + n) = w;
                    }
                    if (p.out_op) {
                        if (flags & ADA_EP_RELU_OP) {
                            v.x = __builtin_fmaxf(v.x, 0.f); v.y = __builtin_fmaxf(v.y, 0.f);
                            v.z = __builtin_fmaxf(v.z, 0.f); v.w = __builtin_fmaxf(v.w, 0.f);
                        }
                        long orow;
                        int ocol = n;
                        if constexpr (EPI == EPI_SHUFFLE) {
                            uint32_t sb, rem, sy, sx;
                            fast_divmod((uint32_t)m, p.dMapHW, sb, rem);
                            fast_divmod(rem, p.dMapW, sy, sx);
                            orow = ((long)sb * (p.shuffle_s * p.map_h + 2) + (p.shuffle_s * sy + sh_i + 1)) *
                                       (p.shuffle_s * p.map_w + 2) + (p.shuffle_s * sx + sh_j + 1);
                            ocol = (int)sh_co;
                        } else {
                            orow = map_row(p, p.map_op, (uint32_t)m);
                        }
                        store_op4<EPI != EPI_SHUFFLE>(p, p.out_op + orow * p.ldo_op + ocol, ocol, v);
                    }
                }
            }
            if constexpr (AHEAD) {
                if (has_res) {
#pragma unroll
                    for (int k = 0; k < NKI; ++k) rcur[k] = rnext[k];
                }
            }
        }
    }
    if (!PIPE4 && p.dbg && tid == 0) {
        const unsigned long long t_issued = __builtin_amdgcn_s_memtime();   // epilogue instructions issued, stores in flight
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned long long* d = p.dbg + (long)blockIdx.x * 8;
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        d[0] = t_entry; d[1] = t_first; d[2] = t_loop; d[3] = __builtin_amdgcn_s_memtime();
        d[4] = ((unsigned long long)xcc << 32) | hwid; d[5] = t_issued;
        d[6] = t_vm; d[7] = t_bar;
    }
}

static std::atomic<int> g_group_override{0};  // debug: force the column-group width (0 = model)
static thread_local int g_last_tile = -1;   // tile configuration of the calling thread's most recent launch (ada_debug_last_tile)

template <int BM, int BN, int BK, int WAVES_M, int WAVES_N, int EPI, int LOOP = 0>
int launch_cfg(IgemmDev& d, hipStream_t stream) {
    constexpr int NT = WAVES_M * WAVES_N * 64;
    constexpr int SMEM = 2 * (BM + BN) * BK * 2;
    d.tiles_m = (d.M + BM - 1) / BM;
    d.tiles_n = (d.N + BN - 1) / BN;
    {
        // L2 traffic model (bytes missing the XCD L2s): row-major re-streams W once per XCD per round of tiles when W exceeds
        // the L2; column groups of width g keep g weight slabs resident but re-read the A panels once per group.
        const double a_bytes = (double)d.M * (d.a_mode == ADA_A_PLAIN ? d.K : d.lda) * 2.0;
        const double slab = (double)BN * d.K * 2.0, w_bytes = slab * d.tiles_n;
        const double rounds = (double)(((long)d.tiles_m * d.tiles_n + 255) / 256);
        double best = a_bytes + (w_bytes > 3.0e6 ? rounds * 8.0 * w_bytes : 8.0 * w_bytes);
        int gbest = d.tiles_n;
        for (int g = 1; g < d.tiles_n; ++g) {
            if (g * slab > 2.6e6 && g > 1) break;
            const double groups = (double)((d.tiles_n + g - 1) / g);
            const double est = groups * a_bytes + 8.0 * w_bytes;
            if (est < 0.9 * best) { best = est; gbest = g; }
        }
        const int go = g_group_override.load(std::memory_order_relaxed);
        d.group_n = go > 0 ? (go < d.tiles_n ? go : d.tiles_n) : gbest;
    }
    auto kern = igemm_kernel<BM, BN, BK, WAVES_M, WAVES_N, EPI, LOOP>;
    static std::once_flag attr_once;   // one per template instantiation; concurrent first calls are serialised
    std::call_once(attr_once, [&]() {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            (void)hipGetLastError();
        }
    });
    g_last_tile = (BM == 256 && BN == 32 ? 0 : BM == 128 && BN == 64 ? 1 : BM == 256 && BN == 128 ? 2 : BM == 256 && BN == 256 ? 3 : 4) + 100 * LOOP;
    const long nblk = (long)d.tiles_m * d.tiles_n;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(NT), SMEM, stream, d);
    return ada_check_launch("ada_igemm");
}

// tile configurations: 0: 256x32, 1: 128x64, 2: 256x128, 3: 256x256 (1 WG/CU), 4: 128x128 (2 WG/CU)
// Relative time of launching `tiles` workgroups of a BMxBN tile with `occ` workgroups resident per CU and main-loop
// efficiency `eff` (measured, relative to the 256x256 tile): whole rounds of 256*occ tiles, co-resident tiles share a CU.
static inline double tile_time(long M, long N, int bm, int bn, int occ, double eff) {
    const long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const double t = (double)bm * bn / eff;
    const long slots = 256L * occ;
    if (tiles <= slots) return ((tiles + 255) / 256) * t;   // partially filled single round
    return (double)((tiles + slots - 1) / slots) * occ * t;
}
// Second estimate for the co-resident small tiles (round 5, profiles/r05_e_config2_shapes.txt): the busiest CU runs ceil(tiles / 256) of them whatever
// the residency -- the slot-rounded model above charges 2064 tiles of 128x128 five rounds of 512 slots (2560) where the CUs see 8.06 -> 9 tiles each --
// with a main-loop efficiency that FALLS with K (fitted on the ViT-B bs=8 and ViT-L bs=32 sweeps: 128x128 0.89 / 0.70 / 0.58 of the 256x256 tile at
// K = 768 / 2304 / 9216): short-K launches with few 256x256 tiles (ViT-B at bs=8: 129 tiles on 256 CUs) go to the small tiles, long-K ones never do.
static inline double tile_time_cu(long M, long N, long K, int bm, int bn, double e0, double slope, double emax) {
    const long tiles = ((M + bm - 1) / bm) * ((N + bn - 1) / bn);
    const double lg = K > 512 ? log2((double)K / 512.0) : 0.0;
    double eff = e0 - slope * lg;
    eff = eff > emax ? emax : (eff < 0.45 ? 0.45 : eff);
    return (double)((tiles + 255) / 256) * bm * bn / eff;
}

// main loop of the 256x256 tile: variant 16 forces the hand-scheduled 4-wave loop, 4 the single-barrier 8-wave loop; 0 (default) picks the
// 4-wave loop where its main loop outweighs its slower prologue / epilogue (measured: k-loops of >= 128 k-tiles)
// (its scalar A-offset counters assume a monotonic walk: a split operand, whose third k segment re-reads the first, stays on the 8-wave loop)
static inline bool use_pipe4(const IgemmDev& d) {
    if (d.a_dup_seg != 0 || d.a_wrap != 0 || d.f8_from != 0 || d.tap_cols != 0) return false;
    if ((d.flags & ADA_EP_SWIGLU) && d.split_seg != 0) return false;     // the SwiGLU epilogue of the 4-wave kernel writes the plain form only
    if (d.variant != 0) return d.variant >= 16;
    if (d.K >= 8192) return true;
    // With a cheap operand-typed epilogue (bias, SwiGLU: no fp32 output, no residual / LayerScale / GELU / row statistics) the slower epilogue of the
    // one-wave-per-SIMD loop weighs less and it wins from 24 k-steps on launches of many rounds: raw ViT-G at 8 x 1022^2, w12 (N = 8192, K = 1536)
    // 966 -> 909 us, qkv 526 -> 515 (profiles/r05_j_config5_shapes.txt).  At K = 1024 (ViT-L: qkv -3 %, fc1 + GELU -7 %) it still loses.
    const bool cheap = d.a_mode == ADA_A_PLAIN && d.out_f32 == nullptr && d.bias_row_mod == 0 &&
                       !(d.flags & (ADA_EP_RESIDUAL | ADA_EP_GAMMA | ADA_EP_GELU | ADA_EP_RELU_OP));
    return cheap && d.K >= 1536 && (long)((d.M + 255) / 256) * ((d.N + 255) / 256) >= 2048;
}

template <int EPI>
int launch_epi(IgemmDev& d, hipStream_t s, int force) {
    int cfg;
    if (d.N <= 32) cfg = 0;
    else if (d.N <= 64) cfg = 1;
    else if (d.N <= 128) {
        // two co-resident 128x128 workgroups per CU: -12 % vs the 512x128 / 256x128 tiles at every batch size -- unless that leaves more than half of
        // the CUs without a tile: then 128x64 tiles (ViT-B head at bs <= 8, single images: -25 ... -33 % per launch, profiles/r05_e_*)
        cfg = (EPI != EPI_SWIGLU && ((long)(d.M + 127) / 128) <= 128) ? 1 : 4;
    } else {
        // large problems: the 256x256 tile (best MFMA efficiency); small ones (single images, ViT-S/B at small batch)
        // would leave most CUs idle with it, so pick the tile that minimises the quantised time estimate
        cfg = 3;
        double best = tile_time(d.M, d.N, 256, 256, 1, 1.0);
        // relative main-loop efficiencies re-fitted after the move to 16x16x32 MFMAs (tools/autotune_shapes.py at B = 1 and 4:
        // the 128x64 tile with three co-resident workgroups wins more of the small problems than it used to)
        const long Kt = d.K;     // k-loop length (all taps / split segments)
        // (very long k-loops -- the 41472-deep split-precision 3x3 convs of the raw ViT-G head: the small tiles' efficiency is down to ~0.45 of the 256x256
        //  tile's, profiles/r05_j_config5_shapes.txt: 128x64 3220 us against 2133 us there -- in the slot-rounded estimate too)
        const double e4 = Kt >= 16384 ? 0.45 : 0.85, e1 = Kt >= 16384 ? 0.45 : 0.80;
        double t2 = tile_time(d.M, d.N, 256, 128, 1, 0.72), t4 = tile_time(d.M, d.N, 128, 128, 2, e4), t1 = tile_time(d.M, d.N, 128, 64, 3, e1);
        const double t4c = tile_time_cu(d.M, d.N, Kt, 128, 128, 0.95, 0.095, 0.90), t1c = tile_time_cu(d.M, d.N, Kt, 128, 64, 0.90, 0.10, 0.85);
        if (t4c < t4) t4 = t4c;     // either estimate may make the case for the small tile
        if (t1c < t1) t1 = t1c;
        if (t2 < 0.95 * best) { best = t2; cfg = 2; }
        if (t4 < 0.95 * best) { best = t4; cfg = 4; }
        if (EPI != EPI_SWIGLU && t1 < 0.90 * best) { best = t1; cfg = 1; }   // the small tile loses on long k-loops at equal estimate
    }
    if (d.M < 256 && cfg >= 2 && cfg != 4) cfg = 4;
    if (force >= 0 && force <= 4) cfg = force;
    if constexpr (EPI == EPI_SWIGLU) {
        if (cfg == 4 || cfg == 2) return launch_cfg<128, 128, 64, 2, 2, EPI>(d, s);
        if (use_pipe4(d)) return launch_cfg<256, 256, 64, 2, 2, EPI, 2>(d, s);
        return launch_cfg<256, 256, 64, 2, 4, EPI>(d, s);
    } else if constexpr (EPI == EPI_TAIL) {
        return cfg == 0 ? launch_cfg<256, 32, 64, 4, 1, EPI>(d, s) : launch_cfg<128, 64, 64, 4, 1, EPI>(d, s);
    } else {
        switch (cfg) {
            case 0: return launch_cfg<256, 32, 64, 4, 1, EPI>(d, s);
            case 1: return launch_cfg<128, 64, 64, 4, 1, EPI>(d, s);
            case 2: return launch_cfg<256, 128, 64, 4, 2, EPI>(d, s);
            case 4: return launch_cfg<128, 128, 64, 2, 2, EPI>(d, s);
            default:
                if (use_pipe4(d)) return launch_cfg<256, 256, 64, 2, 2, EPI, 2>(d, s);
                return launch_cfg<256, 256, 64, 2, 4, EPI>(d, s);
        }
    }
}

}  // namespace

// ---- tuning / diagnostic hooks (declared in include/ada_hip.h; process-global atomics, not needed for correct operation) ----
static std::atomic<unsigned long long*> g_dbg{nullptr};
static std::atomic<int> g_force_tile{-1};
static std::atomic<int> g_variant{0};   // main loop of the 256x256 tile -- 0: by shape (default); 4: single-barrier 8-wave loop; 16: hand-scheduled 4-wave loop
static std::once_flag g_env_once;
// A/B switches for kernel experiments: the environment (ADA_IGEMM_TILE / _GROUP / _VARIANT) presets the hooks below ONCE per process, and it
// does so before the first explicit ada_debug_set_* call as well as before the first launch -- an explicit call always has the last word
// (a preset applied lazily at the first ada_igemm used to overwrite hooks set before it).
static void apply_env_presets() {
    std::call_once(g_env_once, []() {
        if (const char* e = getenv("ADA_IGEMM_TILE")) g_force_tile.store(atoi(e), std::memory_order_relaxed);
        if (const char* gr = getenv("ADA_IGEMM_GROUP")) g_group_override.store(atoi(gr), std::memory_order_relaxed);
        if (const char* va = getenv("ADA_IGEMM_VARIANT")) {
            const int v = atoi(va);
            g_variant.store(v >= 16 ? 16 : v >= 4 ? 4 : 0, std::memory_order_relaxed);
        }
    });
}
// debug hook (not part of the stable ABI): override the tile configuration (-1 = heuristic)
extern "C" void ada_debug_set_tile(int cfg) { apply_env_presets(); g_force_tile.store(cfg, std::memory_order_relaxed); }
extern "C" void ada_debug_set_variant(int v) { apply_env_presets(); g_variant.store(v >= 16 ? 16 : v >= 4 ? 4 : 0, std::memory_order_relaxed); }
extern "C" void ada_debug_set_group(int g) { apply_env_presets(); g_group_override.store(g, std::memory_order_relaxed); }
extern "C" int ada_debug_last_tile(void) { return g_last_tile; }
// debug hook (not part of the stable ABI): device buffer of 8 x u64 per workgroup, or NULL to disable
extern "C" void ada_debug_set_timestamps(void* dev_buf) { g_dbg.store((unsigned long long*)dev_buf, std::memory_order_relaxed); }

extern "C" int ada_igemm(const ada_igemm_args* a, void* stream) {
    ADA_REQUIRE(a != nullptr, ADA_EINVAL, "ada_igemm: null args");
    ADA_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, ADA_EINVAL, "ada_igemm: M/N/K must be positive (%d,%d,%d)", a->M, a->N, a->K);
    ADA_REQUIRE(a->A && a->W, ADA_EINVAL, "ada_igemm: null operand pointer");
    ADA_REQUIRE(a->K % 64 == 0, ADA_EINVAL, "ada_igemm: K=%d must be a multiple of 64", a->K);
    ADA_REQUIRE(a->lda % 8 == 0 && a->lda > 0, ADA_EINVAL, "ada_igemm: lda=%ld must be a positive multiple of 8", (long)a->lda);
    ADA_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, ADA_EINVAL, "ada_igemm: operands must be 16-byte aligned");
    ADA_REQUIRE(a->out_f32 || a->out_op, ADA_EINVAL, "ada_igemm: no output buffer");
    ADA_REQUIRE((long)a->M < (1L << 24), ADA_EUNSUPPORTED, "ada_igemm: M=%d exceeds 2^24 rows", a->M);
    ADA_REQUIRE(a->a_dup_seg >= 0 && a->a_dup_seg % 64 == 0, ADA_EINVAL, "ada_igemm: a_dup_seg=%d must be a non-negative multiple of 64", a->a_dup_seg);
    if (a->a_dup_seg > 0) {
        const long taps = a->a_mode == ADA_A_CONV3 ? 9 : 1;
        ADA_REQUIRE((long)a->K == 3L * taps * a->a_dup_seg && a->lda >= 2L * a->a_dup_seg, ADA_EINVAL,
                    "ada_igemm: a split A operand needs K == %ld * a_dup_seg and lda >= 2 * a_dup_seg (K=%d a_dup_seg=%d lda=%ld)", 3 * taps, a->K, a->a_dup_seg, (long)a->lda);
    }
    if (a->a_mode == ADA_A_CONV3) {
        ADA_REQUIRE(a->lda % 64 == 0, ADA_EINVAL, "ada_igemm: CONV3 needs lda %% 64 == 0 (got %ld)", (long)a->lda);
        ADA_REQUIRE(a->a_dup_seg > 0 || a->K == 9 * a->lda, ADA_EINVAL, "ada_igemm: CONV3 needs K == 9*lda (K=%d lda=%ld)", a->K, (long)a->lda);
        ADA_REQUIRE(a->Ho > 0 && a->Wo > 0 && (a->stride == 1 || a->stride == 2), ADA_EINVAL, "ada_igemm: bad conv geometry");
        ADA_REQUIRE(a->Hp >= (a->Ho - 1) * a->stride + 3 && a->Wp >= (a->Wo - 1) * a->stride + 3, ADA_EINVAL,
                    "ada_igemm: padded input %dx%d too small for output %dx%d stride %d", a->Hp, a->Wp, a->Ho, a->Wo, a->stride);
        ADA_REQUIRE(a->M % (a->Ho * a->Wo) == 0, ADA_EINVAL, "ada_igemm: M must be batch*Ho*Wo");
    } else {
        ADA_REQUIRE(a->a_mode == ADA_A_PLAIN, ADA_EINVAL, "ada_igemm: unknown a_mode %d", a->a_mode);
        ADA_REQUIRE(a->a_dup_seg > 0 || a->a_wrap > 0 || a->lda >= a->K, ADA_EINVAL, "ada_igemm: lda=%ld < K=%d", (long)a->lda, a->K);
    }
    const int f = a->flags;
    ADA_REQUIRE(!(f & ADA_EP_BIAS) || a->bias, ADA_EINVAL, "ada_igemm: EP_BIAS without bias");
    ADA_REQUIRE(!(f & ADA_EP_GAMMA) || a->gamma, ADA_EINVAL, "ada_igemm: EP_GAMMA without gamma");
    ADA_REQUIRE(!(f & ADA_EP_RESIDUAL) || a->res, ADA_EINVAL, "ada_igemm: EP_RESIDUAL without res");
    const bool tail = (f & ADA_EP_TAIL) != 0, swiglu = (f & ADA_EP_SWIGLU) != 0;
    const bool shuffle = a->out_op && a->map_op == ADA_MAP_SHUFFLE;
    ADA_REQUIRE(a->N % 4 == 0, ADA_EUNSUPPORTED, "ada_igemm: N=%d must be a multiple of 4", a->N);
    ADA_REQUIRE(!a->out_f32 || tail || (a->ldo_f32 % 4 == 0 && ((uintptr_t)a->out_f32 % 16) == 0), ADA_EINVAL, "ada_igemm: out_f32 must be 16-byte aligned with ldo %% 4 == 0");
    ADA_REQUIRE(!a->out_op || (a->ldo_op % 4 == 0 && ((uintptr_t)a->out_op % 8) == 0), ADA_EINVAL, "ada_igemm: out_op must be 8-byte aligned with ldo %% 4 == 0");
    ADA_REQUIRE(!(f & ADA_EP_RESIDUAL) || (a->ldr % 4 == 0 && ((uintptr_t)a->res % 16) == 0), ADA_EINVAL, "ada_igemm: res must be 16-byte aligned with ldr %% 4 == 0");
    ADA_REQUIRE(!(f & ADA_EP_BIAS) || ((uintptr_t)a->bias % 16) == 0, ADA_EINVAL, "ada_igemm: bias must be 16-byte aligned");
    ADA_REQUIRE(!(f & ADA_EP_GAMMA) || ((uintptr_t)a->gamma % 16) == 0, ADA_EINVAL, "ada_igemm: gamma must be 16-byte aligned");
    if (tail) {
        ADA_REQUIRE(a->tail_w && a->out_f32, ADA_EINVAL, "ada_igemm: EP_TAIL needs tail_w and out_f32");
        ADA_REQUIRE(((uintptr_t)a->tail_w % 16) == 0, ADA_EINVAL, "ada_igemm: tail_w must be 16-byte aligned");
        ADA_REQUIRE(a->N <= 64, ADA_EUNSUPPORTED, "ada_igemm: EP_TAIL supports N <= 64 (got %d)", a->N);
        ADA_REQUIRE(!swiglu && !shuffle && !(f & ADA_EP_GELU), ADA_EUNSUPPORTED, "ada_igemm: EP_TAIL cannot be combined with other epilogues");
    }
    if (swiglu) {
        ADA_REQUIRE(a->N % 64 == 0 && a->out_op && !a->out_f32, ADA_EINVAL, "ada_igemm: EP_SWIGLU needs N %% 64 == 0 and only out_op");
        ADA_REQUIRE(a->map_op == ADA_MAP_PLAIN && !(f & (ADA_EP_GELU | ADA_EP_GAMMA | ADA_EP_RESIDUAL)), ADA_EUNSUPPORTED,
                    "ada_igemm: EP_SWIGLU supports bias + MAP_PLAIN only");
    }
    const int maps_needing_grid = (a->out_op && (a->map_op == ADA_MAP_PAD || a->map_op == ADA_MAP_SHUFFLE));
    if (maps_needing_grid) {
        ADA_REQUIRE(a->map_h > 0 && a->map_w > 0 && a->M % (a->map_h * a->map_w) == 0, ADA_EINVAL, "ada_igemm: bad output grid %dx%d for M=%d", a->map_h, a->map_w, a->M);
    }
    if (shuffle) {
        ADA_REQUIRE(a->shuffle_s > 0 && a->shuffle_c > 0 && a->N == a->shuffle_s * a->shuffle_s * a->shuffle_c, ADA_EINVAL,
                    "ada_igemm: SHUFFLE needs N == s*s*c");
        ADA_REQUIRE(a->shuffle_c % 4 == 0 && !a->out_f32 && !(f & ADA_EP_GELU), ADA_EUNSUPPORTED, "ada_igemm: SHUFFLE needs c %% 4 == 0, operand output only");
    }
    const int split_abs = a->split_seg < 0 ? -a->split_seg : a->split_seg;   // < 0: the [hi | lo8 | hi8] form
    if (a->split_seg != 0) {
        ADA_REQUIRE(a->out_op && split_abs % 8 == 0, ADA_EINVAL, "ada_igemm: split_seg needs out_op and a multiple of 8");
        // (behind a shuffle only the 8-column operand-only epilogue writes the [hi | lo8 | hi8] form: its conditions are required here, or the 4-column
        //  path -- compiled without the fp8 form for EPI_SHUFFLE -- would write [hi | lo] where the consumer reads bytes)
        ADA_REQUIRE(a->split_seg > 0 || !shuffle || (a->shuffle_c % 8 == 0 && !(f & ADA_EP_RESIDUAL) && a->ldo_op % 8 == 0), ADA_EUNSUPPORTED,
                    "ada_igemm: the fp8 form of a split output (split_seg < 0) behind MAP_SHUFFLE needs shuffle_c %% 8 == 0, ldo_op %% 8 == 0 and no residual");
        const int cols = shuffle ? a->shuffle_c : swiglu ? a->N / 2 : a->N;
        ADA_REQUIRE(cols <= split_abs && a->ldo_op >= 2L * split_abs, ADA_EINVAL, "ada_igemm: split_seg=%d too small for %d columns / ldo_op=%ld", a->split_seg, cols, (long)a->ldo_op);
    }
    if (a->bias_row_mod != 0) {
        ADA_REQUIRE(a->bias_row_mod > 0 && (f & ADA_EP_BIAS) && a->out_op && !a->out_f32 && !(f & ADA_EP_RESIDUAL) && !tail && !swiglu && !shuffle &&
                    a->ldo_op % 8 == 0 && a->N % 4 == 0, ADA_EUNSUPPORTED,
                    "ada_igemm: bias_row_mod (a bias vector per group of rows) is implemented for operand-typed outputs with ldo_op %% 8 == 0 (bias / GELU epilogues)");
    }
    if (a->a_wrap != 0) {
        ADA_REQUIRE(a->a_mode == ADA_A_PLAIN && a->a_dup_seg == 0 && a->a_wrap > 0 && a->a_wrap % 64 == 0 && a->K == 2 * a->a_wrap && a->lda >= a->a_wrap, ADA_EINVAL,
                    "ada_igemm: a_wrap=%d needs a plain operand, K == 2 * a_wrap (K=%d) and lda >= a_wrap", a->a_wrap, a->K);
    }
    if (a->f8_from != 0) {
        const long per = a->a_mode == ADA_A_CONV3 ? a->lda : (long)a->K;     // period of the k-walk in operand slots
        ADA_REQUIRE(a->a_dup_seg == 0 && a->a_wrap == 0, ADA_EINVAL, "ada_igemm: f8_from excludes a_dup_seg / a_wrap");
        ADA_REQUIRE(a->f8_from > 0 && a->f8_from % 64 == 0 && a->f8_mid % 64 == 0 && a->f8_from <= a->f8_mid && a->f8_mid <= per && a->f8_from < per, ADA_EINVAL,
                    "ada_igemm: f8_from=%d / f8_mid=%d must be multiples of 64 with 0 < f8_from <= f8_mid <= %ld (the period of the k-walk)", a->f8_from, a->f8_mid, per);
    }
    if (a->tap_cols != 0) {
        ADA_REQUIRE(a->a_mode == ADA_A_CONV3 && a->tap_cols > 0 && a->N % a->tap_cols == 0 && a->N / a->tap_cols <= 16, ADA_EINVAL,
                    "ada_igemm: tap_cols=%d needs a CONV3 operand and N = (1..16) * tap_cols (N=%d)", a->tap_cols, a->N);
        for (int i = 0; i < a->N / a->tap_cols; ++i)
            ADA_REQUIRE(a->tap_mask[i] != 0 && a->tap_mask[i] < 512, ADA_EINVAL, "ada_igemm: tap_mask[%d]=0x%x must name 1..9 of the taps", i, (unsigned)a->tap_mask[i]);
    }
    if ((a->out_f32 && a->map_f32 == ADA_MAP_TOKEN) || (a->out_op && a->map_op == ADA_MAP_TOKEN)) {
        ADA_REQUIRE(a->map_h > 0 && a->M % a->map_h == 0, ADA_EINVAL, "ada_igemm: TOKEN map needs map_h = patches per image");
    }
    ADA_REQUIRE(!a->out_f32 || a->map_f32 == ADA_MAP_PLAIN || a->map_f32 == ADA_MAP_TOKEN, ADA_EUNSUPPORTED, "ada_igemm: fp32 output supports PLAIN/TOKEN maps");

    IgemmDev d;
    d.M = a->M; d.N = a->N; d.K = a->K; d.a_mode = a->a_mode;
    d.A = (const op_t*)a->A; d.lda = a->lda;
    d.Ho = a->Ho; d.Wo = a->Wo; d.Hp = a->Hp; d.Wp = a->Wp; d.stride = a->stride;
    d.dWo = make_fastdiv(a->Wo > 0 ? a->Wo : 1);
    d.dHoWo = make_fastdiv(a->Ho > 0 && a->Wo > 0 ? a->Ho * a->Wo : 1);
    d.W = (const op_t*)a->W;
    d.bias = a->bias; d.gamma = a->gamma; d.res = a->res; d.ldr = a->ldr;
    d.res_row_mod = a->res_row_mod; d.res_row_off = a->res_row_off;
    d.dResMod = make_fastdiv(a->res_row_mod > 0 ? a->res_row_mod : 1);
    d.flags = a->flags;
    d.out_f32 = a->out_f32; d.ldo_f32 = a->ldo_f32; d.map_f32 = a->map_f32;
    d.out_op = (op_t*)a->out_op; d.ldo_op = a->ldo_op; d.map_op = a->map_op;
    d.map_h = a->map_h; d.map_w = a->map_w;
    const bool token = (a->out_f32 && a->map_f32 == ADA_MAP_TOKEN) || (a->out_op && a->map_op == ADA_MAP_TOKEN);
    d.dMapW = make_fastdiv(a->map_w > 0 ? a->map_w : 1);
    d.dMapHW = make_fastdiv(token ? a->map_h : (a->map_h > 0 && a->map_w > 0 ? a->map_h * a->map_w : 1));
    d.shuffle_s = a->shuffle_s; d.shuffle_c = a->shuffle_c;
    d.split_seg = split_abs;
    d.split_f8 = a->split_seg < 0;
    d.f8_from = a->f8_from / 64; d.f8_mid = a->f8_mid / 64; d.f8_scales = a->f8_scales;
    d.a_dup_seg = a->a_dup_seg;
    d.a_wrap = a->a_wrap;
    d.bias_row_mod = a->bias_row_mod;
    d.dBiasMod = make_fastdiv(a->bias_row_mod > 0 ? a->bias_row_mod : 1);
    d.tap_cols = a->tap_cols;
    d.tap_bits[0] = d.tap_bits[1] = d.tap_bits[2] = 0;
    if (a->tap_cols > 0)
        for (int i = 0; i < a->N / a->tap_cols; ++i) d.tap_bits[i / 7] |= (unsigned long long)a->tap_mask[i] << (9 * (i % 7));
    d.dShC = make_fastdiv(a->shuffle_c > 0 ? a->shuffle_c : 1);
    d.dShS = make_fastdiv(a->shuffle_s > 0 ? a->shuffle_s : 1);
    d.tail_w = a->tail_w; d.tail_b = a->tail_b; d.tail_act = a->tail_act;
    d.cps = 0;
    d.group_n = 1;
    d.tiles_m = d.tiles_n = 0;

    apply_env_presets();
    const int force = g_force_tile.load(std::memory_order_relaxed);
    d.dbg = g_dbg.load(std::memory_order_relaxed);
    d.variant = g_variant.load(std::memory_order_relaxed);
    hipStream_t s = (hipStream_t)stream;
    if (tail) return launch_epi<EPI_TAIL>(d, s, a->N <= 32 ? 0 : 1);
    if (swiglu) return launch_epi<EPI_SWIGLU>(d, s, force);
    if (shuffle) return launch_epi<EPI_SHUFFLE>(d, s, force);
    if (f & ADA_EP_GELU) return launch_epi<EPI_GELU>(d, s, force);
    return launch_epi<EPI_STD>(d, s, force);
}
