// Implicit-GEMM contraction with fused epilogues (see include/ada_hip.h: ada_igemm).
//
// Tiling (gfx950): a workgroup of 4 waves computes a BM x BN output tile with BK = 64; each wave owns
// TI x TJ MFMA tiles of 32x32 (v_mfma_f32_32x32x16, fp32 accumulate).  A and W k-slabs (rows of 64
// operands = 128 B) are copied HBM -> LDS with 16-byte global_load_lds (no VGPR round trip), two LDS
// stages, one barrier per k-step.  LDS rows are stored linearly (a global_load_lds constraint: the
// destination is wave base + lane*16) but each lane *fetches* the 16-byte chunk
// c ^ ((row>>1)&7) of its row, and the MFMA fragment reads apply the same XOR, which makes every
// ds_read_b128 lane group hit 16 distinct 16-byte bank slots (cdna_hip_programming.md T2 / rule 21).
//
// For a 3x3 convolution the A slab of k-step (tap, kc) is the same 128-byte row segment shifted by
// (dy*Wp + dx) pixels in the zero-bordered NHWC input, so the gather costs one scalar add per k-step.
#include <stdarg.h>
#include "ada_common.h"

namespace {

constexpr int BK = 64;

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
    FastDiv dShC, dShS;
    const float* tail_w;
    float tail_b;
    int tail_act;
    int tiles_m, tiles_n;
    int cps;  // k-steps per conv tap = lda / 64
};

ADA_DEV float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

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

template <int WAVES_M, int WAVES_N, int TI, int TJ>
__global__ __launch_bounds__(256, 2) void igemm_kernel(IgemmDev p) {
    constexpr int BM = WAVES_M * TI * 32;
    constexpr int BN = WAVES_N * TJ * 32;
    constexpr int A_BYTES = BM * BK * 2;
    constexpr int B_BYTES = BN * BK * 2;
    constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    constexpr int A_IT = BM / 32;  // 32 rows (x 8 chunks) per 256-thread pass
    constexpr int B_IT = BN / 32;
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    // XCD-aware (bijective) block remap: each XCD's L2 sees a contiguous run of tiles that share A panels.
    int tm, tn;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        tm = logical / p.tiles_n;
        tn = logical - tm * p.tiles_n;
    }
    const int m0 = tm * BM, n0 = tn * BN;

    // ---- per-thread staging addresses --------------------------------------------------
    const int srow = tid >> 3;                         // row inside a 32-row pass
    const int gchunk = (tid & 7) ^ ((tid >> 4) & 7);   // swizzled source chunk of that row
    const op_t* a_ptr[A_IT];
    const op_t* b_ptr[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        uint32_t m = (uint32_t)(m0 + it * 32 + srow);
        if (m >= (uint32_t)p.M) m = (uint32_t)p.M - 1;
        long base;
        if (p.a_mode == ADA_A_PLAIN) {
            base = (long)m * p.lda;
        } else {
            uint32_t b, rem, y, x;
            fast_divmod(m, p.dHoWo, b, rem);
            fast_divmod(rem, p.dWo, y, x);
            base = (((long)b * p.Hp + (long)y * p.stride) * p.Wp + (long)x * p.stride) * p.lda;
        }
        a_ptr[it] = p.A + base + gchunk * 8;
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
        int n = n0 + it * 32 + srow;
        if (n >= p.N) n = p.N - 1;
        b_ptr[it] = p.W + (long)n * p.K + gchunk * 8;
    }

    auto stage = [&](int buf, int kt) {
        long aoff;
        if (p.a_mode == ADA_A_PLAIN) {
            aoff = (long)kt * BK;
        } else {
            const int tap = kt / p.cps;
            const int kc = kt - tap * p.cps;
            const int dy = tap / 3, dx = tap - dy * 3;
            aoff = ((long)dy * p.Wp + dx) * p.lda + (long)kc * BK;
        }
        const long boff = (long)kt * BK;
        char* sa = smem + buf * STAGE_BYTES + wave * 1024;
        char* sb = sa + A_BYTES;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(a_ptr[it] + aoff),
                                             (__attribute__((address_space(3))) void*)(sa + it * 4096), 16, 0, 0);
        }
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(b_ptr[it] + boff),
                                             (__attribute__((address_space(3))) void*)(sb + it * 4096), 16, 0, 0);
        }
    };

    // ---- main loop -----------------------------------------------------------------------
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    f32x16 acc[TI][TJ];
#pragma unroll
    for (int i = 0; i < TI; ++i)
#pragma unroll
        for (int j = 0; j < TJ; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

    const int swz = (l31 >> 1) & 7;
    // fragment row byte offsets inside a stage (row * 128 B)
    const int a_row_off = (wm * TI * 32 + l31) * 128;
    const int b_row_off = A_BYTES + (wn * TJ * 32 + l31) * 128;

    const int nk = p.K / BK;
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (kt + 1 < nk) stage(cur ^ 1, kt + 1);
        const char* sbase = smem + cur * STAGE_BYTES;
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const int coff = ((2 * s + hi) ^ swz) * 16;
            opx8 af[TI], bf[TJ];
#pragma unroll
            for (int i = 0; i < TI; ++i) af[i] = *(const opx8*)(sbase + a_row_off + i * 32 * 128 + coff);
#pragma unroll
            for (int j = 0; j < TJ; ++j) bf[j] = *(const opx8*)(sbase + b_row_off + j * 32 * 128 + coff);
#pragma unroll
            for (int i = 0; i < TI; ++i)
#pragma unroll
                for (int j = 0; j < TJ; ++j) acc[i][j] = mfma32(af[i], bf[j], acc[i][j]);
        }
    }

    // ---- epilogue ------------------------------------------------------------------------
    const int flags = p.flags;
    int ncol[TJ];
    float biasv[TJ], gammav[TJ], tailw[TJ];
    bool nvalid[TJ];
#pragma unroll
    for (int j = 0; j < TJ; ++j) {
        ncol[j] = n0 + (wn * TJ + j) * 32 + l31;
        nvalid[j] = ncol[j] < p.N;
        const int nc = nvalid[j] ? ncol[j] : 0;
        biasv[j] = (flags & ADA_EP_BIAS) ? p.bias[nc] : 0.0f;
        gammav[j] = (flags & ADA_EP_GAMMA) ? p.gamma[nc] : 1.0f;
        tailw[j] = ((flags & ADA_EP_TAIL) && nvalid[j]) ? p.tail_w[nc] : 0.0f;
    }

    if (flags & ADA_EP_TAIL) {
        // relu(conv + bias) . tail_w + tail_b -> activation; one output per GEMM row.  N <= BN, WAVES_N == 1.
#pragma unroll
        for (int i = 0; i < TI; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + (wm * TI + i) * 32 + crow32(r, hi);
                float part = 0.0f;
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    float v = acc[i][j][r] + biasv[j];
                    part += __builtin_fmaxf(v, 0.0f) * tailw[j];
                }
                // sum over the 32 lanes that share this row (xor shuffles stay inside a 32-lane half)
                part += __shfl_xor(part, 1);
                part += __shfl_xor(part, 2);
                part += __shfl_xor(part, 4);
                part += __shfl_xor(part, 8);
                part += __shfl_xor(part, 16);
                if (l31 == 0 && m < p.M) {
                    float d = part + p.tail_b;
                    if (p.tail_act == ADA_ACT_SIGMOID) d = 1.0f / (1.0f + __expf(-d));
                    else if (p.tail_act == ADA_ACT_RELU) d = __builtin_fmaxf(d, 0.0f);
                    p.out_f32[m] = d;
                }
            }
        }
        return;
    }

    if (flags & ADA_EP_SWIGLU) {
        // packer interleaves the w12 rows in 32-wide groups: MFMA tile j = x1 group, tile j+1 = its x2 group
        if constexpr (TJ % 2 == 0) {
#pragma unroll
            for (int i = 0; i < TI; ++i) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * TI + i) * 32 + crow32(r, hi);
                    if (m < p.M) {
#pragma unroll
                        for (int j = 0; j < TJ; j += 2) {
                            if (nvalid[j]) {
                                const float x1 = acc[i][j][r] + biasv[j];
                                const float x2 = acc[i][j + 1][r] + biasv[j + 1];
                                const float g = x1 / (1.0f + __expf(-x1)) * x2;
                                const int nh = (ncol[j] >> 6) * 32 + l31;  // hidden column
                                p.out_op[(long)m * p.ldo_op + nh] = to_op(g);
                            }
                        }
                    }
                }
            }
        }
        return;
    }

#pragma unroll
    for (int i = 0; i < TI; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * TI + i) * 32 + crow32(r, hi);
            if (m < p.M) {
                long frow = 0, orow = 0, rrow = 0;
                uint32_t sb = 0, sy = 0, sx = 0;
                if (p.out_f32 || (flags & ADA_EP_RESIDUAL)) frow = map_row(p, p.map_f32, (uint32_t)m);
                if (flags & ADA_EP_RESIDUAL) {
                    if (p.res_row_mod > 0) {
                        uint32_t qq, rr;
                        fast_divmod((uint32_t)m, p.dResMod, qq, rr);
                        rrow = (long)rr + p.res_row_off;
                    } else {
                        rrow = frow;
                    }
                }
                if (p.out_op) {
                    if (p.map_op == ADA_MAP_SHUFFLE) {
                        uint32_t rem;
                        fast_divmod((uint32_t)m, p.dMapHW, sb, rem);
                        fast_divmod(rem, p.dMapW, sy, sx);
                    } else {
                        orow = map_row(p, p.map_op, (uint32_t)m);
                    }
                }
#pragma unroll
                for (int j = 0; j < TJ; ++j) {
                    if (nvalid[j]) {
                        const int n = ncol[j];
                        float v = acc[i][j][r] + biasv[j];
                        if (flags & ADA_EP_GELU) v = gelu_erf(v);
                        v *= gammav[j];
                        if (flags & ADA_EP_RESIDUAL) v += p.res[rrow * p.ldr + n];
                        if (p.out_f32) p.out_f32[frow * p.ldo_f32 + n] = (flags & ADA_EP_RELU_F32) ? __builtin_fmaxf(v, 0.0f) : v;
                        if (p.out_op) {
                            const float vo = (flags & ADA_EP_RELU_OP) ? __builtin_fmaxf(v, 0.0f) : v;
                            if (p.map_op == ADA_MAP_SHUFFLE) {
                                uint32_t ij, co, ii, jj;
                                fast_divmod((uint32_t)n, p.dShC, ij, co);
                                fast_divmod(ij, p.dShS, ii, jj);
                                const long prow = ((long)sb * (p.shuffle_s * p.map_h + 2) + (p.shuffle_s * sy + ii + 1)) *
                                                      (p.shuffle_s * p.map_w + 2) +
                                                  (p.shuffle_s * sx + jj + 1);
                                p.out_op[prow * p.ldo_op + co] = to_op(vo);
                            } else {
                                p.out_op[orow * p.ldo_op + n] = to_op(vo);
                            }
                        }
                    }
                }
            }
        }
    }
}

template <int WAVES_M, int WAVES_N, int TI, int TJ>
int launch_igemm(IgemmDev& d, hipStream_t stream) {
    constexpr int BM = WAVES_M * TI * 32;
    constexpr int BN = WAVES_N * TJ * 32;
    constexpr int SMEM = 2 * (BM + BN) * BK * 2;
    d.tiles_m = (d.M + BM - 1) / BM;
    d.tiles_n = (d.N + BN - 1) / BN;
    auto kern = igemm_kernel<WAVES_M, WAVES_N, TI, TJ>;
    static bool attr_done = false;
    if (!attr_done) {
        if (hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, SMEM) != hipSuccess) {
            (void)hipGetLastError();
        }
        attr_done = true;
    }
    const long nblk = (long)d.tiles_m * d.tiles_n;
    hipLaunchKernelGGL(kern, dim3((unsigned)nblk), dim3(256), SMEM, stream, d);
    return ada_check_launch("ada_igemm");
}

}  // namespace

extern "C" int ada_igemm(const ada_igemm_args* a, void* stream) {
    ADA_REQUIRE(a != nullptr, ADA_EINVAL, "ada_igemm: null args");
    ADA_REQUIRE(a->M > 0 && a->N > 0 && a->K > 0, ADA_EINVAL, "ada_igemm: M/N/K must be positive (%d,%d,%d)", a->M, a->N, a->K);
    ADA_REQUIRE(a->A && a->W, ADA_EINVAL, "ada_igemm: null operand pointer");
    ADA_REQUIRE(a->K % BK == 0, ADA_EINVAL, "ada_igemm: K=%d must be a multiple of 64", a->K);
    ADA_REQUIRE(a->lda % 8 == 0 && a->lda > 0, ADA_EINVAL, "ada_igemm: lda=%ld must be a positive multiple of 8", (long)a->lda);
    ADA_REQUIRE(((uintptr_t)a->A % 16) == 0 && ((uintptr_t)a->W % 16) == 0, ADA_EINVAL, "ada_igemm: operands must be 16-byte aligned");
    ADA_REQUIRE(a->out_f32 || a->out_op, ADA_EINVAL, "ada_igemm: no output buffer");
    ADA_REQUIRE((long)a->M < (1L << 24), ADA_EUNSUPPORTED, "ada_igemm: M=%d exceeds 2^24 rows", a->M);
    if (a->a_mode == ADA_A_CONV3) {
        ADA_REQUIRE(a->lda % BK == 0, ADA_EINVAL, "ada_igemm: CONV3 needs lda %% 64 == 0 (got %ld)", (long)a->lda);
        ADA_REQUIRE(a->K == 9 * a->lda, ADA_EINVAL, "ada_igemm: CONV3 needs K == 9*lda (K=%d lda=%ld)", a->K, (long)a->lda);
        ADA_REQUIRE(a->Ho > 0 && a->Wo > 0 && (a->stride == 1 || a->stride == 2), ADA_EINVAL, "ada_igemm: bad conv geometry");
        ADA_REQUIRE(a->Hp >= (a->Ho - 1) * a->stride + 3 && a->Wp >= (a->Wo - 1) * a->stride + 3, ADA_EINVAL,
                    "ada_igemm: padded input %dx%d too small for output %dx%d stride %d", a->Hp, a->Wp, a->Ho, a->Wo, a->stride);
        ADA_REQUIRE(a->M % (a->Ho * a->Wo) == 0, ADA_EINVAL, "ada_igemm: M must be batch*Ho*Wo");
    } else {
        ADA_REQUIRE(a->a_mode == ADA_A_PLAIN, ADA_EINVAL, "ada_igemm: unknown a_mode %d", a->a_mode);
        ADA_REQUIRE(a->lda >= a->K, ADA_EINVAL, "ada_igemm: lda=%ld < K=%d", (long)a->lda, a->K);
    }
    const int f = a->flags;
    ADA_REQUIRE(!(f & ADA_EP_BIAS) || a->bias, ADA_EINVAL, "ada_igemm: EP_BIAS without bias");
    ADA_REQUIRE(!(f & ADA_EP_GAMMA) || a->gamma, ADA_EINVAL, "ada_igemm: EP_GAMMA without gamma");
    ADA_REQUIRE(!(f & ADA_EP_RESIDUAL) || a->res, ADA_EINVAL, "ada_igemm: EP_RESIDUAL without res");
    if (f & ADA_EP_TAIL) {
        ADA_REQUIRE(a->tail_w && a->out_f32, ADA_EINVAL, "ada_igemm: EP_TAIL needs tail_w and out_f32");
        ADA_REQUIRE(a->N <= 64, ADA_EUNSUPPORTED, "ada_igemm: EP_TAIL supports N <= 64 (got %d)", a->N);
    }
    if (f & ADA_EP_SWIGLU) {
        ADA_REQUIRE(a->N % 64 == 0 && a->out_op && !a->out_f32, ADA_EINVAL, "ada_igemm: EP_SWIGLU needs N %% 64 == 0 and only out_op");
        ADA_REQUIRE(a->map_op == ADA_MAP_PLAIN, ADA_EUNSUPPORTED, "ada_igemm: EP_SWIGLU supports MAP_PLAIN only");
    }
    const int maps_needing_grid = (a->out_op && (a->map_op == ADA_MAP_PAD || a->map_op == ADA_MAP_SHUFFLE));
    if (maps_needing_grid) {
        ADA_REQUIRE(a->map_h > 0 && a->map_w > 0 && a->M % (a->map_h * a->map_w) == 0, ADA_EINVAL, "ada_igemm: bad output grid %dx%d for M=%d", a->map_h, a->map_w, a->M);
    }
    if (a->out_op && a->map_op == ADA_MAP_SHUFFLE) {
        ADA_REQUIRE(a->shuffle_s > 0 && a->shuffle_c > 0 && a->N == a->shuffle_s * a->shuffle_s * a->shuffle_c, ADA_EINVAL,
                    "ada_igemm: SHUFFLE needs N == s*s*c");
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
    d.dShC = make_fastdiv(a->shuffle_c > 0 ? a->shuffle_c : 1);
    d.dShS = make_fastdiv(a->shuffle_s > 0 ? a->shuffle_s : 1);
    d.tail_w = a->tail_w; d.tail_b = a->tail_b; d.tail_act = a->tail_act;
    d.cps = (int)(a->lda / BK);

    hipStream_t s = (hipStream_t)stream;
    if (a->N <= 32) return launch_igemm<4, 1, 2, 1>(d, s);   // 256 x 32 tile
    if (a->N <= 64) return launch_igemm<4, 1, 1, 2>(d, s);   // 128 x 64 tile
    return launch_igemm<2, 2, 2, 2>(d, s);                                        // 128 x 128 tile
}
