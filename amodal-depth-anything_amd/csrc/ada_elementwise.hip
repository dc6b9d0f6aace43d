// HBM-bound kernels of the path: LayerNorm, patchify (im2col + ImageNet normalise), cls row, bilinear
// resize (align_corners=True) with fused skip add.  See include/ada_hip.h for the reference call sites.
// All of them move each byte once with 16-byte (fp32) / 8-byte (operand) accesses; row statistics use
// wavefront shuffles only (one wave per row, no LDS).
#include <atomic>
#include <mutex>
#include <stdlib.h>
#include "ada_common.h"

namespace {

// operand-typed float4 store; split_seg > 0 writes [hi | lo] column segments (split precision, see ada_igemm_args.split_seg), < 0 the fp8 form
ADA_DEV void store_op4_split(op_t* row, int c, float4 r, int split_seg) {
    opx4 o;
    o[0] = to_op(r.x); o[1] = to_op(r.y); o[2] = to_op(r.z); o[3] = to_op(r.w);
    ((opx4*)row)[c] = o;
    if (split_seg > 0) {
        opx4 l;
        l[0] = to_op(r.x - (float)o[0]); l[1] = to_op(r.y - (float)o[1]); l[2] = to_op(r.z - (float)o[2]); l[3] = to_op(r.w - (float)o[3]);
        ((opx4*)(row + split_seg))[c] = l;
    } else if (split_seg < 0) {   // [hi | lo8 | hi8]: e5m2 bytes behind the hi segment (ada_igemm_args.f8_from)
        uint32_t* b = (uint32_t*)(row - split_seg);
        b[c] = bf8x4((r.x - (float)o[0]) * ADA_F8_LO_SHIFT, (r.y - (float)o[1]) * ADA_F8_LO_SHIFT, (r.z - (float)o[2]) * ADA_F8_LO_SHIFT, (r.w - (float)o[3]) * ADA_F8_LO_SHIFT);
        b[c + (-split_seg) / 4] = bf8x4(r.x, r.y, r.z, r.w);
    }
}

// The resize kernels' outputs are written once and read by a later launch: non-temporal stores keep them
// from displacing the input rows the neighbouring workgroups still read (profiles/r03_o, +0.3 % end to end).
ADA_DEV void store_op4_bil(op_t* row, int c, float4 r, int split_seg) {
    opx4 o;
    o[0] = to_op(r.x); o[1] = to_op(r.y); o[2] = to_op(r.z); o[3] = to_op(r.w);
    __builtin_nontemporal_store(o, ((opx4*)row) + c);
    if (split_seg > 0) {
        opx4 l;
        l[0] = to_op(r.x - (float)o[0]); l[1] = to_op(r.y - (float)o[1]); l[2] = to_op(r.z - (float)o[2]); l[3] = to_op(r.w - (float)o[3]);
        __builtin_nontemporal_store(l, ((opx4*)(row + split_seg)) + c);
    } else if (split_seg < 0) {
        uint32_t* b = (uint32_t*)(row - split_seg);
        b[c] = bf8x4((r.x - (float)o[0]) * ADA_F8_LO_SHIFT, (r.y - (float)o[1]) * ADA_F8_LO_SHIFT, (r.z - (float)o[2]) * ADA_F8_LO_SHIFT, (r.w - (float)o[3]) * ADA_F8_LO_SHIFT);
        b[c + (-split_seg) / 4] = bf8x4(r.x, r.y, r.z, r.w);
    }
}
ADA_DEV void store_f32_bil(float* ptr, float4 r) {
    const f32x4 t = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(t, (f32x4*)ptr);
}

constexpr int LN_MAX_CHUNKS = 6;  // float4 chunks per lane: dim <= 6 * 64 * 4 = 1536

ADA_DEV float wave_sum(float v) {
    v += __shfl_xor(v, 32);
    v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

template <int LPR>
ADA_DEV float row_sum(float v) {      // sum over the LPR lanes that hold one row
    if constexpr (LPR == 64) v += __shfl_xor(v, 32);
    if constexpr (LPR == 64) v += __shfl_xor(v, 16);
    v += __shfl_xor(v, 8);
    v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 2);
    v += __shfl_xor(v, 1);
    return v;
}

ADA_DEV long pad_row(uint32_t m, int h, int w, FastDiv dW, FastDiv dHW) {
    uint32_t b, rem, y, x;
    fast_divmod(m, dHW, b, rem);
    fast_divmod(rem, dW, y, x);
    return ((long)b * (h + 2) + (y + 1)) * (w + 2) + (x + 1);
}

struct LnArgs {
    const float* in;
    long ld_in;
    int rows_out, dim, group_in, skip;
    FastDiv dGroupOut;
    const float* weight;
    const float* bias;
    float eps;
    op_t* out_op;
    long ld_op;
    int map_op, map_h, map_w;
    FastDiv dMapW, dMapHW;
    int relu;
    float* out_f32;
    long ld_f32;
    int split_seg;
    // second normalised output of the same rows (same statistics, its own gain / bias): rows are taken in groups of out2_group, the first
    // out2_skip rows of each group are left out and the rest compacted -- the tap LayerNorm (cls token dropped) rides on the next block's LN1
    const float* weight2;
    const float* bias2;
    op_t* out2;
    long ld2;
    int out2_group, out2_skip, split_seg2;
    FastDiv dOut2Group;
    // input rows are fine-grid pixels of a sub-pixel convolution's [coarse pixel, s*s*dim] output (unshuffle_s = s > 0; map_h / map_w = fine grid)
    int unshuffle_s;
    FastDiv dUnS;
    const float* tap_bias;   // [s*s*dim, 9]: bias contribution of coarse tap t to column (phase, c); subtracted where the tap falls outside the grid
    int identity;            // 1: no normalisation (y = x): the kernel is then the un-shuffle / re-layout pass of a sub-pixel convolution's output
};

// LPR = lanes per row.  64: one wave per row (rows of 1024-1536 floats: 4-6 float4 per lane in flight).  16: a quarter wave per row, four rows per wave -- the
// head's channel LayerNorms run over 256 / 512-wide pixel rows, where one wave per row has a single 1 KB load in flight and the launch stopped at
// 3.1-3.7 TB/s; with four rows per wave every lane keeps 4-8 float4 in flight and the reductions stay inside a 16-lane DPP row.
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_kernel(LnArgs p) {
    constexpr int RPW = 64 / LPR;                       // rows per wave
    constexpr int MAXC = LPR == 64 ? LN_MAX_CHUNKS : 8;  // float4 chunks per lane: dim <= 1536 / 512
    const int lane = threadIdx.x & (LPR - 1);
    const int ro = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RPW + ((threadIdx.x & 63) / LPR);
    if (ro >= p.rows_out) return;                        // (the reductions below only exchange between the LPR lanes of one row)
    long in_row = ro;
    if (p.group_in > 0) {
        uint32_t g, w;
        fast_divmod((uint32_t)ro, p.dGroupOut, g, w);
        in_row = (long)g * p.group_in + p.skip + w;
    }
    const float4* src = (const float4*)(p.in + in_row * p.ld_in);
    const int nchunk = p.dim >> 2;
    // sub-pixel input: output row ro is fine pixel (b, Y, X); its dim values are columns [phase * dim, (phase + 1) * dim) of coarse row (b, Y / s, X / s).
    // On the outermost ring of the fine grid some coarse taps fall into the zero border, where the merged convolution must not see the
    // transposed conv's bias either: their share of the (interior) bias the GEMM added is taken out again here.
    unsigned ring = 0;          // bit t: coarse tap t is outside the grid for this pixel
    const float* tapb = nullptr;
    if (p.unshuffle_s > 0) {
        uint32_t b, rem, Y, X, cy, py, cx, px;
        fast_divmod((uint32_t)ro, p.dMapHW, b, rem);
        fast_divmod(rem, p.dMapW, Y, X);
        fast_divmod(Y, p.dUnS, cy, py);
        fast_divmod(X, p.dUnS, cx, px);
        const int ss = p.unshuffle_s, ch = p.map_h / ss, cw = p.map_w / ss;
        const int phase = (int)(py * ss + px);
        src = (const float4*)(p.in + (((long)b * ch + cy) * cw + cx) * p.ld_in + (long)phase * p.dim);
        if (p.tap_bias) {
            if (Y == 0) ring |= 0x007u;
            if ((int)Y == p.map_h - 1) ring |= 0x1c0u;
            if (X == 0) ring |= 0x049u;
            if ((int)X == p.map_w - 1) ring |= 0x124u;
            tapb = p.tap_bias + (long)phase * p.dim * 9;
        }
    }
    float4 v[MAXC];
    float s = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + LPR * i;
        if (c < nchunk) {
            v[i] = src[c];
            if (ring) {   // wave-uniform, ~2 % of the rows
                const float* tb = tapb + (long)c * 36;
                for (int t = 0; t < 9; ++t)
                    if ((ring >> t) & 1u) { v[i].x -= tb[t]; v[i].y -= tb[9 + t]; v[i].z -= tb[18 + t]; v[i].w -= tb[27 + t]; }
            }
            s += (v[i].x + v[i].y) + (v[i].z + v[i].w);
        } else {
            v[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    float mean = row_sum<LPR>(s) / (float)p.dim;
    float q = 0.0f;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + LPR * i;
        if (c < nchunk) {
            const float a = v[i].x - mean, b = v[i].y - mean, cc = v[i].z - mean, d = v[i].w - mean;
            q += (a * a + b * b) + (cc * cc + d * d);
        }
    }
    float rstd = 1.0f / sqrtf(row_sum<LPR>(q) / (float)p.dim + p.eps);
    if (p.identity) { mean = 0.0f; rstd = 1.0f; }
    long orow = ro;
    if (p.out_op && p.map_op == ADA_MAP_PAD) orow = pad_row((uint32_t)ro, p.map_h, p.map_w, p.dMapW, p.dMapHW);
    const float4* w4 = (const float4*)p.weight;
    const float4* b4 = (const float4*)p.bias;
#pragma unroll
    for (int i = 0; i < MAXC; ++i) {
        const int c = lane + LPR * i;
        if (c < nchunk) {
            float4 y = v[i];
            if (!p.identity) {
                const float4 w = w4[c], bb = b4[c];
                y.x = (v[i].x - mean) * rstd * w.x + bb.x;
                y.y = (v[i].y - mean) * rstd * w.y + bb.y;
                y.z = (v[i].z - mean) * rstd * w.z + bb.z;
                y.w = (v[i].w - mean) * rstd * w.w + bb.w;
            }
            if (p.relu == 1) {
                y.x = __builtin_fmaxf(y.x, 0.f); y.y = __builtin_fmaxf(y.y, 0.f);
                y.z = __builtin_fmaxf(y.z, 0.f); y.w = __builtin_fmaxf(y.w, 0.f);
            }
            if (p.out_f32) ((float4*)(p.out_f32 + (long)ro * p.ld_f32))[c] = y;
            if (p.relu == 2) {   // ReLU on the operand-typed copy only (the fp32 copy feeds a residual add: util/blocks.py:57-80)
                y.x = __builtin_fmaxf(y.x, 0.f); y.y = __builtin_fmaxf(y.y, 0.f);
                y.z = __builtin_fmaxf(y.z, 0.f); y.w = __builtin_fmaxf(y.w, 0.f);
            }
            if (p.out_op) store_op4_split(p.out_op + orow * p.ld_op, c, y, p.split_seg);
        }
    }
    if (p.out2) {
        uint32_t g, w;
        fast_divmod((uint32_t)ro, p.dOut2Group, g, w);
        if ((int)w < p.out2_skip) return;
        op_t* dst = p.out2 + ((long)g * (p.out2_group - p.out2_skip) + (w - p.out2_skip)) * p.ld2;
        const float4* w24 = (const float4*)p.weight2;
        const float4* b24 = (const float4*)p.bias2;
#pragma unroll
        for (int i = 0; i < MAXC; ++i) {
            const int c = lane + LPR * i;
            if (c < nchunk) {
                const float4 w = w24[c], bb = b24[c];
                float4 y;
                y.x = (v[i].x - mean) * rstd * w.x + bb.x;
                y.y = (v[i].y - mean) * rstd * w.y + bb.y;
                y.z = (v[i].z - mean) * rstd * w.z + bb.z;
                y.w = (v[i].w - mean) * rstd * w.w + bb.w;
                store_op4_split(dst, c, y, p.split_seg2);
            }
        }
    }
}

// one workgroup per patch row; a thread handles column pairs (dx, dx+1) of the same (c, dy)
__global__ __launch_bounds__(256) void patchify_kernel(const float* __restrict__ x, const float* __restrict__ guide, int cg,
                                                       int H, int W, int ph, int pw, float3 mean, float3 inv_std, int normalise,
                                                       op_t* __restrict__ out, long ld, int seg) {
    const int p = blockIdx.x;  // (b, py, px)
    const int px = p % pw;
    const int py = (p / pw) % ph;
    const int b = p / (pw * ph);
    const int kreal = (3 + cg) * 196;
    // seg == 0: one segment of ld columns.  seg > 0 (split precision): two segments of `seg` columns holding
    // [hi | lo] with hi = round(x), lo = round(x - hi): contracted as (hi, lo, hi) against weights [w_hi | w_hi | w_lo] (ada_igemm
    // a_dup_seg) one GEMM computes x_hi w_hi + x_lo w_hi + x_hi w_lo, i.e. the patch embedding to ~fp32 accuracy from fp16 MFMAs.
    const int width = seg > 0 ? seg : (int)ld;
    for (int pair = threadIdx.x; pair * 2 < width; pair += blockDim.x) {
        const int col = pair * 2;
        float a0 = 0.f, a1 = 0.f;
        if (col < kreal) {
            const int c = col / 196;
            const int rem = col - c * 196;
            const int dy = rem / 14, dx = rem - dy * 14;
            const long pix = (long)(py * 14 + dy) * W + px * 14 + dx;
            if (c < 3) {
                const float2 v = *(const float2*)(x + ((long)b * 3 + c) * H * W + pix);
                a0 = v.x; a1 = v.y;
                if (normalise) {
                    const float mu = c == 0 ? mean.x : (c == 1 ? mean.y : mean.z);
                    const float is = c == 0 ? inv_std.x : (c == 1 ? inv_std.y : inv_std.z);
                    a0 = (a0 - mu) * is;
                    a1 = (a1 - mu) * is;
                }
            } else {
                const float2 v = *(const float2*)(guide + ((long)b * cg + (c - 3)) * H * W + pix);
                a0 = v.x; a1 = v.y;
            }
        }
        opx2 o;
        o[0] = to_op(a0);
        o[1] = to_op(a1);
        *(opx2*)(out + (long)p * ld + col) = o;
        if (seg > 0) {
            opx2 lo;
            lo[0] = to_op(a0 - (float)o[0]);
            lo[1] = to_op(a1 - (float)o[1]);
            *(opx2*)(out + (long)p * ld + seg + col) = lo;
        }
    }
}

__global__ void write_cls_kernel(float* tokens, int batch, int n_tok, int dim, const float* cls, const float* pos0) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= batch * dim) return;
    const int b = i / dim, d = i - b * dim;
    tokens[(long)b * n_tok * dim + d] = cls[d] + pos0[d];
}

struct BilinearArgs {
    const float* in;
    long ld_in;
    int batch, hi, wi, ho, wo, c4;  // c4 = channels / 4
    float sy, sx;                   // (in-1)/(out-1), 0 when out == 1
    const float* add;
    long ld_add;
    float* out_f32;
    long ld_f32;
    op_t* out_op;
    long ld_op;
    int map_op, relu;
    int split_seg;
    FastDiv dC4, dWo, dHo;
};

// grid: (ceil(wo*c4 / 256), ho, batch) -- one thread per (x, 4-channel group) of an output row: the row's vertical
// taps are wave-uniform scalars, consecutive lanes walk the channels of one pixel then the next pixel (coalesced
// float4 loads from the two source rows, contiguous 8/16-byte stores).
__global__ __launch_bounds__(256) void bilinear_kernel(BilinearArgs p) {
    const uint32_t idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= (uint32_t)(p.wo * p.c4)) return;
    uint32_t x, c;
    fast_divmod(idx, p.dC4, x, c);
    const int y = blockIdx.y, b = blockIdx.z;
    const float fy = p.sy * (float)y, fx = p.sx * (float)x;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + (y0 < p.hi - 1 ? 1 : 0), x1 = x0 + (x0 < p.wi - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
    const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
    const long rb = (long)b * p.hi;
    const float* r0 = p.in + (rb + y0) * p.wi * p.ld_in;
    const float* r1 = p.in + (rb + y1) * p.wi * p.ld_in;
    const float4 v00 = ((const float4*)(r0 + (long)x0 * p.ld_in))[c];
    const float4 v01 = ((const float4*)(r0 + (long)x1 * p.ld_in))[c];
    const float4 v10 = ((const float4*)(r1 + (long)x0 * p.ld_in))[c];
    const float4 v11 = ((const float4*)(r1 + (long)x1 * p.ld_in))[c];
    float4 r;
    r.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
    r.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
    r.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
    r.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
    const long pix = ((long)b * p.ho + y) * p.wo + x;
    if (p.add) {
        const float4 a = ((const float4*)(p.add + pix * p.ld_add))[c];
        r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
    }
    if (p.out_f32) store_f32_bil(p.out_f32 + pix * p.ld_f32 + 4 * c, r);
    if (p.out_op) {
        long orow = pix;
        if (p.map_op == ADA_MAP_PAD) orow = ((long)b * (p.ho + 2) + (y + 1)) * (p.wo + 2) + (x + 1);
        if (p.relu) {
            r.x = __builtin_fmaxf(r.x, 0.f); r.y = __builtin_fmaxf(r.y, 0.f);
            r.z = __builtin_fmaxf(r.z, 0.f); r.w = __builtin_fmaxf(r.w, 0.f);
        }
        store_op4_bil(p.out_op + orow * p.ld_op, c, r, p.split_seg);
    }
}

// LDS-tiled variant for up-sampling wide-channel maps (the DPT fusion path: 128 / 256 channels).  One workgroup produces an
// 8 x 16 tile of output pixels for a chunk of 128 channels (8 x 16 measured best among 4x16 .. 8x64: 36 KB of LDS, four workgroups
// per CU overlap each other's staging).  The source patch the tile touches (at most 6 x 12 pixels at the scales used here) is copied ONCE into LDS by global_load_lds (16 B per lane, lane-linear: pixel-major, 512 B per pixel), then
// every output reads its four taps from LDS.  The per-pixel kernel above re-fetches each source pixel ~12x through L2 (the
// workgroups sharing a source row run on different CUs) and tops out near 3 TB/s; this one reads the source about once.
constexpr int BT_TH = 8, BT_TW = 16, BT_CG = 32;   // tile height / width in output pixels, float4 channel groups per chunk

__global__ __launch_bounds__(256) void bilinear_tiled_kernel(BilinearArgs p, int ph, int pw, int nchunk) {
    extern __shared__ __attribute__((aligned(16))) char bl_smem[];
    const int tid = threadIdx.x;
    const int g = tid & (BT_CG - 1), psub = tid >> 5;     // channel group, pixel slot (8 pixels per pass)
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.z / nchunk, chunk = blockIdx.z - b * nchunk;
    const int c0 = chunk * BT_CG;                           // first float4 channel group of this chunk
    const int tx0 = blockIdx.x * BT_TW, ty0 = blockIdx.y * BT_TH;
    const int py0 = (int)(p.sy * (float)ty0), px0 = (int)(p.sx * (float)tx0);   // top-left source pixel of the patch
    // ---- stage the patch: pass i copies patch pixels 8i .. 8i+7 (clamped to the image), 32 channel groups each ----------
    const int npix = ph * pw;
    const long img = (long)b * p.hi;
    for (int base = 0; base < npix; base += 8) {
        int pp = base + psub;
        if (pp >= npix) pp = npix - 1;
        const int r = pp / pw, c = pp - r * pw;
        int yy = py0 + r, xx = px0 + c;
        if (yy > p.hi - 1) yy = p.hi - 1;
        if (xx > p.wi - 1) xx = p.wi - 1;
        const float* src = p.in + ((img + yy) * p.wi + xx) * p.ld_in + 4 * (c0 + g);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                         (__attribute__((address_space(3))) void*)(bl_smem + base * 512 + wave * 1024), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // ---- 256 output pixels, 8 per pass --------------------------------------------------------------------------
#pragma unroll 4
    for (int it = 0; it < BT_TH * BT_TW / 8; ++it) {
        const int py = it / (BT_TW / 8), px = (it % (BT_TW / 8)) * 8 + psub;
        const int y = ty0 + py, x = tx0 + px;
        if (y >= p.ho || x >= p.wo) continue;
        const float fy = p.sy * (float)y, fx = p.sx * (float)x;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = y0 + (y0 < p.hi - 1 ? 1 : 0), x1 = x0 + (x0 < p.wi - 1 ? 1 : 0);
        const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
        const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
        const char* t0 = bl_smem + ((y0 - py0) * pw - px0) * 512 + g * 16;
        const char* t1 = bl_smem + ((y1 - py0) * pw - px0) * 512 + g * 16;
        const float4 v00 = *(const float4*)(t0 + x0 * 512);
        const float4 v01 = *(const float4*)(t0 + x1 * 512);
        const float4 v10 = *(const float4*)(t1 + x0 * 512);
        const float4 v11 = *(const float4*)(t1 + x1 * 512);
        float4 r;
        r.x = ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x);
        r.y = ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y);
        r.z = ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z);
        r.w = ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w);
        const int c = c0 + g;
        const long pix = ((long)b * p.ho + y) * p.wo + x;
        if (p.add) {
            const float4 a = ((const float4*)(p.add + pix * p.ld_add))[c];
            r.x += a.x; r.y += a.y; r.z += a.z; r.w += a.w;
        }
        if (p.out_f32) store_f32_bil(p.out_f32 + pix * p.ld_f32 + 4 * c, r);
        if (p.out_op) {
            long orow = pix;
            if (p.map_op == ADA_MAP_PAD) orow = ((long)b * (p.ho + 2) + (y + 1)) * (p.wo + 2) + (x + 1);
            if (p.relu) {
                r.x = __builtin_fmaxf(r.x, 0.f); r.y = __builtin_fmaxf(r.y, 0.f);
                r.z = __builtin_fmaxf(r.z, 0.f); r.w = __builtin_fmaxf(r.w, 0.f);
            }
            store_op4_bil(p.out_op + orow * p.ld_op, c, r, p.split_seg);
        }
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// 3x3 convolution of an align-corners up-sampled map, with the channel mixing done BEFORE the resize (ada_tapsum_resize_fwd).
//   conv3x3(bilinear_ac(z))[Y, X, co] = b[co] + sum over taps t = (dy, dx) of  bilinear_ac(W_t z)[Y + dy - 1, X + dx - 1, co]
// (a 1x1 channel mix commutes with a per-channel resample; positions outside the fine grid are the convolution's zero padding and drop out,
// constant terms included).  The nine tap maps T[:, t*C + co] = W_t z live on the COARSE grid -- one GEMM with N = 9 C, a quarter of the MACs
// of the 3x3 convolution on the 2x finer grid -- and this kernel gathers them: per output element 9 taps x 4 bilinear corners.
// One workgroup per 8 x 16 output tile and all C <= 128 channels; the coarse patch under the tile's halo is staged through LDS one tap row
// (three taps) at a time by global_load_lds; thread (pixel slot, 4-channel group) keeps its 16 output pixels' sums in registers.
constexpr int TS_TH = 8, TS_TW = 16;

struct TapSumArgs {
    const void* in;      // [B * hi * wi, ld_in] operand-typed or fp32, columns t * C + c
    long ld_in;
    int batch, hi, wi, ho, wo, C;
    float sy, sx;
    const float* bias;   // [C] or null
    float* out;          // [B * ho * wo, ld_out] fp32
    long ld_out;
    int ph, pw;          // patch extent (source pixels), maximised over all tiles by the host
};

// acc += w * (float)h.lo / h.hi with h a packed pair of operand-typed values: v_fma_mix_f32 reads the half directly (no v_cvt; hipcc's SLP
// vectoriser turns the plain C++ form into v_cvt_f32_f16 + v_pk_fma_f32, 2.5 x the issue cycles)
ADA_DEV void fma_mix_lo(float& acc, float w, uint32_t h) {
#ifdef ADA_OPERAND_BF16
    acc += w * __builtin_bit_cast(float, h << 16);
#else
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,0,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(w), "v"(h));
#endif
}
ADA_DEV void fma_mix_hi(float& acc, float w, uint32_t h) {
#ifdef ADA_OPERAND_BF16
    acc += w * __builtin_bit_cast(float, h & 0xffff0000u);
#else
    asm("v_fma_mix_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(acc) : "v"(w), "v"(h));
#endif
}

template <int C, typename TIN>
__global__ __launch_bounds__(256) void tapsum_resize_kernel(TapSumArgs p) {
    constexpr int ES = (int)sizeof(TIN);     // 2: operand-typed tap maps, 4: fp32 tap maps
    extern __shared__ __attribute__((aligned(16))) char ts_smem[];
    __shared__ float4 rowinfo[TS_TH + 2];    // per fine row of the halo: (byte offset of source row y0 in a tap's patch, of y1, ly1, valid)
    __shared__ float4 colinfo[TS_TW + 2];    // per fine column: (byte offset of source column x0, of x1, lx1, valid)
    constexpr int LPP = C * ES / 16;         // 16-byte lanes per source pixel and tap: 4 ... 32
    constexpr int PPP = 256 / LPP;           // pixels staged per workgroup pass
    constexpr int NG = C / 4;                // 4-channel groups
    constexpr int PIXB = C * ES;             // bytes per pixel and tap in LDS
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = tid & 31, psub = tid >> 5;           // compute role: channel group (idle when >= NG), pixel slot
    const int b = blockIdx.z;
    const int tx0 = blockIdx.x * TS_TW, ty0 = blockIdx.y * TS_TH;
    const int yf = ty0 > 0 ? ty0 - 1 : 0, xf = tx0 > 0 ? tx0 - 1 : 0;
    const int py0 = (int)(p.sy * (float)yf), px0 = (int)(p.sx * (float)xf);   // top-left source pixel of the patch
    const int npix = p.ph * p.pw;
    const int tapbytes = ((npix + PPP - 1) / PPP) * PPP * PIXB;
    if (tid < TS_TH + 2) {
        const int Y = ty0 - 1 + tid;
        const bool ok = Y >= 0 && Y < p.ho;
        // a fine row / column outside the image (the convolution's zero padding) carries weight 0, but the compute loop still issues the LDS
        // reads of a masked COLUMN: its offsets must point at staged data (the patch origin), never below it -- a source index of 0 would give
        // (0 - px0) * PIXB < 0, i.e. bytes of another tap's slab or of the tables below, and 0 * Inf / NaN bit patterns poison the edge pixel
        const float f = ok ? p.sy * (float)Y : (float)py0;
        const int y0 = (int)f, y1 = y0 + (ok && y0 < p.hi - 1 ? 1 : 0);
        rowinfo[tid] = make_float4(__builtin_bit_cast(float, (y0 - py0) * p.pw * PIXB), __builtin_bit_cast(float, (y1 - py0) * p.pw * PIXB), f - (float)y0, ok ? 1.0f : 0.0f);
    } else if (tid >= 64 && tid < 64 + TS_TW + 2) {
        const int i = tid - 64, X = tx0 - 1 + i;
        const bool ok = X >= 0 && X < p.wo;
        const float f = ok ? p.sx * (float)X : (float)px0;
        const int x0 = (int)f, x1 = x0 + (ok && x0 < p.wi - 1 ? 1 : 0);
        colinfo[i] = make_float4(__builtin_bit_cast(float, (x0 - px0) * PIXB), __builtin_bit_cast(float, (x1 - px0) * PIXB), f - (float)x0, ok ? 1.0f : 0.0f);
    }
    float4 acc[TS_TH * TS_TW / 8];
#pragma unroll
    for (int it = 0; it < TS_TH * TS_TW / 8; ++it) acc[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int spx = tid / LPP, sl = tid % LPP;         // staging role: pixel inside a pass, 16-byte piece of its C channels
    const long img = (long)b * p.hi;
    const unsigned lds_g = (unsigned)(size_t)ts_smem + (unsigned)(g * 4 * ES);
    // A thread's output pixels sit in two columns (psub, psub + 8) of the tile: the LDS addresses of the two source columns under each of the
    // 2 x 3 (column, tap) positions and their horizontal weights do not change over the tile's rows or the three tap rows -- kept in registers
    // (filled after the first barrier below, when colinfo is visible)
    unsigned ca0[2][3], ca1[2][3];
    float cw0[2][3], cw1[2][3];
    for (int dy = 0; dy < 3; ++dy) {
        // ---- stage the three taps (dy, 0..2) of the patch: LDS [tap][pixel][C] -------------------------------------------
        __syncthreads();     // the previous tap row has been consumed (and rowinfo / colinfo are written)
        for (int base = 0; base < npix; base += PPP) {
            int pp = base + spx;
            if (pp >= npix) pp = npix - 1;
            const int r = pp / p.pw, c = pp - r * p.pw;
            int yy = py0 + r, xx = px0 + c;
            if (yy > p.hi - 1) yy = p.hi - 1;
            if (xx > p.wi - 1) xx = p.wi - 1;
            const TIN* src = (const TIN*)p.in + ((img + yy) * p.wi + xx) * p.ld_in + dy * 3 * C + sl * (16 / ES);
#pragma unroll
            for (int t = 0; t < 3; ++t)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + t * C),
                                                 (__attribute__((address_space(3))) void*)(ts_smem + t * tapbytes + base * PIXB + wave * 1024), 16, 0, 0);
        }
        if (dy == 0) {
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float4 ci = colinfo[h * 8 + psub + t];
                    ca0[h][t] = lds_g + __builtin_bit_cast(unsigned, ci.x) + (unsigned)(t * tapbytes);
                    ca1[h][t] = lds_g + __builtin_bit_cast(unsigned, ci.y) + (unsigned)(t * tapbytes);
                    cw1[h][t] = ci.z * ci.w;                      // a fine column outside the image contributes nothing
                    cw0[h][t] = (1.0f - ci.z) * ci.w;
                }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (g < NG) {
#pragma unroll
            for (int it = 0; it < TS_TH * TS_TW / 8; ++it) {
                const int py = it >> 1, h = it & 1;
                const float4 ri = rowinfo[py + dy];
                if (ri.w == 0.0f) continue;                       // fine row outside the image: zero padding (workgroup-uniform)
                const unsigned r0 = __builtin_bit_cast(unsigned, ri.x), r1 = __builtin_bit_cast(unsigned, ri.y);
                const float ly1 = ri.z, ly0 = 1.0f - ly1;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const unsigned a00 = ca0[h][t] + r0, a01 = ca1[h][t] + r0, a10 = ca0[h][t] + r1, a11 = ca1[h][t] + r1;
                    const float w00 = ly0 * cw0[h][t], w01 = ly0 * cw1[h][t], w10 = ly1 * cw0[h][t], w11 = ly1 * cw1[h][t];
                    if constexpr (ES == 2) {
                        const u32x2 v00 = *(const __attribute__((address_space(3))) u32x2*)(size_t)a00, v01 = *(const __attribute__((address_space(3))) u32x2*)(size_t)a01;
                        const u32x2 v10 = *(const __attribute__((address_space(3))) u32x2*)(size_t)a10, v11 = *(const __attribute__((address_space(3))) u32x2*)(size_t)a11;
                        fma_mix_lo(acc[it].x, w00, v00[0]); fma_mix_hi(acc[it].y, w00, v00[0]); fma_mix_lo(acc[it].z, w00, v00[1]); fma_mix_hi(acc[it].w, w00, v00[1]);
                        fma_mix_lo(acc[it].x, w01, v01[0]); fma_mix_hi(acc[it].y, w01, v01[0]); fma_mix_lo(acc[it].z, w01, v01[1]); fma_mix_hi(acc[it].w, w01, v01[1]);
                        fma_mix_lo(acc[it].x, w10, v10[0]); fma_mix_hi(acc[it].y, w10, v10[0]); fma_mix_lo(acc[it].z, w10, v10[1]); fma_mix_hi(acc[it].w, w10, v10[1]);
                        fma_mix_lo(acc[it].x, w11, v11[0]); fma_mix_hi(acc[it].y, w11, v11[0]); fma_mix_lo(acc[it].z, w11, v11[1]); fma_mix_hi(acc[it].w, w11, v11[1]);
                    } else {
                        const f32x4 v00 = *(const __attribute__((address_space(3))) f32x4*)(size_t)a00, v01 = *(const __attribute__((address_space(3))) f32x4*)(size_t)a01;
                        const f32x4 v10 = *(const __attribute__((address_space(3))) f32x4*)(size_t)a10, v11 = *(const __attribute__((address_space(3))) f32x4*)(size_t)a11;
                        acc[it].x += w00 * v00[0] + w01 * v01[0] + w10 * v10[0] + w11 * v11[0];
                        acc[it].y += w00 * v00[1] + w01 * v01[1] + w10 * v10[1] + w11 * v11[1];
                        acc[it].z += w00 * v00[2] + w01 * v01[2] + w10 * v10[2] + w11 * v11[2];
                        acc[it].w += w00 * v00[3] + w01 * v01[3] + w10 * v10[3] + w11 * v11[3];
                    }
                }
            }
        }
    }
    if (g >= NG) return;
    float4 bias4 = make_float4(0.f, 0.f, 0.f, 0.f);
    if (p.bias) bias4 = ((const float4*)p.bias)[g];
#pragma unroll
    for (int it = 0; it < TS_TH * TS_TW / 8; ++it) {
        const int y = ty0 + (it >> 1), x = tx0 + (it & 1) * 8 + psub;
        if (y >= p.ho || x >= p.wo) continue;
        float4 r = acc[it];
        r.x += bias4.x; r.y += bias4.y; r.z += bias4.z; r.w += bias4.w;
        store_f32_bil(p.out + (((long)b * p.ho + y) * p.wo + x) * p.ld_out + 4 * g, r);
    }
}

// Bicubic resample of the learned position table to another patch grid (reference DA2/dinov2.py:199-230: F.interpolate(mode="bicubic",
// antialias=False, scale_factor=...), i.e. ATen's upsample_bicubic2d with align_corners=False): cubic-convolution weights with A = -0.75,
// source coordinate (dst + 0.5) * (1 / scale_factor) - 0.5 (not clamped), tap indices clamped to the grid.  One thread per (output
// token, 4 channels); row 0 (cls position) is copied.  A parameter transform -- it runs once per grid size, not per image.
ADA_DEV void cubic_coeffs(float t, float (&c)[4]) {
    const float A = -0.75f;
    const float x1 = t, x2 = 1.0f - t;
    c[0] = ((A * (x1 + 1.0f) - 5.0f * A) * (x1 + 1.0f) + 8.0f * A) * (x1 + 1.0f) - 4.0f * A;
    c[1] = ((A + 2.0f) * x1 - (A + 3.0f)) * x1 * x1 + 1.0f;
    c[2] = ((A + 2.0f) * x2 - (A + 3.0f)) * x2 * x2 + 1.0f;
    c[3] = ((A * (x2 + 1.0f) - 5.0f * A) * (x2 + 1.0f) + 8.0f * A) * (x2 + 1.0f) - 4.0f * A;
}

__global__ __launch_bounds__(256) void pos_embed_resize_kernel(const float* __restrict__ pos, int sq, int dim, int ph, int pw, float inv_sh,
                                                               float inv_sw, float* __restrict__ out) {
    const int c4 = dim >> 2;
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)(1 + ph * pw) * c4) return;
    const int tok = (int)(idx / c4), c = (int)(idx - (long)tok * c4);
    if (tok == 0) {
        ((float4*)out)[c] = ((const float4*)pos)[c];
        return;
    }
    const int oy = (tok - 1) / pw, ox = (tok - 1) - oy * pw;
    const float fy = ((float)oy + 0.5f) * inv_sh - 0.5f, fx = ((float)ox + 0.5f) * inv_sw - 0.5f;
    const float fly = floorf(fy), flx = floorf(fx);
    const int iy = (int)fly, ix = (int)flx;
    float cy[4], cx[4];
    cubic_coeffs(fy - fly, cy);
    cubic_coeffs(fx - flx, cx);
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int ky = 0; ky < 4; ++ky) {
        const int yy = min(max(iy - 1 + ky, 0), sq - 1);
        float4 row = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int kx = 0; kx < 4; ++kx) {
            const int xx = min(max(ix - 1 + kx, 0), sq - 1);
            const float4 v = ((const float4*)(pos + (long)(1 + yy * sq + xx) * dim))[c];
            row.x += v.x * cx[kx]; row.y += v.y * cx[kx]; row.z += v.z * cx[kx]; row.w += v.w * cx[kx];
        }
        acc.x += row.x * cy[ky]; acc.y += row.y * cy[ky]; acc.z += row.z * cy[ky]; acc.w += row.w * cy[ky];
    }
    ((float4*)(out + (long)tok * dim))[c] = acc;
}

}  // namespace

extern "C" int ada_pos_embed_resize(const float* pos, int32_t sq, int32_t dim, int32_t ph, int32_t pw, double scale_h, double scale_w, float* out,
                                    void* stream) {
    ADA_REQUIRE(pos && out && sq > 0 && ph > 0 && pw > 0 && dim > 0 && dim % 4 == 0 && scale_h > 0 && scale_w > 0, ADA_EINVAL, "ada_pos_embed_resize: bad arguments");
    const long n = (long)(1 + ph * pw) * (dim / 4);
    hipLaunchKernelGGL(pos_embed_resize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, pos, sq, dim, ph, pw,
                       (float)(1.0 / scale_h), (float)(1.0 / scale_w), out);
    return ada_check_launch("ada_pos_embed_resize");
}

static std::atomic<int> g_ln_lpr{[]() { const char* e = getenv("ADA_LN_LPR"); return e ? atoi(e) : 0; }()};   // experiment switch: 64 = always one wave per row

extern "C" int ada_layernorm_ex(const ada_layernorm_args* a, void* stream) {
    ADA_REQUIRE(a != nullptr, ADA_EINVAL, "ada_layernorm: null args");
    const int rows_out = a->rows_out, dim = a->dim;
    ADA_REQUIRE(a->in && ((a->weight && a->bias) || a->identity), ADA_EINVAL, "ada_layernorm_fwd: null pointer");
    ADA_REQUIRE(a->relu >= 0 && a->relu <= 2 && (a->identity == 0 || a->identity == 1) && !(a->identity && a->out2_op), ADA_EINVAL, "ada_layernorm: relu in 0..2, identity in 0..1 (no second output)");
    ADA_REQUIRE(a->out_op || a->out_f32 || a->out2_op, ADA_EINVAL, "ada_layernorm_fwd: no output buffer");
    ADA_REQUIRE(rows_out > 0 && dim > 0, ADA_EINVAL, "ada_layernorm_fwd: bad shape rows=%d dim=%d", rows_out, dim);
    ADA_REQUIRE(dim % 4 == 0 && dim <= LN_MAX_CHUNKS * 256, ADA_EUNSUPPORTED, "ada_layernorm_fwd: dim=%d must be a multiple of 4 and <= 1536", dim);
    ADA_REQUIRE(a->ld_in % 4 == 0 && (!a->out_f32 || a->ld_f32 % 4 == 0) && (!a->out_op || a->ld_op % 4 == 0), ADA_EINVAL, "ada_layernorm_fwd: leading dims must be multiples of 4");
    ADA_REQUIRE(a->group_in == 0 || (a->skip >= 0 && a->skip < a->group_in), ADA_EINVAL, "ada_layernorm_fwd: bad group/skip");
    ADA_REQUIRE(a->map_op == ADA_MAP_PLAIN || a->map_op == ADA_MAP_PAD, ADA_EUNSUPPORTED, "ada_layernorm_fwd: map must be PLAIN or PAD");
    if (a->out_op && a->map_op == ADA_MAP_PAD) ADA_REQUIRE(a->map_h > 0 && a->map_w > 0 && rows_out % (a->map_h * a->map_w) == 0, ADA_EINVAL, "ada_layernorm_fwd: bad PAD grid");
    ADA_REQUIRE((long)rows_out < (1L << 24), ADA_EUNSUPPORTED, "ada_layernorm_fwd: too many rows");
    LnArgs p;
    p.in = a->in; p.ld_in = a->ld_in; p.rows_out = rows_out; p.dim = dim; p.group_in = a->group_in; p.skip = a->skip;
    p.dGroupOut = make_fastdiv(a->group_in > 0 ? a->group_in - a->skip : 1);
    p.weight = a->weight; p.bias = a->bias; p.eps = a->eps;
    p.out_op = (op_t*)a->out_op; p.ld_op = a->ld_op; p.map_op = a->map_op; p.map_h = a->map_h; p.map_w = a->map_w;
    p.dMapW = make_fastdiv(a->map_w > 0 ? a->map_w : 1);
    p.dMapHW = make_fastdiv(a->map_h > 0 && a->map_w > 0 ? a->map_h * a->map_w : 1);
    p.relu = a->relu; p.out_f32 = a->out_f32; p.ld_f32 = a->ld_f32;
    const int seg_abs = a->split_seg < 0 ? -a->split_seg : a->split_seg, seg2_abs = a->split_seg2 < 0 ? -a->split_seg2 : a->split_seg2;   // < 0: the [hi | lo8 | hi8] form
    ADA_REQUIRE(a->split_seg == 0 || (a->out_op && seg_abs >= dim && seg_abs % 4 == 0 && a->ld_op >= 2L * seg_abs), ADA_EINVAL, "ada_layernorm_fwd: bad split_seg=%d for dim=%d ld_op=%ld", a->split_seg, dim, (long)a->ld_op);
    p.split_seg = a->split_seg;
    p.weight2 = a->weight2; p.bias2 = a->bias2; p.out2 = (op_t*)a->out2_op; p.ld2 = a->ld2_op;
    p.out2_group = a->out2_group > 0 ? a->out2_group : 1; p.out2_skip = a->out2_group > 0 ? a->out2_skip : 0; p.split_seg2 = a->split_seg2;
    p.dOut2Group = make_fastdiv((uint32_t)p.out2_group);
    if (a->out2_op) {
        ADA_REQUIRE(a->weight2 && a->bias2 && a->ld2_op % 4 == 0 && a->group_in == 0 && a->unshuffle_s == 0, ADA_EINVAL, "ada_layernorm: the second output needs its own gain / bias, ld2 %% 4 == 0 and plainly ordered input rows");
        ADA_REQUIRE(a->out2_group == 0 || (a->out2_skip >= 0 && a->out2_skip < a->out2_group && rows_out % a->out2_group == 0), ADA_EINVAL, "ada_layernorm: bad out2 group/skip");
        ADA_REQUIRE(a->split_seg2 == 0 || (seg2_abs >= dim && seg2_abs % 4 == 0 && a->ld2_op >= 2L * seg2_abs), ADA_EINVAL, "ada_layernorm: bad split_seg2");
    }
    p.identity = a->identity;
    p.unshuffle_s = a->unshuffle_s; p.dUnS = make_fastdiv(a->unshuffle_s > 0 ? a->unshuffle_s : 1); p.tap_bias = a->tap_bias;
    if (a->unshuffle_s != 0) {
        ADA_REQUIRE(a->unshuffle_s > 0 && a->unshuffle_s <= 4 && a->group_in == 0 && a->map_h > 0 && a->map_w > 0 && a->map_h % a->unshuffle_s == 0 &&
                    a->map_w % a->unshuffle_s == 0 && rows_out % (a->map_h * a->map_w) == 0 && a->ld_in >= (long)a->unshuffle_s * a->unshuffle_s * dim, ADA_EINVAL,
                    "ada_layernorm: sub-pixel input needs s in 1..4, a fine grid map_h x map_w divisible by s and ld_in >= s*s*dim");
    } else {
        ADA_REQUIRE(!a->tap_bias, ADA_EINVAL, "ada_layernorm: tap_bias without unshuffle_s");
    }
    if (dim <= 512 && g_ln_lpr.load(std::memory_order_relaxed) != 64)     // a quarter wave per row (the kernel's comment); ADA_LN_LPR=64 keeps one wave per row (A/B)
        hipLaunchKernelGGL(layernorm_kernel<16>, dim3((rows_out + 15) / 16), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(layernorm_kernel<64>, dim3((rows_out + 3) / 4), dim3(256), 0, (hipStream_t)stream, p);
    return ada_check_launch("ada_layernorm_fwd");
}

extern "C" int ada_layernorm_fwd(const float* in, int64_t ld_in, int32_t rows_out, int32_t dim, int32_t group_in, int32_t skip,
                                 const float* weight, const float* bias, float eps, void* out_op, int64_t ld_op, int32_t map_op,
                                 int32_t map_h, int32_t map_w, int32_t relu, float* out_f32, int64_t ld_f32, int32_t split_seg, void* stream) {
    ada_layernorm_args a = {};
    a.in = in; a.ld_in = ld_in; a.rows_out = rows_out; a.dim = dim; a.group_in = group_in; a.skip = skip;
    a.weight = weight; a.bias = bias; a.eps = eps;
    a.out_op = out_op; a.ld_op = ld_op; a.map_op = map_op; a.map_h = map_h; a.map_w = map_w; a.relu = relu;
    a.out_f32 = out_f32; a.ld_f32 = ld_f32; a.split_seg = split_seg;
    return ada_layernorm_ex(&a, stream);
}

extern "C" int ada_tapsum_resize_fwd(const void* in, int32_t in_dtype, int64_t ld_in, int32_t batch, int32_t hi, int32_t wi, int32_t ho, int32_t wo, int32_t channels,
                                     const float* bias, float* out, int64_t ld_out, void* stream) {
    ADA_REQUIRE(in_dtype == ADA_DT_F32 || in_dtype == ADA_OP_DTYPE, ADA_EINVAL, "ada_tapsum_resize_fwd: in_dtype must be fp32 or the library's operand type");
    const bool f32in = in_dtype == ADA_DT_F32;
    ADA_REQUIRE(in && out, ADA_EINVAL, "ada_tapsum_resize_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0, ADA_EINVAL, "ada_tapsum_resize_fwd: bad shape");
    ADA_REQUIRE(channels == 32 || channels == 64 || channels == 128, ADA_EUNSUPPORTED, "ada_tapsum_resize_fwd: %d channels (supported: 32, 64, 128)", channels);
    ADA_REQUIRE(ld_in >= 9L * channels && ld_in % 8 == 0 && ((uintptr_t)in % 16) == 0, ADA_EINVAL, "ada_tapsum_resize_fwd: in must be 16-byte aligned with ld_in >= 9 * channels, ld_in %% 8 == 0");
    ADA_REQUIRE(ld_out >= channels && ld_out % 4 == 0 && ((uintptr_t)out % 16) == 0 && (!bias || ((uintptr_t)bias % 16) == 0), ADA_EINVAL, "ada_tapsum_resize_fwd: out / bias alignment");
    ADA_REQUIRE(ho <= 65535 * TS_TH && batch <= 65535, ADA_EUNSUPPORTED, "ada_tapsum_resize_fwd: grid limits");
    TapSumArgs p;
    p.in = in; p.ld_in = ld_in; p.batch = batch; p.hi = hi; p.wi = wi; p.ho = ho; p.wo = wo; p.C = channels;
    p.sy = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.0f;
    p.sx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.0f;
    p.bias = bias; p.out = out; p.ld_out = ld_out;
    // patch extent under a tile's halo: the same fp32 expressions the kernel evaluates, maximised over the tiles
    auto extent = [](float sc, int n_in, int n_out, int tile) {
        int best = 1;
        for (int t0 = 0; t0 < n_out; t0 += tile) {
            const int first = t0 > 0 ? t0 - 1 : 0;
            const int last = t0 + tile < n_out - 1 ? t0 + tile : n_out - 1;
            int hi_tap = (int)(sc * (float)last) + 1;
            if (hi_tap > n_in - 1) hi_tap = n_in - 1;
            const int e = hi_tap - (int)(sc * (float)first) + 1;
            if (e > best) best = e;
        }
        return best;
    };
    p.ph = extent(p.sy, hi, ho, TS_TH);
    p.pw = extent(p.sx, wi, wo, TS_TW);
    const int es = f32in ? 4 : 2;
    const int ppp = 256 / (channels * es / 16);
    const long tapbytes = (((long)p.ph * p.pw + ppp - 1) / ppp) * ppp * channels * es;
    ADA_REQUIRE(3 * tapbytes <= 150 * 1024, ADA_EUNSUPPORTED, "ada_tapsum_resize_fwd: the source patch of a tile (%d x %d pixels) does not fit LDS -- an up-sampling is expected", p.ph, p.pw);
    static std::once_flag once;
    std::call_once(once, []() {
        for (const void* k : {(const void*)tapsum_resize_kernel<32, op_t>, (const void*)tapsum_resize_kernel<64, op_t>, (const void*)tapsum_resize_kernel<128, op_t>,
                              (const void*)tapsum_resize_kernel<32, float>, (const void*)tapsum_resize_kernel<64, float>, (const void*)tapsum_resize_kernel<128, float>})
            if (hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) (void)hipGetLastError();
    });
    const dim3 grid((unsigned)((wo + TS_TW - 1) / TS_TW), (unsigned)((ho + TS_TH - 1) / TS_TH), (unsigned)batch);
#define ADA_TS_LAUNCH(CH, T) hipLaunchKernelGGL((tapsum_resize_kernel<CH, T>), grid, dim3(256), (size_t)(3 * tapbytes), (hipStream_t)stream, p)
    if (f32in) {
        if (channels == 128) ADA_TS_LAUNCH(128, float);
        else if (channels == 64) ADA_TS_LAUNCH(64, float);
        else ADA_TS_LAUNCH(32, float);
    } else {
        if (channels == 128) ADA_TS_LAUNCH(128, op_t);
        else if (channels == 64) ADA_TS_LAUNCH(64, op_t);
        else ADA_TS_LAUNCH(32, op_t);
    }
#undef ADA_TS_LAUNCH
    return ada_check_launch("ada_tapsum_resize_fwd");
}

extern "C" int ada_patchify(const float* x, const float* guide, int32_t batch, int32_t cg, int32_t height, int32_t width,
                            const float* mean, const float* inv_std, void* out, int64_t ld, int32_t split, void* stream) {
    ADA_REQUIRE(x && out, ADA_EINVAL, "ada_patchify: null pointer");
    ADA_REQUIRE(cg == 0 || guide, ADA_EINVAL, "ada_patchify: guide channels without guide tensor");
    ADA_REQUIRE(batch > 0 && cg >= 0 && height > 0 && width > 0, ADA_EINVAL, "ada_patchify: bad shape");
    ADA_REQUIRE(height % 14 == 0 && width % 14 == 0, ADA_EINVAL, "ada_patchify: %dx%d is not a multiple of the 14-pixel patch", height, width);
    ADA_REQUIRE(ld % 2 == 0 && ld >= (3 + cg) * 196, ADA_EINVAL, "ada_patchify: ld=%ld too small", (long)ld);
    ADA_REQUIRE(split == 0 || split == 1, ADA_EINVAL, "ada_patchify: split must be 0 or 1");
    ADA_REQUIRE(!split || (ld % 4 == 0 && ld / 2 >= (3 + cg) * 196), ADA_EINVAL, "ada_patchify: split needs ld = 2 * segment, segment >= (3+cg)*196");
    ADA_REQUIRE((mean == nullptr) == (inv_std == nullptr), ADA_EINVAL, "ada_patchify: mean and inv_std go together");
    float3 mu = make_float3(0.f, 0.f, 0.f), is = make_float3(1.f, 1.f, 1.f);
    if (mean) {  // host pointers: three floats each
        mu = make_float3(mean[0], mean[1], mean[2]);
        is = make_float3(inv_std[0], inv_std[1], inv_std[2]);
    }
    const int ph = height / 14, pw = width / 14;
    hipLaunchKernelGGL(patchify_kernel, dim3(batch * ph * pw), dim3(256), 0, (hipStream_t)stream, x, guide, cg, height, width, ph, pw,
                       mu, is, mean ? 1 : 0, (op_t*)out, (long)ld, split ? (int)(ld / 2) : 0);
    return ada_check_launch("ada_patchify");
}

extern "C" int ada_write_cls(float* tokens, int32_t batch, int32_t n_tokens, int32_t dim, const float* cls, const float* pos0, void* stream) {
    ADA_REQUIRE(tokens && cls && pos0, ADA_EINVAL, "ada_write_cls: null pointer");
    ADA_REQUIRE(batch > 0 && n_tokens > 0 && dim > 0, ADA_EINVAL, "ada_write_cls: bad shape");
    const int total = batch * dim;
    hipLaunchKernelGGL(write_cls_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, tokens, batch, n_tokens, dim, cls, pos0);
    return ada_check_launch("ada_write_cls");
}

extern "C" int ada_bilinear_fwd(const float* in, int64_t ld_in, int32_t batch, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                                int32_t channels, const float* add, int64_t ld_add, float* out_f32, int64_t ld_f32, void* out_op,
                                int64_t ld_op, int32_t map_op, int32_t relu, int32_t split_seg, void* stream) {
    ADA_REQUIRE(in && (out_f32 || out_op), ADA_EINVAL, "ada_bilinear_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0 && channels > 0, ADA_EINVAL, "ada_bilinear_fwd: bad shape");
    ADA_REQUIRE(channels % 4 == 0 && ld_in % 4 == 0 && (!add || ld_add % 4 == 0) && (!out_f32 || ld_f32 % 4 == 0) && (!out_op || ld_op % 4 == 0),
                ADA_EINVAL, "ada_bilinear_fwd: channels and leading dims must be multiples of 4");
    ADA_REQUIRE(map_op == ADA_MAP_PLAIN || map_op == ADA_MAP_PAD, ADA_EUNSUPPORTED, "ada_bilinear_fwd: map must be PLAIN or PAD");
    ADA_REQUIRE((long)batch * ho * wo < (1L << 24), ADA_EUNSUPPORTED, "ada_bilinear_fwd: more than 2^24 output pixels");
    BilinearArgs p;
    p.in = in; p.ld_in = ld_in; p.batch = batch; p.hi = hi; p.wi = wi; p.ho = ho; p.wo = wo; p.c4 = channels / 4;
    p.sy = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.0f;
    p.sx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.0f;
    p.add = add; p.ld_add = ld_add; p.out_f32 = out_f32; p.ld_f32 = ld_f32; p.out_op = (op_t*)out_op; p.ld_op = ld_op;
    p.map_op = map_op; p.relu = relu;
    const int seg_abs = split_seg < 0 ? -split_seg : split_seg;   // < 0: the [hi | lo8 | hi8] form
    ADA_REQUIRE(split_seg == 0 || (out_op && seg_abs >= channels && seg_abs % 4 == 0 && ld_op >= 2L * seg_abs), ADA_EINVAL, "ada_bilinear_fwd: bad split_seg=%d for %d channels, ld_op=%ld", split_seg, channels, (long)ld_op);
    p.split_seg = split_seg;
    p.dC4 = make_fastdiv(p.c4); p.dWo = make_fastdiv(wo); p.dHo = make_fastdiv(ho);
    ADA_REQUIRE(ho <= 65535 && batch <= 65535, ADA_EUNSUPPORTED, "ada_bilinear_fwd: ho / batch exceed the grid limits");
    // LDS-tiled path: up-sampling, channels in chunks of 128, source patch of an 8 x 16 output tile within 64 KiB
    // exact patch extent: the same fp32 expressions the kernel evaluates, maximised over the tile rows / columns
    auto extent = [](float sc, int n_in, int n_out, int tile) {
        int best = 1;
        for (int t0 = 0; t0 < n_out; t0 += tile) {
            const int last = t0 + tile - 1 < n_out - 1 ? t0 + tile - 1 : n_out - 1;
            int hi_tap = (int)(sc * (float)last) + 1;
            if (hi_tap > n_in - 1) hi_tap = n_in - 1;
            const int e = hi_tap - (int)(sc * (float)t0) + 1;
            if (e > best) best = e;
        }
        return best;
    };
    const int ph = extent(p.sy, hi, ho, BT_TH), pw = extent(p.sx, wi, wo, BT_TW);
    const int nchunk = channels / (4 * BT_CG);
    const bool tiled = channels % (4 * BT_CG) == 0 && p.sy <= 1.0f && p.sx <= 1.0f && ho >= BT_TH && wo >= BT_TW && ph * pw * 512 <= 65536 &&
                       (long)batch * nchunk <= 65535 && !getenv("ADA_BILINEAR_SIMPLE");
    if (tiled) {
        const int smem = ((ph * pw + 7) / 8) * 8 * 512;
        static bool attr_done = false;
        if (!attr_done) {
            if (hipFuncSetAttribute((const void*)bilinear_tiled_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 65536) != hipSuccess) (void)hipGetLastError();
            attr_done = true;
        }
        hipLaunchKernelGGL(bilinear_tiled_kernel, dim3((unsigned)((wo + BT_TW - 1) / BT_TW), (unsigned)((ho + BT_TH - 1) / BT_TH), (unsigned)(batch * nchunk)),
                           dim3(256), smem, (hipStream_t)stream, p, ph, pw, nchunk);
        return ada_check_launch("ada_bilinear_fwd");
    }
    hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)((wo * p.c4 + 255) / 256), (unsigned)ho, (unsigned)batch), dim3(256), 0,
                       (hipStream_t)stream, p);
    return ada_check_launch("ada_bilinear_fwd");
}
