// Depth evaluation on the device: one pass over (prediction, ground truth, valid mask) produces every
// masked per-image sum that the reference's metrics and its least-squares alignment need (src/util/metric.py:37-160,
// src/util/alignment.py:7-54).  The reference evaluates on the host in numpy/torch, one full-image temporary per metric;
// here the maps stay in HBM and are read once.  Sums are accumulated in fp64 (a 518x518 map has 2.7e5 terms).
#include "ada_common.h"

namespace {

constexpr int NSUM = ADA_EVAL_NSUM;

struct EvalArgs {
    const float* pred;
    const float* gt;
    const unsigned char* mask;   // may be null: every pixel valid
    long n;                      // pixels per image
    const float* scale_shift;    // [B][2] or null (identity)
    float clip_lo, clip_hi;      // applied after the affine map when clip_lo < clip_hi
    double* out;                 // [B][NSUM], zeroed by the caller side of the C entry point
};

// grid: (blocks per image, batch), 256 threads; per-thread fp64 partial sums, wave shuffle tree, LDS across the 4 waves,
// one fp64 atomic per sum per workgroup
__global__ __launch_bounds__(256) void eval_kernel(EvalArgs a) {
    __shared__ double part[4][NSUM];
    const int b = blockIdx.y;
    const float* pred = a.pred + (long)b * a.n;
    const float* gt = a.gt + (long)b * a.n;
    const unsigned char* mask = a.mask ? a.mask + (long)b * a.n : nullptr;
    float sc = 1.0f, sh = 0.0f;
    if (a.scale_shift) { sc = a.scale_shift[2 * b]; sh = a.scale_shift[2 * b + 1]; }
    const bool clip = a.clip_lo < a.clip_hi;
    double s[NSUM];
#pragma unroll
    for (int i = 0; i < NSUM; ++i) s[i] = 0.0;
    constexpr float T1 = 1.25f, T2 = 1.25f * 1.25f, T3 = 1.25f * 1.25f * 1.25f;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < a.n; i += (long)gridDim.x * 256) {
        if (mask && !mask[i]) continue;
        float p = pred[i] * sc + sh;
        if (clip) p = __builtin_fminf(__builtin_fmaxf(p, a.clip_lo), a.clip_hi);
        const float g = gt[i];
        const float d = p - g;
        const float lp = __logf(p), lg = __logf(g);
        const float dl = lp - lg;
        const float ratio = __builtin_fmaxf(p / g, g / p);
        const float di = 1.0f / p - 1.0f / g;
        s[ADA_EVAL_N] += 1.0;
        s[ADA_EVAL_SUM_P] += (double)p;
        s[ADA_EVAL_SUM_G] += (double)g;
        s[ADA_EVAL_SUM_PP] += (double)p * (double)p;
        s[ADA_EVAL_SUM_PG] += (double)p * (double)g;
        s[ADA_EVAL_ABS_REL] += (double)(__builtin_fabsf(d) / g);
        s[ADA_EVAL_SQ_REL] += (double)(d * d / g);
        s[ADA_EVAL_SQ] += (double)(d * d);
        s[ADA_EVAL_LOG_SQ] += (double)(dl * dl);
        s[ADA_EVAL_LOG] += (double)dl;
        s[ADA_EVAL_LOG10_ABS] += (double)(__builtin_fabsf(dl) * 0.43429448190325176f);
        s[ADA_EVAL_D1] += ratio < T1 ? 1.0 : 0.0;
        s[ADA_EVAL_D2] += ratio < T2 ? 1.0 : 0.0;
        s[ADA_EVAL_D3] += ratio < T3 ? 1.0 : 0.0;
        s[ADA_EVAL_INV_SQ] += (double)(di * di);
    }
#pragma unroll
    for (int i = 0; i < NSUM; ++i) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) s[i] += __shfl_xor(s[i], o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int i = 0; i < NSUM; ++i) part[w][i] = s[i];
    }
    __syncthreads();
    if (threadIdx.x < NSUM) {
        const double t = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
        unsafeAtomicAdd(a.out + (long)b * NSUM + threadIdx.x, t);
    }
}

__global__ void zero_kernel(double* p, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) p[i] = 0.0;
}

}  // namespace

extern "C" int ada_depth_eval_fwd(const float* pred, const float* gt, const uint8_t* mask, int32_t batch, int64_t n_per_image,
                                  const float* scale_shift, float clip_lo, float clip_hi, double* sums, void* stream) {
    ADA_REQUIRE(pred && gt && sums, ADA_EINVAL, "ada_depth_eval_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && n_per_image > 0, ADA_EINVAL, "ada_depth_eval_fwd: bad shape batch=%d n=%ld", batch, (long)n_per_image);
    ADA_REQUIRE(batch <= 65535, ADA_EUNSUPPORTED, "ada_depth_eval_fwd: batch=%d exceeds 65535", batch);
    EvalArgs a;
    a.pred = pred; a.gt = gt; a.mask = mask; a.n = n_per_image; a.scale_shift = scale_shift;
    a.clip_lo = clip_lo; a.clip_hi = clip_hi; a.out = sums;
    const int total = batch * NSUM;
    hipLaunchKernelGGL(zero_kernel, dim3((total + 255) / 256), dim3(256), 0, (hipStream_t)stream, sums, total);
    long blocks = (n_per_image + 256 * 8 - 1) / (256 * 8);   // ~8 pixels per thread
    if (blocks > 512) blocks = 512;
    hipLaunchKernelGGL(eval_kernel, dim3((unsigned)blocks, (unsigned)batch), dim3(256), 0, (hipStream_t)stream, a);
    return ada_check_launch("ada_depth_eval_fwd");
}
