// Device-side glue of the two-model infer.py pipeline: per-image min/max of the base depth map,
// its normalisation to the +-1 "observation" the amodal network expects, and the final paste + border box-blur blend.
// The reference does all of this on the host in numpy/cv2 between the two forwards (infer.py:21-22,92,30-44), bouncing
// the depth map through PCIe twice; here it stays in HBM.  All three are single-pass byte movers.
#include <float.h>
#include "ada_common.h"

namespace {

// one workgroup per image: per-lane running min/max, wave shuffle reduce, 16 waves combined through LDS
__global__ __launch_bounds__(1024) void minmax_kernel(const float* __restrict__ in, long n_per_image, float* __restrict__ out) {
    __shared__ float smin[16], smax[16];
    const float* src = in + (long)blockIdx.x * n_per_image;
    float lo = FLT_MAX, hi = -FLT_MAX;
    for (long i = threadIdx.x; i < n_per_image; i += 1024) {
        const float v = src[i];
        lo = __builtin_fminf(lo, v);
        hi = __builtin_fmaxf(hi, v);
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        lo = __builtin_fminf(lo, __shfl_xor(lo, o));
        hi = __builtin_fmaxf(hi, __shfl_xor(hi, o));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { smin[w] = lo; smax[w] = hi; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < 16; ++i) { lo = __builtin_fminf(lo, smin[i]); hi = __builtin_fmaxf(hi, smax[i]); }
        out[2 * blockIdx.x] = lo;
        out[2 * blockIdx.x + 1] = hi;
    }
}

// Per-image sums of a sigmoid-head depth map s in (0, 1): (sum s, sum s (1 - s)) -- the two moments the engine's precision ladder needs.  The relative
// L1 of a sigmoid output against its fp32 reference is  mean|sigma'(z) dz| / mean s = mean(s (1 - s) |dz|) / mean(s): the ratio of the two sums is the
// factor by which the sigmoid compresses the head's logit error for THIS image (0.3-0.5 for maps centred in (0, 1), -> 1 as the map approaches 0).
// grid (chunks, batch): every workgroup sums a contiguous chunk in a fixed order (no atomics: bit-reproducible); the host adds the chunk sums.
// act (the final activation of the head, ADA_ACT_*): SIGMOID (sum s, sum s (1 - s)); RELU (sum out, number of positive outputs); NONE (sum |out|, number of outputs)
__global__ __launch_bounds__(1024) void depth_stats_kernel(const float* __restrict__ in, long n_per_image, int act, float* __restrict__ out) {
    __shared__ float s1[16], s2[16];
    const int chunks = gridDim.x, c = blockIdx.x, b = blockIdx.y;
    const long per = (n_per_image + chunks - 1) / chunks;
    const long lo = (long)c * per, hi = lo + per < n_per_image ? lo + per : n_per_image;
    const float* src = in + (long)b * n_per_image;
    float a = 0.0f, q = 0.0f;
    for (long i = lo + threadIdx.x; i < hi; i += 1024) {
        const float v = src[i];
        if (act == ADA_ACT_SIGMOID) {
            a += v;
            q += v * (1.0f - v);
        } else if (act == ADA_ACT_RELU) {
            a += v;
            q += v > 0.0f ? 1.0f : 0.0f;
        } else {
            a += __builtin_fabsf(v);
            q += 1.0f;
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) {
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s1[w] = a; s2[w] = q; }
    __syncthreads();
    if (threadIdx.x == 0) {
#pragma unroll
        for (int i = 1; i < 16; ++i) { a += s1[i]; q += s2[i]; }
        out[((long)b * chunks + c) * 2] = a;
        out[((long)b * chunks + c) * 2 + 1] = q;
    }
}

// Token diversity of one encoder tap: per image  sum_c Var_p(t[p, c])  and  sum_c E_p[t[p, c]^2]  over the image's patch tokens p (rows) and the feature
// columns c.  Their ratio is ~0.3-0.5 for images and ~0.02 for constant inputs (every patch token equal up to its position): there the head's operand
// rounding errors add coherently over positions and the single-precision head's relative L1 doubles -- the second trigger of the engine's precision
// ladder (the first, ada_depth_stats_fwd, only sees the output).  grid (column chunks of 64, batch); 256 threads = 8 column groups (8 columns = one
// 16-byte load) x 32 row groups; every workgroup writes (sum of the column variances, sum of the column mean squares) of its chunk: fixed order, no
// atomics.  (A first version with one 2-byte load per thread ran at 0.6 TB/s: 150 us per ViT-L bs=32 tap.)
__global__ __launch_bounds__(256) void token_diversity_kernel(const op_t* __restrict__ tap, long ld, int rows_per_image, int dim, float* __restrict__ out) {
    __shared__ float s1[32][65], s2[32][65];
    const int cg = threadIdx.x & 7, rg = threadIdx.x >> 3;
    const int col0 = blockIdx.x * 64 + cg * 8, b = blockIdx.y;
    const op_t* src = tap + (long)b * rows_per_image * ld + col0;
    float a[8], q[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = q[i] = 0.0f;
    if (col0 + 8 <= dim && (ld & 7) == 0) {
        for (int r = rg; r < rows_per_image; r += 32) {
            const opx8 v = *(const opx8*)(src + (long)r * ld);
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float f = (float)v[i]; a[i] += f; q[i] += f * f; }
        }
    } else {
        for (int r = rg; r < rows_per_image; r += 32)
#pragma unroll
            for (int i = 0; i < 8; ++i)
                if (col0 + i < dim) { const float f = (float)src[(long)r * ld + i]; a[i] += f; q[i] += f * f; }
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) { s1[rg][cg * 8 + i] = a[i]; s2[rg][cg * 8 + i] = q[i]; }
    __syncthreads();
    if (threadIdx.x < 64) {
        const int c = threadIdx.x, col = blockIdx.x * 64 + c;
        float sa = 0.0f, sq = 0.0f;
#pragma unroll
        for (int r = 0; r < 32; ++r) { sa += s1[r][c]; sq += s2[r][c]; }
        const float inv = 1.0f / (float)rows_per_image;
        const float mean = sa * inv, msq = sq * inv;
        float var = col < dim ? __builtin_fmaxf(msq - mean * mean, 0.0f) : 0.0f;
        float ms = col < dim ? msq : 0.0f;
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            var += __shfl_xor(var, o);
            ms += __shfl_xor(ms, o);
        }
        if (c == 0) {
            out[((long)b * gridDim.x + blockIdx.x) * 2] = var;
            out[((long)b * gridDim.x + blockIdx.x) * 2 + 1] = ms;
        }
    }
}

// norm = (d - min) / (max - min)   (reference infer.py:22);  obs = norm * 2 - 1   (infer.py:92)
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ in, const float* __restrict__ minmax, long n_per_image,
                                                        float* __restrict__ norm, float* __restrict__ obs) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n_per_image) return;
    const int b = blockIdx.y;
    const float lo = minmax[2 * b], hi = minmax[2 * b + 1];
    const float v = (in[(long)b * n_per_image + i] - lo) / (hi - lo);
    if (norm) norm[(long)b * n_per_image + i] = v;
    if (obs) obs[(long)b * n_per_image + i] = v * 2.0f - 1.0f;
}

ADA_DEV int reflect101(int i, int n) {  // cv2 BORDER_REFLECT_101 for a radius-1 filter
    if (i < 0) return -i;
    if (i >= n) return 2 * n - 2 - i;
    return i;
}

// blended = mask > 0 ? amodal : base;  border = 0 < boxsum3x3(mask) < 9 (zero padded);  out = border ? blur3x3(blended) : blended
// (reference infer.py:30-44; cv2.blur = normalised box filter with reflect-101 borders)
__global__ __launch_bounds__(256) void blend_kernel(const float* __restrict__ amodal, const float* __restrict__ base,
                                                    const float* __restrict__ mask, int H, int W, float* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, b = blockIdx.z;
    const long img = (long)b * H * W;
    auto blended = [&](int yy, int xx) {
        const long i = img + (long)yy * W + xx;
        return mask[i] > 0.0f ? amodal[i] : base[i];
    };
    float msum = 0.0f;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
        for (int dx = -1; dx <= 1; ++dx) {
            const int yy = y + dy, xx = x + dx;
            if (yy >= 0 && yy < H && xx >= 0 && xx < W) msum += mask[img + (long)yy * W + xx] > 0.0f ? 1.0f : 0.0f;
        }
    float v = blended(y, x);
    if (msum > 0.0f && msum < 9.0f) {
        float s = 0.0f;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) s += blended(reflect101(y + dy, H), reflect101(x + dx, W));
        v = s / 9.0f;
    }
    out[img + (long)y * W + x] = v;
}

// Overlap-blend of tile predictions into the full-resolution map (tiled inference for inputs larger than the network's 518 x 518).
// Tile t of image b covers rows oy[t] .. oy[t] + th - 1, columns ox[t] .. ox[t] + tw - 1 and carries the separable feather weight
// w(i, n) = min(i + 1, n - i, ramp) / ramp along each axis (1 in the tile interior, a linear ramp over the `ramp` outermost
// pixels), so two overlapping tiles cross-fade linearly and a pixel covered by a single tile keeps that tile's value:
//     out(b, y, x) = sum_t w_t(y, x) * tile_t(y - oy_t, x - ox_t) / sum_t w_t(y, x)
// One thread per output pixel, gathering from the (at most four) tiles that cover it.
__global__ __launch_bounds__(256) void tile_blend_kernel(const float* __restrict__ tiles, int T, int th, int tw, const int* __restrict__ oy,
                                                         const int* __restrict__ ox, int H, int W, int ramp, float* __restrict__ out) {
    const int x = blockIdx.x * 256 + threadIdx.x;
    if (x >= W) return;
    const int y = blockIdx.y, b = blockIdx.z;
    const float inv = 1.0f / (float)ramp;
    float acc = 0.0f, wsum = 0.0f;
    for (int t = 0; t < T; ++t) {
        const int ty = y - oy[t], tx = x - ox[t];
        if (ty < 0 || ty >= th || tx < 0 || tx >= tw) continue;
        const float wy = (float)min(min(ty + 1, th - ty), ramp) * inv;
        const float wx = (float)min(min(tx + 1, tw - tx), ramp) * inv;
        const float w = wy * wx;
        acc += w * tiles[(((long)b * T + t) * th + ty) * tw + tx];
        wsum += w;
    }
    out[((long)b * H + y) * W + x] = acc / wsum;
}

}  // namespace

extern "C" int ada_tile_blend_fwd(const float* tiles, int32_t batch, int32_t n_tiles, int32_t tile_h, int32_t tile_w, const int32_t* origin_y,
                                  const int32_t* origin_x, int32_t height, int32_t width, int32_t ramp, float* out, void* stream) {
    ADA_REQUIRE(tiles && origin_y && origin_x && out, ADA_EINVAL, "ada_tile_blend_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && batch <= 65535 && n_tiles > 0 && tile_h > 0 && tile_w > 0 && height >= tile_h && width >= tile_w && height <= 65535,
                ADA_EINVAL, "ada_tile_blend_fwd: bad shape B=%d T=%d tile=%dx%d image=%dx%d", batch, n_tiles, tile_h, tile_w, height, width);
    ADA_REQUIRE(ramp >= 1 && 2 * ramp <= tile_h && 2 * ramp <= tile_w, ADA_EINVAL, "ada_tile_blend_fwd: ramp=%d must be in [1, tile/2]", ramp);
    hipLaunchKernelGGL(tile_blend_kernel, dim3((width + 255) / 256, height, batch), dim3(256), 0, (hipStream_t)stream, tiles, n_tiles, tile_h,
                       tile_w, origin_y, origin_x, height, width, ramp, out);
    return ada_check_launch("ada_tile_blend_fwd");
}

extern "C" int ada_minmax_fwd(const float* in, int32_t batch, int64_t n_per_image, float* minmax, void* stream) {
    ADA_REQUIRE(in && minmax, ADA_EINVAL, "ada_minmax_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && n_per_image > 0, ADA_EINVAL, "ada_minmax_fwd: bad shape");
    hipLaunchKernelGGL(minmax_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, in, (long)n_per_image, minmax);
    return ada_check_launch("ada_minmax_fwd");
}

extern "C" int ada_depth_stats_fwd(const float* in, int32_t batch, int64_t n_per_image, int32_t chunks, int32_t act, float* sums, void* stream) {
    ADA_REQUIRE(in && sums, ADA_EINVAL, "ada_depth_stats_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && batch <= 65535 && n_per_image > 0 && chunks > 0 && chunks <= 1024 && act >= ADA_ACT_NONE && act <= ADA_ACT_RELU, ADA_EINVAL, "ada_depth_stats_fwd: bad shape (batch=%d chunks=%d)", batch, chunks);
    hipLaunchKernelGGL(depth_stats_kernel, dim3(chunks, batch), dim3(1024), 0, (hipStream_t)stream, in, (long)n_per_image, (int)act, sums);
    return ada_check_launch("ada_depth_stats_fwd");
}

extern "C" int ada_token_diversity_fwd(const void* tap, int64_t ld, int32_t batch, int32_t rows_per_image, int32_t dim, float* sums, void* stream) {
    ADA_REQUIRE(tap && sums, ADA_EINVAL, "ada_token_diversity_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && batch <= 65535 && rows_per_image > 0 && dim > 0 && ld >= dim && ((uintptr_t)tap % 16) == 0, ADA_EINVAL, "ada_token_diversity_fwd: bad shape (batch=%d rows=%d dim=%d ld=%ld)", batch, rows_per_image, dim, (long)ld);
    hipLaunchKernelGGL(token_diversity_kernel, dim3((dim + 63) / 64, batch), dim3(256), 0, (hipStream_t)stream, (const op_t*)tap, (long)ld, rows_per_image, dim, sums);
    return ada_check_launch("ada_token_diversity_fwd");
}

extern "C" int ada_normalize_fwd(const float* in, const float* minmax, int32_t batch, int64_t n_per_image, float* norm, float* obs,
                                 void* stream) {
    ADA_REQUIRE(in && minmax && (norm || obs), ADA_EINVAL, "ada_normalize_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && batch <= 65535 && n_per_image > 0, ADA_EINVAL, "ada_normalize_fwd: bad shape");
    hipLaunchKernelGGL(normalize_kernel, dim3((unsigned)((n_per_image + 255) / 256), batch), dim3(256), 0, (hipStream_t)stream, in, minmax,
                       (long)n_per_image, norm, obs);
    return ada_check_launch("ada_normalize_fwd");
}

extern "C" int ada_blend_fwd(const float* amodal, const float* base, const float* mask, int32_t batch, int32_t height, int32_t width,
                             float* out, void* stream) {
    ADA_REQUIRE(amodal && base && mask && out, ADA_EINVAL, "ada_blend_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && batch <= 65535 && height > 1 && height <= 65535 && width > 1, ADA_EINVAL, "ada_blend_fwd: bad shape");
    hipLaunchKernelGGL(blend_kernel, dim3((width + 255) / 256, height, batch), dim3(256), 0, (hipStream_t)stream, amodal, base, mask, height,
                       width, out);
    return ada_check_launch("ada_blend_fwd");
}
