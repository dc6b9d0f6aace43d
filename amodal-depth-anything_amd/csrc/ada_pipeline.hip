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

}  // namespace

extern "C" int ada_minmax_fwd(const float* in, int32_t batch, int64_t n_per_image, float* minmax, void* stream) {
    ADA_REQUIRE(in && minmax, ADA_EINVAL, "ada_minmax_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && n_per_image > 0, ADA_EINVAL, "ada_minmax_fwd: bad shape");
    hipLaunchKernelGGL(minmax_kernel, dim3(batch), dim3(1024), 0, (hipStream_t)stream, in, (long)n_per_image, minmax);
    return ada_check_launch("ada_minmax_fwd");
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
