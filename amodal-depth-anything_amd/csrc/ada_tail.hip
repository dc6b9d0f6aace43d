// Fused DPT tail (see include/ada_hip.h: ada_dpt_tail_fwd):
//     out = act( sum_n relu( conv3x3( bilinear_ac(in) )[.., n] + bias[n] ) * tail_w[n] + tail_b )
// i.e. reference DA2/dpt.py:194-195 -- F.interpolate(out, (14 ph, 14 pw), bilinear, align_corners=True) followed by
// scratch.output_conv2 = Conv2d(C, 32, 3, padding 1) -> ReLU -> Conv2d(32, 1, 1) -> Sigmoid / ReLU / Identity -- in ONE kernel.
//
// The two-kernel path materialises the up-sampled map ("fin": [B, H+2, W+2, C] operand-typed, 2.2 GB at ViT-L bs=32), writes it
// once and re-reads it nine times through L2 from a 32-column GEMM whose A tile is all traffic and no reuse (0.80 ms + 1.31 ms).
// Here a workgroup owns an 8 x 32 block of output pixels:
//   1. the source patch of `in` (fp32 NHWC rows) under the block's 10 x 34 halo is copied once into LDS by LDS-DMA (64 channels per pass);
//   2. the halo tile is interpolated from LDS into LDS, operand-typed, zero outside the image (= the convolution's padding), each pixel a
//      128-byte row whose 16-byte chunks are XOR-swizzled so that the fragment reads below are bank-conflict free;
//   3. the 3 x 3 convolution runs as 9 taps x 2 k-steps of v_mfma_f32_16x16x32 straight from the halo tile (a tap is a constant row /
//      column shift of the fragment address) against the weights of this channel pass, staged once per pass in LDS;
//   4. bias, ReLU, the 32 -> 1 projection (16-lane xor-shuffle reduction), activation, one fp32 store per pixel.
// Nothing but `in`, the 73 KB of weights and the output touches HBM / L2.
#include <mutex>
#include <stdlib.h>
#include "ada_common.h"

namespace {

constexpr int T_TH = 8, T_TW = 32;                 // output pixels per workgroup
constexpr int T_HH = T_TH + 2, T_HW = T_TW + 2;    // halo
constexpr int T_NPIX = T_HH * T_HW;                // 340 halo pixels
constexpr int T_CH = 64;                           // channels per pass
constexpr int T_PIXB = T_CH * 2;                   // bytes per halo pixel (128)
constexpr int T_N = 32;                            // output channels of the 3x3 convolution
constexpr int T_WROW = 9 * T_PIXB;                 // bytes per weight row in LDS (9 taps x 64 channels)
constexpr int T_HALO_BYTES = T_NPIX * T_PIXB;      // 43520
constexpr int T_W_BYTES = T_N * T_WROW;            // 36864

struct TailArgs {
    const float* in;
    long ld_in;
    int batch, hi, wi, ho, wo, passes;   // passes = padded channels / 64
    float sy, sx;
    const op_t* w;                       // [32, 9 * cp] tap-major
    int cp;
    const float* bias;
    const float* tail_w;
    float tail_b;
    int tail_act;
    float* out;
    int pw_max;                          // patch row stride bound used for the LDS carve (host-computed)
    int ablate;                          // timing experiments (ADA_TAIL_ABLATE): 1 no staging, 2 no interpolation, 4 no convolution
};

__global__ __launch_bounds__(256, 1) void dpt_tail_kernel(TailArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const halo = smem;
    char* const wbuf = smem + T_HALO_BYTES;
    char* const src = smem + T_HALO_BYTES + T_W_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, q4 = lane >> 4;
    const int tx0 = blockIdx.x * T_TW, ty0 = blockIdx.y * T_TH, b = blockIdx.z;

    // source patch under the halo (clamped to the image): rows py0 .. py1, columns px0 .. px1
    const int oy0 = ty0 > 0 ? ty0 - 1 : 0, oy1 = min(ty0 + T_TH, p.ho - 1);
    const int ox0 = tx0 > 0 ? tx0 - 1 : 0, ox1 = min(tx0 + T_TW, p.wo - 1);
    const int py0 = (int)(p.sy * (float)oy0), px0 = (int)(p.sx * (float)ox0);
    const int py1 = min((int)(p.sy * (float)oy1) + 1, p.hi - 1), px1 = min((int)(p.sx * (float)ox1) + 1, p.wi - 1);
    const int ph = py1 - py0 + 1, pw = px1 - px0 + 1;
    const int npatch = ph * pw;
    const long img = (long)b * p.hi;

    f32x4 acc[4][2];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    // fragment addresses: A rows are halo pixels (2 * wave + (f >> 1) + dy, 16 * (f & 1) + dx + l15); the 16-byte chunk of a pixel row is
    // XOR-ed with (column >> 1) & 7, so 16 consecutive pixels of one halo row hit 16 distinct bank slots for any starting column
    unsigned abase[2][3][2], bbase[2][2];
#pragma unroll
    for (int fx = 0; fx < 2; ++fx)
#pragma unroll
        for (int dx = 0; dx < 3; ++dx)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int hx = 16 * fx + dx + l15;
                abase[fx][dx][s] = (unsigned)((2 * wave * T_HW + hx) * T_PIXB + (((4 * s + q4) ^ ((hx >> 1) & 7)) * 16));
            }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const int n = 16 * j + l15;
            bbase[j][s] = (unsigned)(T_HALO_BYTES + n * T_WROW + (((4 * s + q4) ^ ((n >> 1) & 7)) * 16));
        }

    for (int pass = 0; pass < p.passes; ++pass) {
        // ---- 1. source patch -> LDS (16 pixels of 64 fp32 channels per round, lane-linear: 256 B per pixel) and weights -> LDS ----
        if (!(p.ablate & 1))
        for (int base = 0; base < npatch; base += 16) {
            int pp = base + (tid >> 4);
            if (pp >= npatch) pp = npatch - 1;
            const int r = pp / pw, c = pp - r * pw;
            const float* g = p.in + ((img + py0 + r) * p.wi + (px0 + c)) * p.ld_in + pass * T_CH + 4 * (tid & 15);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                             (__attribute__((address_space(3))) void*)(src + base * 256 + wave * 1024), 16, 0, 0);
        }
        if (!(p.ablate & 1))
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            const int q = tid + 256 * i;            // 2304 chunks of 16 B: (n, tap, chunk)
            const int n = q / 72, rem = q - n * 72, t = rem >> 3, c = rem & 7;
            const u32x4 v = *(const u32x4*)(p.w + (long)n * (9 * p.cp) + t * p.cp + pass * T_CH + 8 * c);
            *(u32x4*)(wbuf + n * T_WROW + t * T_PIXB + ((c ^ ((n >> 1) & 7)) * 16)) = v;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();

        // ---- 2. halo tile = bilinear(source patch), operand-typed, zero outside the image ----
        if (!(p.ablate & 2))
        for (int item = tid; item < T_NPIX * 8; item += 256) {
            const int hp = item >> 3, c = item & 7;
            const int hy = hp / T_HW, hx = hp - hy * T_HW;
            const int y = ty0 - 1 + hy, x = tx0 - 1 + hx;
            opx8 o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = (op_t)0.0f;
            if (y >= 0 && y < p.ho && x >= 0 && x < p.wo) {
                const float fy = p.sy * (float)y, fx = p.sx * (float)x;
                const int y0 = (int)fy, x0 = (int)fx;
                const int y1 = y0 + (y0 < p.hi - 1 ? 1 : 0), x1 = x0 + (x0 < p.wi - 1 ? 1 : 0);
                const float ly1 = fy - (float)y0, lx1 = fx - (float)x0;
                const float ly0 = 1.0f - ly1, lx0 = 1.0f - lx1;
                const char* t0 = src + ((y0 - py0) * pw - px0) * 256 + c * 32;
                const char* t1 = src + ((y1 - py0) * pw - px0) * 256 + c * 32;
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const float4 v00 = *(const float4*)(t0 + x0 * 256 + h * 16);
                    const float4 v01 = *(const float4*)(t0 + x1 * 256 + h * 16);
                    const float4 v10 = *(const float4*)(t1 + x0 * 256 + h * 16);
                    const float4 v11 = *(const float4*)(t1 + x1 * 256 + h * 16);
                    o[4 * h + 0] = to_op(ly0 * (lx0 * v00.x + lx1 * v01.x) + ly1 * (lx0 * v10.x + lx1 * v11.x));
                    o[4 * h + 1] = to_op(ly0 * (lx0 * v00.y + lx1 * v01.y) + ly1 * (lx0 * v10.y + lx1 * v11.y));
                    o[4 * h + 2] = to_op(ly0 * (lx0 * v00.z + lx1 * v01.z) + ly1 * (lx0 * v10.z + lx1 * v11.z));
                    o[4 * h + 3] = to_op(ly0 * (lx0 * v00.w + lx1 * v01.w) + ly1 * (lx0 * v10.w + lx1 * v11.w));
                }
            }
            *(opx8*)(halo + hp * T_PIXB + ((c ^ ((hx >> 1) & 7)) * 16)) = o;
        }
        __syncthreads();

        // ---- 3. 3x3 convolution from the halo tile: 9 taps x 2 k-steps of 32 channels ----
        if (!(p.ablate & 4))
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx)
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    opx8 af[4], bf[2];
#pragma unroll
                    for (int f = 0; f < 4; ++f) af[f] = *(const opx8*)(smem + abase[f & 1][dx][s] + ((f >> 1) + dy) * (T_HW * T_PIXB));
#pragma unroll
                    for (int j = 0; j < 2; ++j) bf[j] = *(const opx8*)(smem + bbase[j][s] + (dy * 3 + dx) * T_PIXB);
#pragma unroll
                    for (int f = 0; f < 4; ++f)
#pragma unroll
                        for (int j = 0; j < 2; ++j) acc[f][j] = mfma16(af[f], bf[j], acc[f][j]);
                }
        __syncthreads();   // every wave is done with the halo tile, the weights and (long since) the source patch of this pass
    }

    // ---- 4. bias, ReLU, 32 -> 1, activation.  D[4 * q4 + r][l15]: row = pixel inside the 16-pixel fragment, column = output channel ----
    const float b0 = p.bias[l15], b1 = p.bias[16 + l15], w0 = p.tail_w[l15], w1 = p.tail_w[16 + l15];
#pragma unroll
    for (int f = 0; f < 4; ++f) {
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            v[r] = __builtin_fmaxf(acc[f][0][r] + b0, 0.0f) * w0 + __builtin_fmaxf(acc[f][1][r] + b1, 0.0f) * w1;
#pragma unroll
            for (int o = 1; o < 16; o <<= 1) v[r] += __shfl_xor(v[r], o);
        }
        if (l15 == 0) {
            const int y = ty0 + 2 * wave + (f >> 1);
            if (y < p.ho) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int x = tx0 + 16 * (f & 1) + 4 * q4 + r;
                    if (x < p.wo) {
                        float d = v[r] + p.tail_b;
                        if (p.tail_act == ADA_ACT_SIGMOID) d = 1.0f / (1.0f + __expf(-d));
                        else if (p.tail_act == ADA_ACT_RELU) d = __builtin_fmaxf(d, 0.0f);
                        p.out[((long)b * p.ho + y) * p.wo + x] = d;
                    }
                }
            }
        }
    }
}

}  // namespace

extern "C" int ada_dpt_tail_fwd(const float* in, int64_t ld_in, int32_t batch, int32_t hi, int32_t wi, int32_t ho, int32_t wo, int32_t cp,
                                const void* w, const float* bias, const float* tail_w, float tail_b, int32_t tail_act, float* out,
                                void* stream) {
    ADA_REQUIRE(in && w && bias && tail_w && out, ADA_EINVAL, "ada_dpt_tail_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && batch <= 65535 && hi > 0 && wi > 0 && ho > 0 && wo > 0, ADA_EINVAL, "ada_dpt_tail_fwd: bad shape");
    ADA_REQUIRE(cp > 0 && cp % T_CH == 0 && ld_in >= cp && ld_in % 4 == 0, ADA_EUNSUPPORTED,
                "ada_dpt_tail_fwd: the padded channel count (%d) must be a multiple of 64 and fit the input row (ld_in=%ld)", cp, (long)ld_in);
    ADA_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)w % 16) == 0, ADA_EINVAL, "ada_dpt_tail_fwd: in / w must be 16-byte aligned");
    TailArgs p;
    p.in = in; p.ld_in = ld_in; p.batch = batch; p.hi = hi; p.wi = wi; p.ho = ho; p.wo = wo; p.passes = cp / T_CH;
    p.sy = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.0f;
    p.sx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.0f;
    p.w = (const op_t*)w; p.cp = cp; p.bias = bias; p.tail_w = tail_w; p.tail_b = tail_b; p.tail_act = tail_act; p.out = out;
    // LDS carve: the largest source patch a block can need
    const int ph_max = (int)(p.sy * (float)(T_TH + 1)) + 3, pw_max = (int)(p.sx * (float)(T_TW + 1)) + 3;
    const int npatch_max = ((ph_max * pw_max + 15) / 16) * 16;
    const size_t smem = (size_t)T_HALO_BYTES + T_W_BYTES + (size_t)npatch_max * 256;
    ADA_REQUIRE(smem <= 160 * 1024, ADA_EUNSUPPORTED, "ada_dpt_tail_fwd: source patch of %d x %d pixels does not fit in LDS (down-sampling is not supported)", ph_max, pw_max);
    p.pw_max = pw_max;
#ifdef ADA_TAIL_ABLATION   // timing experiments only (csrc/build.py --tag abl -D ADA_TAIL_ABLATION): never in the shipped library
    {
        const char* e = getenv("ADA_TAIL_ABLATE");
        p.ablate = e ? atoi(e) : 0;
    }
#else
    p.ablate = 0;
#endif
    static std::once_flag once;
    std::call_once(once, []() {
        if (hipFuncSetAttribute((const void*)dpt_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
    });
    hipLaunchKernelGGL(dpt_tail_kernel, dim3((wo + T_TW - 1) / T_TW, (ho + T_TH - 1) / T_TH, batch), dim3(256), smem, (hipStream_t)stream, p);
    return ada_check_launch("ada_dpt_tail_fwd");
}
