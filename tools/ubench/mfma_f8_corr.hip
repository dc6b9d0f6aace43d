// Micro-benchmark + semantics check (yardstick, not product) for the fp8 correction terms of a split-precision product on gfx950:
//   x w  ~  x_hi w_hi (fp16 MFMA)  +  x_lo8 w_hi8 + x_hi8 w_lo8 (v_mfma_scale_f32_16x16x128_f8f6f4, twice the fp16 rate)
// Part A checks what the kernel relies on: (1) lane l supplies row / column l & 15 and the SAME (lane group, byte) -> k map for A and B, so any
// k order that A and B share contracts correctly; (2) an E8M0 scale byte e multiplies by 2^(e - 127), byte 0 of the scale register with op_sel 0;
// (3) A in e5m2 (cbsz 1) against B in e4m3 (blgp 0); (4) v_cvt_pk_bf8_f32 / v_cvt_pk_fp8_f32 round to nearest even and what they do above the range.
// Part B measures the sustained rate of the fp8 instruction beside the fp16 one under the power cap (registers only, 2 waves per SIMD).
//   hipcc --offload-arch=gfx950 -O3 -o mfma_f8_corr mfma_f8_corr.hip && ./mfma_f8_corr
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef int v8i __attribute__((ext_vector_type(8)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4v __attribute__((ext_vector_type(4)));

static float dec_e5m2(uint8_t b) {
    const int s = b >> 7, e = (b >> 2) & 31, m = b & 3;
    float v;
    if (e == 31) v = m ? NAN : INFINITY;
    else if (e == 0) v = std::ldexp((float)m, -16);
    else v = std::ldexp(1.0f + m / 4.0f, e - 15);
    return s ? -v : v;
}
static float dec_e4m3(uint8_t b) {
    const int s = b >> 7, e = (b >> 3) & 15, m = b & 7;
    float v;
    if (e == 15 && m == 7) v = NAN;
    else if (e == 0) v = std::ldexp((float)m, -9);
    else v = std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -v : v;
}
// nearest-even encoders by exhaustive search over the 256 codes (slow, exact)
template <typename D>
static uint8_t enc_nearest(float x, D dec) {
    int best = 0;
    double bd = 1e300;
    for (int c = 0; c < 256; ++c) {
        const float v = dec((uint8_t)c);
        if (!std::isfinite(v)) continue;
        const double d = std::fabs((double)v - (double)x);
        if (d < bd || (d == bd && !(c & 1) && (best & 1))) {
            bd = d;
            best = c;
        }
    }
    if (dec((uint8_t)best) == 0.0f) best = std::signbit(x) ? 0x80 : 0;   // underflow keeps the sign
    return (uint8_t)best;
}

__global__ void k_sem(const uint8_t* a, const uint8_t* b, float* c, int sa, int sb, int mode) {
    const int l = threadIdx.x, r16 = l & 15, g = l >> 4;
    v8i fa, fb;
    // mode 0: lane group g holds k = 32 g .. 32 g + 31; mode 1: k = 16 g .. + 15 and 64 + 16 g .. + 15 (what two 16-byte LDS chunks q4, 4 + q4 give)
    const uint8_t* pa = a + r16 * 128;
    const uint8_t* pb = b + r16 * 128;
    if (mode == 0) {
        fa = *(const v8i*)(pa + 32 * g);
        fb = *(const v8i*)(pb + 32 * g);
    } else {
        const int4 a0 = *(const int4*)(pa + 16 * g), a1 = *(const int4*)(pa + 64 + 16 * g);
        const int4 b0 = *(const int4*)(pb + 16 * g), b1 = *(const int4*)(pb + 64 + 16 * g);
        fa = v8i{a0.x, a0.y, a0.z, a0.w, a1.x, a1.y, a1.z, a1.w};
        fb = v8i{b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
    }
    f4v acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa, fb, acc, 1, 0, 0, sa, 0, sb);   // A e5m2, B e4m3
    for (int r = 0; r < 4; ++r) c[(4 * g + r) * 16 + r16] = acc[r];
}

__global__ void k_cvt(const float* x, uint8_t* bf8, uint8_t* fp8, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (2 * i + 1 >= n + 1) return;
    const int p = __builtin_amdgcn_cvt_pk_bf8_f32(x[2 * i], x[2 * i + 1], 0, false);
    const int q = __builtin_amdgcn_cvt_pk_fp8_f32(x[2 * i], x[2 * i + 1], 0, false);
    bf8[2 * i] = p & 255; bf8[2 * i + 1] = (p >> 8) & 255;
    fp8[2 * i] = q & 255; fp8[2 * i + 1] = (q >> 8) & 255;
}

// rate: a 128 x 64 wave tile (the product kernel's), fragments in registers
// MIX 0: fp16 16x16x32 only (2 sub-steps = 64 k per iteration); 1: fp8 16x16x128 only (128 k per iteration)
template <int MIX>
__global__ __launch_bounds__(512, 2) void k_rate(const v8i* __restrict__ src, float* __restrict__ out, int iters, int sa, int sb) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    v8i fa[8], fb[4];
#pragma unroll
    for (int i = 0; i < 8; ++i) fa[i] = src[(size_t)t * 12 + i];
#pragma unroll
    for (int j = 0; j < 4; ++j) fb[j] = src[(size_t)t * 12 + 8 + j];
    f4v acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f4v{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        if (MIX == 0) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int i = 0; i < 8; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int4 av = s ? int4{fa[i][4], fa[i][5], fa[i][6], fa[i][7]} : int4{fa[i][0], fa[i][1], fa[i][2], fa[i][3]};
                        const int4 bv = s ? int4{fb[j][4], fb[j][5], fb[j][6], fb[j][7]} : int4{fb[j][0], fb[j][1], fb[j][2], fb[j][3]};
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, av), __builtin_bit_cast(h8, bv), acc[i][j], 0, 0, 0);
                    }
        } else {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(fa[i], fb[j], acc[i][j], 1, 0, 0, sa, 0, sb);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
    out[t] = sum;
}

template <typename K>
static void run(const char* name, K kern, const v8i* src, float* out, int iters, double flop_per_iter_wave) {
    const int blocks = 256;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, src, out, iters, 127, 127);
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(512), 0, 0, src, out, iters, 127, 127);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = flop_per_iter_wave * iters * (double)blocks * 8;
    printf("  %-52s %8.3f ms  %8.1f TFLOP/s\n", name, ms, flop / ms / 1e9);
}

int main() {
    srand(3);
    // ---- part A: semantics --------------------------------------------------------------------------------------------------
    std::vector<uint8_t> ha(16 * 128), hb(16 * 128);
    auto rnd = []() { float s = 0; for (int k = 0; k < 4; ++k) s += (float)rand() / RAND_MAX - 0.5f; return s * 1.7320508f; };
    for (auto& v : ha) v = enc_nearest(rnd() * 3.0f, dec_e5m2);
    for (auto& v : hb) v = enc_nearest(rnd() * 40.0f, dec_e4m3);
    uint8_t *da, *db;
    float* dc;
    (void)hipMalloc(&da, ha.size());
    (void)hipMalloc(&db, hb.size());
    (void)hipMalloc(&dc, 256 * sizeof(float));
    (void)hipMemcpy(da, ha.data(), ha.size(), hipMemcpyHostToDevice);
    (void)hipMemcpy(db, hb.data(), hb.size(), hipMemcpyHostToDevice);
    printf("part A: v_mfma_scale_f32_16x16x128_f8f6f4, A e5m2 (cbsz 1) x B e4m3 (blgp 0), against a double-precision contraction of the decoded bytes\n");
    const int scales[][2] = {{127, 127}, {116, 127}, {127, 110}, {120, 130}, {0, 0}};
    for (int mode = 0; mode < 2; ++mode)
        for (auto& sc : scales) {
            const int sa = sc[0] * 0x01010101, sb = sc[1] * 0x01010101;
            hipLaunchKernelGGL(k_sem, dim3(1), dim3(64), 0, 0, da, db, dc, sa, sb, mode);
            std::vector<float> hc(256);
            (void)hipMemcpy(hc.data(), dc, 256 * sizeof(float), hipMemcpyDeviceToHost);
            double worst = 0, mag = 0;
            for (int i = 0; i < 16; ++i)
                for (int j = 0; j < 16; ++j) {
                    double ref = 0;
                    for (int k = 0; k < 128; ++k) ref += (double)dec_e5m2(ha[i * 128 + k]) * (double)dec_e4m3(hb[j * 128 + k]);
                    ref *= std::ldexp(1.0, sc[0] - 127) * std::ldexp(1.0, sc[1] - 127);
                    worst = std::fmax(worst, std::fabs(ref - hc[i * 16 + j]));
                    mag = std::fmax(mag, std::fabs(ref));
                }
            if (mag < 1e-37) {   // 2^-254: below fp32 -- the instruction must return 0
                printf("  k map %d  scale bytes (%3d, %3d): reference %.3e is below fp32, device returns %s\n", mode, sc[0], sc[1], mag, worst <= mag ? "0" : "something else");
                continue;
            }
            printf("  k map %d  scale bytes (%3d, %3d): max |device - reference| %.3e of max |reference| %.3e  -> %s\n", mode, sc[0], sc[1], worst, mag,
                   worst <= 1e-4 * mag ? "agrees (the instruction aligns its 128 products before adding: ~2^-16 of the result, not fp32 rounding)" : "DIFFERS");
        }
    {   // conversions
        std::vector<float> hx;
        for (int e = -20; e <= 17; ++e)
            for (int m = 0; m < 32; ++m) {
                hx.push_back(std::ldexp(1.0f + m / 32.0f, e));
                hx.push_back(-std::ldexp(1.0f + m / 32.0f + 1.0f / 64.0f, e));
            }
        const int n = (int)hx.size();
        float* dx;
        uint8_t *d5, *d4;
        (void)hipMalloc(&dx, n * sizeof(float));
        (void)hipMalloc(&d5, n);
        (void)hipMalloc(&d4, n);
        (void)hipMemcpy(dx, hx.data(), n * sizeof(float), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_cvt, dim3((n / 2 + 63) / 64), dim3(64), 0, 0, dx, d5, d4, n);
        std::vector<uint8_t> h5(n), h4(n);
        (void)hipMemcpy(h5.data(), d5, n, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h4.data(), d4, n, hipMemcpyDeviceToHost);
        int bad5 = 0, bad4 = 0, in5 = 0, in4 = 0;
        float first5 = 0, first4 = 0;
        for (int i = 0; i < n; ++i) {
            const float x = hx[i];
            if (std::fabs(x) <= 57344.0f) { ++in5; if (h5[i] != enc_nearest(x, dec_e5m2)) { if (!bad5) first5 = x; ++bad5; } }
            if (std::fabs(x) <= 448.0f) { ++in4; if (h4[i] != enc_nearest(x, dec_e4m3)) { if (!bad4) first4 = x; ++bad4; } }
        }
        printf("  v_cvt_pk_bf8_f32: %d of %d in-range values differ from round-to-nearest-even (first %g)\n", bad5, in5, first5);
        float probe[4] = {60000.0f, 70000.0f, 460.0f, 500.0f};
        (void)hipMemcpy(dx, probe, sizeof(probe), hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_cvt, dim3(1), dim3(64), 0, 0, dx, d5, d4, 4);
        (void)hipMemcpy(h5.data(), d5, 4, hipMemcpyDeviceToHost);
        (void)hipMemcpy(h4.data(), d4, 4, hipMemcpyDeviceToHost);
        printf("  above the range: bf8(60000) = 0x%02x, bf8(70000) = 0x%02x (0x7b = 57344, 0x7c = inf); fp8(460) = 0x%02x, fp8(500) = 0x%02x (0x7e = 448, 0x7f = nan)\n", h5[0], h5[1], h4[2], h4[3]);
        printf("  v_cvt_pk_fp8_f32: %d of %d in-range values differ from round-to-nearest-even (first %g)\n", bad4, in4, first4);
    }
    // ---- part B: rate ---------------------------------------------------------------------------------------------------------
    const size_t nthreads = 256 * 512;
    std::vector<_Float16> h(nthreads * 12 * 16);
    for (auto& v : h) v = (_Float16)rnd();
    v8i* src;
    float* out;
    (void)hipMalloc(&src, h.size() * 2);
    (void)hipMalloc(&out, nthreads * sizeof(float));
    (void)hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 8000;
    printf("part B: sustained rate, registers only, 8 waves per CU (2 per SIMD), 128 x 64 wave tile; random bits as operands (fp8: random bytes, finite or not)\n");
    run("fp16 16x16x32 (64 k per iteration)", k_rate<0>, src, out, iters, 2.0 * 128 * 64 * 64);
    run("fp8 16x16x128 scaled (128 k per iteration)", k_rate<1>, src, out, iters, 2.0 * 128 * 64 * 128);
    run("fp16 16x16x32 again", k_rate<0>, src, out, iters, 2.0 * 128 * 64 * 64);
    return 0;
}
