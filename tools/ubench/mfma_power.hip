// Micro-benchmark (yardstick, not product): sustained MFMA rate of gfx950 under its power cap, registers only.
// Each wave cycles through 4 sub-steps x (4 A + 4 B) fragments held in registers -- the operand pattern of a 128x128 wave tile
// -- and issues MFMAs back to back for a few milliseconds.  Compares v_mfma_f32_32x32x16_f16 with v_mfma_f32_16x16x32_f16,
// random against all-zero operands, one against two waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 -o mfma_power mfma_power.hip && ./mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef float f4v __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_ticks;

template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 1 : 2) void k32(const h8* __restrict__ src, float* __restrict__ out, int iters) {
    h8 fa[4][4], fb[4][4];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            fa[s][i] = src[(size_t)t * 32 + s * 8 + i];
            fb[s][i] = src[(size_t)t * 32 + s * 8 + 4 + i];
        }
    f16v acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (t == 0) g_ticks = t1 - t0;
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) sum += acc[i][j][r];
    out[t] = sum;
}

// same register footprint and FLOPs per iteration: a 128x128 wave tile as 8x8 tiles of 16x16, k = 32 per sub-step, 2 sub-steps
template <int WAVES>
__global__ __launch_bounds__(WAVES * 64, WAVES == 4 ? 1 : 2) void k16(const h8* __restrict__ src, float* __restrict__ out, int iters) {
    h8 fa[2][8], fb[2][8];
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            fa[s][i] = src[(size_t)t * 32 + s * 16 + i];
            fb[s][i] = src[(size_t)t * 32 + s * 16 + 8 + i];
        }
    f4v acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[i][j][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa[s][i], fb[s][j], acc[i][j], 0, 0, 0);
    }
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) sum += acc[i][j][r];
    out[t] = sum;
}

template <typename K>
static void run(const char* name, K kern, int waves, int blocks_per_cu, const h8* src, float* out, int iters, double flop_per_iter_wave) {
    const int blocks = 256 * blocks_per_cu;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, src, out, iters);
    (void)hipDeviceSynchronize();
    const int reps = 5;
    (void)hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, dim3(blocks), dim3(waves * 64), 0, 0, src, out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    ms /= reps;
    const double flop = flop_per_iter_wave * iters * (double)blocks * waves;
    unsigned long long ticks = 0;
    (void)hipMemcpyFromSymbol(&ticks, HIP_SYMBOL(g_ticks), sizeof(ticks));
    printf("  %-44s %8.3f ms  %8.1f TFLOP/s   s_memtime ticks/MFMA (k32 only) %.2f -> %.3f GHz if tick = shader clock\n", name, ms, flop / ms / 1e9,
           (double)ticks / (64.0 * iters), (double)ticks / (ms * 1e6));
}

int main() {
    const size_t nthreads = 256 * 2 * 512;
    const size_t n = nthreads * 32;  // h8 elements
    std::vector<_Float16> h(n * 8);
    srand(1);
    for (auto& v : h) {  // ~N(0,1) by sum of uniforms
        float s = 0;
        for (int k = 0; k < 4; ++k) s += (float)rand() / RAND_MAX - 0.5f;
        v = (_Float16)(s * 1.7320508f);
    }
    h8 *rnd, *zero;
    float* out;
    (void)hipMalloc(&rnd, n * sizeof(h8));
    (void)hipMalloc(&zero, n * sizeof(h8));
    (void)hipMalloc(&out, nthreads * sizeof(float));
    (void)hipMemcpy(rnd, h.data(), n * sizeof(h8), hipMemcpyHostToDevice);
    (void)hipMemset(zero, 0, n * sizeof(h8));
    const double fl = 64.0 * 32 * 32 * 16 * 2;  // per iteration per wave: both kernels do a 128x128x64 update
    const int iters = 6000;
    printf("sustained MFMA rate, registers only (dense fp16 peak 2500 TFLOP/s at 2.4 GHz)\n");
    run("32x32x16 f16, 1 wave/SIMD, random", k32<4>, 4, 1, rnd, out, iters, fl);
    run("32x32x16 f16, 1 wave/SIMD, zeros", k32<4>, 4, 1, zero, out, iters, fl);
    run("16x16x32 f16, 1 wave/SIMD, random", k16<4>, 4, 1, rnd, out, iters, fl);
    run("16x16x32 f16, 1 wave/SIMD, zeros", k16<4>, 4, 1, zero, out, iters, fl);
    run("32x32x16 f16, 2 waves/SIMD (2 WG/CU), random", k32<4>, 4, 2, rnd, out, iters, fl);
    run("16x16x32 f16, 2 waves/SIMD (2 WG/CU), random", k16<4>, 4, 2, rnd, out, iters, fl);
    return 0;
}
