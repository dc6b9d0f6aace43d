// Micro-benchmark (yardstick, not product): how well do the LDS pipe and the MFMA pipe of a gfx950 CU overlap when two waves per
// SIMD alternate between fragment reads and MFMAs the way the igemm main loop does?  8 waves per workgroup, 1 workgroup per CU.
// Per iteration and wave: NR ds_read_b128 (conflict-free, 1 KiB each) feeding NM v_mfma_f32_16x16x32_f16 (16 cycles each).
//   mode 0: MFMAs only      mode 1: LDS reads only      mode 2: both, as the compiler schedules them
//   hipcc --offload-arch=gfx950 -O3 -o lds_mfma_overlap lds_mfma_overlap.hip && ./lds_mfma_overlap
#include <hip/hip_runtime.h>
#include <cstdio>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

__device__ unsigned long long g_ticks[4];

template <int MODE, int NR, int NM>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // fill 128 KiB of LDS with something
    for (int i = tid; i < 131072 / 16; i += 512) ((float4*)smem)[i] = make_float4(i * 1e-4f, 1.f, -1.f, 0.5f);
    __syncthreads();
    f4 acc[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) acc[i] = f4{0, 0, 0, 0};
    h8 fr[12];
#pragma unroll
    for (int i = 0; i < 12; ++i) fr[i] = h8{1, 0.5, 0.25, 2, 1, 1, 1, 1};
    // swizzled, conflict-free fragment addresses: row = lane&15, chunk = (lane>>4) ^ ((row>>1)&7), 128-byte rows
    const int base = (wave * 8192) + (lane & 15) * 128 + ((((lane >> 4)) ^ (((lane & 15) >> 1) & 7)) * 16);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE != 0) {
#pragma unroll
            for (int r = 0; r < NR; ++r) fr[r % 12] = *(const h8*)(smem + ((base + r * 2048 + (it & 3) * 16384) & 131071));
        }
        if (MODE != 1) {
#pragma unroll
            for (int m = 0; m < NM; ++m) acc[m % 32] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fr[(m / 4) % 12], fr[(m + 5) % 12], acc[m % 32], 0, 0, 0);
        } else {
            float s = 0;
#pragma unroll
            for (int r = 0; r < 12; ++r) s += (float)fr[r][0];
            acc[0][0] += s;
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 32; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    out[blockIdx.x * 512 + tid] = s;
    if (blockIdx.x == 0 && lane == 0) {   // span over all 8 waves of the workgroup (the oldest wave of a SIMD is favoured)
        atomicMin(&g_ticks[1], t0);
        atomicMax(&g_ticks[2], t1);
    }
}

template <int MODE, int NR, int NM>
static void run(const char* name, float* out, int iters) {
    auto kern = k<MODE, NR, NM>;
    (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, out, iters);
    (void)hipDeviceSynchronize();
    unsigned long long init[4] = {0, ~0ull, 0, 0};
    (void)hipMemcpyToSymbol(HIP_SYMBOL(g_ticks), init, 32);
    hipLaunchKernelGGL(kern, dim3(256), dim3(512), 131072, 0, out, iters);
    (void)hipDeviceSynchronize();
    unsigned long long t[4];
    (void)hipMemcpyFromSymbol(t, HIP_SYMBOL(g_ticks), 32);
    printf("  %-52s %8.0f cycles per iteration (all 8 waves of a workgroup)\n", name, (double)(t[2] - t[1]) / iters);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * sizeof(float));
    const int iters = 4000;
    printf("per iteration and wave: 12 ds_read_b128 (12 KiB) and 32 v_mfma_f32_16x16x32_f16; 2 waves per SIMD\n");
    printf("bounds per iteration: MFMA pipe 2 x 32 x 16 = 1024 cycles per SIMD; LDS pipe 8 x 12 KiB / 128 B = 768 cycles per CU\n");
    run<0, 12, 32>("MFMAs only", out, iters);
    run<1, 12, 32>("LDS reads only", out, iters);
    run<2, 12, 32>("both (compiler schedule)", out, iters);
    run<2, 24, 32>("both, twice the reads (24 KiB per wave)", out, iters);
    run<2, 6, 32>("both, half the reads (6 KiB per wave)", out, iters);
    return 0;
}
