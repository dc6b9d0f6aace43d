// Is packed fp32 VALU arithmetic (v_pk_mul_f32 / v_pk_fma_f32, with the op_sel forms hipcc's SLP vectoriser emits) reliable on gfx950 while the
// other wave of the SIMD streams MFMAs?  Round 3 met wrong results in the fused DPT tail whenever the producers' interpolation was SLP-vectorised
// (profiles/r03_p_fused_tail.txt) and could not tell a hardware hazard from a synchronisation bug.  This microbenchmark takes the hardware
// question on its own: waves 0-3 of a 512-thread workgroup run a recurrence made of the exact packed forms of the failing build on per-lane
// data, waves 4-7 (their SIMD partners) stream v_mfma_f32_16x16x32_f16; a checksum of EVERY intermediate result is compared bit for bit with
// the same recurrence in scalar v_mul_f32 / v_fma_f32 run without a partner.  Modes: packed alone, packed beside MFMAs, packed beside MFMAs with
// LDS traffic from both wave kinds (the packed lanes round-trip their state through LDS, the MFMA waves re-read their fragments every step).
//     hipcc --offload-arch=gfx950 -O3 -o pk_f32_beside_mfma pk_f32_beside_mfma.hip && ./pk_f32_beside_mfma
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

typedef __attribute__((ext_vector_type(8))) _Float16 h8;
typedef __attribute__((ext_vector_type(4))) float f4;
typedef __attribute__((ext_vector_type(2))) float f2;

enum { MODE_SCALAR = 0, MODE_PK_ALONE = 1, MODE_PK_MFMA = 2, MODE_PK_MFMA_LDS = 3 };

template <int MODE>
__global__ __launch_bounds__(512) void stream(uint32_t* chk_out, float* sink, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[512 * 4];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned gid = blockIdx.x * 256u + (tid & 255);
    if (tid < 256) {
        // per-lane constants: |a| around 0.7, c negative so that the recurrence contracts; w re-injects iteration-dependent data
        f2 a = {0.55f + 0.001f * (float)(gid % 197), 0.83f - 0.0007f * (float)(gid % 211)};
        f2 c = {-0.31f - 0.0003f * (float)(gid % 89), -0.27f + 0.0002f * (float)(gid % 97)};
        f2 x = {1.0f + 0.01f * (float)lane, -0.5f + 0.003f * (float)(gid % 113)};
        uint32_t chk = 0x9e3779b9u ^ gid;
        for (int it = 0; it < iters; ++it) {
            const float wv = 0.01f * (float)((it * 7 + lane) & 255) - 1.0f;
            f2 w = {wv, -wv * 0.5f};
            f2 y, z, u;
            if (MODE == MODE_SCALAR) {
                float y0, y1, z0, z1, u0, u1, n0, n1;
                asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y0) : "v"(a[0]), "v"(x[0]));
                asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(y1) : "v"(a[1]), "v"(x[1]));
                asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(z0) : "v"(a[1]), "v"(x[0]));          // op_sel:[1,0] -> low: a.hi * x.lo
                asm volatile("v_mul_f32_e32 %0, %1, %2" : "=v"(z1) : "v"(a[0]), "v"(x[1]));          // op_sel_hi:[0,1] -> high: a.lo * x.hi
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(u0) : "v"(a[1]), "v"(y0), "v"(z0));   // op_sel:[1,0,0]
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(u1) : "v"(a[0]), "v"(y1), "v"(z1));   // op_sel_hi:[0,1,1]
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(n0) : "v"(u0), "v"(c[0]), "v"(w[0]));
                asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(n1) : "v"(u1), "v"(c[1]), "v"(w[1]));
                y = f2{y0, y1}; z = f2{z0, z1}; u = f2{u0, u1}; x = f2{n0, n1};
            } else {
                f2 n;
                asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(y) : "v"(a), "v"(x));
                asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(z) : "v"(a), "v"(x));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[0,1,1]" : "=v"(u) : "v"(a), "v"(y), "v"(z));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(n) : "v"(u), "v"(c), "v"(w));
                x = n;
            }
            if (MODE == MODE_PK_MFMA_LDS) {   // state round-trips through LDS (16-byte store, 8-byte loads of its two halves)
                *(f4*)&lds[tid * 4] = f4{x[0], x[1], u[0], u[1]};
                const f2 r0 = *(const f2*)&lds[tid * 4], r1 = *(const f2*)&lds[tid * 4 + 2];
                x = r0; u = r1;
            }
            chk = chk * 31u + __builtin_bit_cast(uint32_t, y[0]) + 3u * __builtin_bit_cast(uint32_t, y[1]) + 5u * __builtin_bit_cast(uint32_t, z[0]) +
                  7u * __builtin_bit_cast(uint32_t, z[1]) + 11u * __builtin_bit_cast(uint32_t, u[0]) + 13u * __builtin_bit_cast(uint32_t, u[1]) +
                  17u * __builtin_bit_cast(uint32_t, x[0]) + 19u * __builtin_bit_cast(uint32_t, x[1]);
        }
        chk_out[gid] = chk;
    } else if (MODE >= MODE_PK_MFMA) {
        h8 fa, fb;
        for (int i = 0; i < 8; ++i) { fa[i] = (_Float16)(0.01f * (float)((tid + i) & 63) - 0.3f); fb[i] = (_Float16)(0.25f - 0.02f * (float)i); }
        f4 acc[4];
        for (int i = 0; i < 4; ++i) acc[i] = f4{0.f, 0.f, 0.f, 0.f};
        if (MODE == MODE_PK_MFMA_LDS) { *(h8*)&lds[tid * 4] = fa; }
        for (int it = 0; it < iters; ++it) {
            if (MODE == MODE_PK_MFMA_LDS) fa = *(const h8*)&lds[(256 + ((tid + it) & 255)) * 4];
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q & 3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fa, fb, acc[q & 3], 0, 0, 0);
        }
        float s = 0.f;
        for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
        sink[blockIdx.x * 256 + (tid - 256)] = s;
    }
}

template <int MODE>
static double run(uint32_t* d_chk, float* d_sink, int grid, int iters, std::vector<uint32_t>& host) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(stream<MODE>, dim3(grid), dim3(512), 0, 0, d_chk, d_sink, 16);   // warm-up
    hipEventRecord(e0);
    hipLaunchKernelGGL(stream<MODE>, dim3(grid), dim3(512), 0, 0, d_chk, d_sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(host.data(), d_chk, host.size() * 4, hipMemcpyDeviceToHost);
    return ms;
}

int main(int argc, char** argv) {
    const int grid = 2048, iters = argc > 1 ? atoi(argv[1]) : 4000, reps = argc > 2 ? atoi(argv[2]) : 5;
    const size_t n = (size_t)grid * 256;
    uint32_t* d_chk; float* d_sink;
    hipMalloc(&d_chk, n * 4); hipMalloc(&d_sink, n * 4);
    std::vector<uint32_t> ref(n), got(n);
    run<MODE_SCALAR>(d_chk, d_sink, grid, iters, ref);
    printf("pk_f32_beside_mfma: %d workgroups x 256 packed lanes x %d iterations x 4 packed instructions = %.2e packed fp32 element results per run, %d runs per mode\n",
           grid, iters, 2.0 * 4.0 * (double)n * iters, reps);
    const char* names[] = {"scalar (reference)", "packed, no partner", "packed beside v_mfma_f32_16x16x32_f16 partner waves", "packed beside MFMA partners, LDS traffic in both"};
    long total_bad = 0;
    for (int mode = 1; mode <= 3; ++mode) {
        long bad = 0; double ms = 0;
        for (int r = 0; r < reps; ++r) {
            ms = mode == 1 ? run<MODE_PK_ALONE>(d_chk, d_sink, grid, iters, got) : mode == 2 ? run<MODE_PK_MFMA>(d_chk, d_sink, grid, iters, got)
                                                                                             : run<MODE_PK_MFMA_LDS>(d_chk, d_sink, grid, iters, got);
            for (size_t i = 0; i < n; ++i) bad += got[i] != ref[i];
        }
        printf("  %-52s %8.2f ms/run   lanes whose checksum differs from the scalar reference: %ld of %zu x %d\n", names[mode], ms, bad, n, reps);
        total_bad += bad;
    }
    printf("verdict: %s\n", total_bad == 0 ? "packed fp32 results are bit-identical to scalar ones with and without MFMA partner waves -- no hardware hazard"
                                            : "MISMATCHES -- packed fp32 beside MFMA is not reliable on this part");
    return total_bad != 0;
}
