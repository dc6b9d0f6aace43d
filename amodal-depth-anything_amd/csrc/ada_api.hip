// Library plumbing: version / dtype queries, thread-local error text, and the hardware self-test that
// pins the MFMA and LDS-transpose fragment layouts every kernel in this library is written against.
#include <stdarg.h>
#include <string.h>
#include "ada_common.h"

static thread_local char g_err[512] = "";

void ada_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ada_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ada_set_error("%s: %s", what, hipGetErrorString(e));
        return ADA_ELAUNCH;
    }
    return ADA_OK;
}

extern "C" int ada_abi_version(void) { return ADA_ABI_VERSION; }
extern "C" int ada_operand_dtype(void) { return ADA_OP_DTYPE; }
extern "C" const char* ada_last_error(void) { return g_err; }

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4;

// bit 0: 32x32x16 A/B/D layout; bit 1: 16x16x32 layout; bit 2: ds_read_b64_tr_b16 layout
__global__ __launch_bounds__(64) void selftest_kernel(unsigned* result) {
    __shared__ __attribute__((aligned(16))) short lds[64 * 64];
    const int lane = threadIdx.x;
    unsigned fail = 0;

    // asymmetric small-integer operands (exact in fp16 and bf16): A[i][k] = (i*3 + k*5) % 7 - 3, B[k][n] = (k*2 + n*7) % 5 - 2
    auto Aval = [](int i, int k) { return (float)((i * 3 + k * 5) % 7 - 3); };
    auto Bval = [](int k, int n) { return (float)((k * 2 + n * 7) % 5 - 2); };
    {
        opx8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (op_t)Aval(lane & 31, 8 * (lane >> 5) + j);
            b[j] = (op_t)Bval(8 * (lane >> 5) + j, lane & 31);
        }
        f32x16 c;
        for (int r = 0; r < 16; ++r) c[r] = 0.f;
        c = mfma32(a, b, c);
        for (int r = 0; r < 16; ++r) {
            const int row = crow32(r, lane >> 5), col = lane & 31;
            float ref = 0.f;
            for (int k = 0; k < 16; ++k) ref += Aval(row, k) * Bval(k, col);
            if (c[r] != ref) fail |= 1u;
        }
    }
    {
        opx8 a, b;
        for (int j = 0; j < 8; ++j) {
            a[j] = (op_t)Aval(lane & 15, 8 * (lane >> 4) + j);
            b[j] = (op_t)Bval(8 * (lane >> 4) + j, lane & 15);
        }
        f32x4 c = {0.f, 0.f, 0.f, 0.f};
        c = mfma16(a, b, c);
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * (lane >> 4) + r, col = lane & 15;
            float ref = 0.f;
            for (int k = 0; k < 32; ++k) ref += Aval(row, k) * Bval(k, col);
            if (c[r] != ref) fail |= 2u;
        }
    }
    {
        // LDS image M[row][col] = row*64 + col (row stride 64 shorts).  Lane i of each 16-lane group g supplies the
        // address of row 4g + (i>>2), columns 4*(i&3)..+3; expected result: element e = M[4g + e][i].
        for (int idx = lane; idx < 64 * 64; idx += 64) lds[idx] = (short)idx;
        __syncthreads();
        const int i = lane & 15, g = lane >> 4;
        const short* addr = &lds[(4 * g + (i >> 2)) * 64 + 4 * (i & 3)];
        s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)addr);
        for (int e = 0; e < 4; ++e)
            if (v[e] != (short)((4 * g + e) * 64 + i)) fail |= 4u;
    }
    if (fail) atomicOr(result, fail);
}

// counts operand-typed elements that sit at the saturation value of to_op (|x| >= 65504 for fp16; bf16 has fp32's range, so there
// only +-inf / NaN count) -- one 8-byte-per-lane strided sweep, per-wave shuffle reduction, one atomic per wave
__global__ __launch_bounds__(256) void count_saturated_kernel(const op_t* __restrict__ buf, long n, unsigned long long* __restrict__ count) {
    unsigned c = 0;
    for (long i = ((long)blockIdx.x * 256 + threadIdx.x) * 4; i < n; i += (long)gridDim.x * 1024) {
        if (i + 4 <= n) {
            const opx4 v = *(const opx4*)(buf + i);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float a = __builtin_fabsf((float)v[e]);
                c += !(a < ADA_OP_SATURATION);     // also counts NaN
            }
        } else {
            for (long k = i; k < n; ++k) c += !(__builtin_fabsf((float)buf[k]) < ADA_OP_SATURATION);
        }
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) c += __shfl_xor(c, o);
    if ((threadIdx.x & 63) == 0 && c) atomicAdd(count, (unsigned long long)c);
}

}  // namespace

extern "C" int ada_debug_count_saturated(const void* buf, int64_t n, void* count, void* stream) {
    ADA_REQUIRE(buf && count && n >= 0, ADA_EINVAL, "ada_debug_count_saturated: null pointer / negative size");
    ADA_REQUIRE(((uintptr_t)buf % 8) == 0 && ((uintptr_t)count % 8) == 0, ADA_EINVAL, "ada_debug_count_saturated: buffers must be 8-byte aligned");
    if (n == 0) return ADA_OK;
    long blocks = (n + 1023) / 1024;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(count_saturated_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const op_t*)buf, (long)n,
                       (unsigned long long*)count);
    return ada_check_launch("ada_debug_count_saturated");
}

extern "C" int ada_selftest(void* scratch, int64_t scratch_bytes, void* stream) {
    ADA_REQUIRE(scratch && scratch_bytes >= 4, ADA_EINVAL, "ada_selftest: need >= 4 bytes of device scratch");
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(scratch, 0, 4, s) != hipSuccess) return ada_check_launch("ada_selftest memset");
    hipLaunchKernelGGL(selftest_kernel, dim3(1), dim3(64), 0, s, (unsigned*)scratch);
    int rc = ada_check_launch("ada_selftest");
    if (rc) return rc;
    unsigned host = 0;
    if (hipMemcpyAsync(&host, scratch, 4, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) {
        return ada_check_launch("ada_selftest copy");
    }
    if (host) ada_set_error("ada_selftest: fragment layout probe failed, mask=0x%x", host);
    return (int)host;
}
