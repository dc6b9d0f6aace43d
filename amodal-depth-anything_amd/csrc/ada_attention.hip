// Fused softmax(q k^T) v for head_dim 64 (see include/ada_hip.h: ada_attention_fwd).
//
// Work decomposition (gfx950): one workgroup = 4 waves = 128 query rows of one (batch, head); each
// wave owns 32 query rows for the whole key loop.  Keys/values are streamed in tiles of 64 through a
// two-stage LDS ring (register-staged: the global loads of tile t+1 are issued before the MFMAs of
// tile t and written to LDS after them -- cdna_hip_programming.md T14).
//
//   S^T = K Q^T   "swapped" product: v_mfma_f32_32x32x16 with A = K fragment (LDS, XOR-swizzled
//                 128-byte rows, ds_read_b128), B = Q fragment (registers, loaded once).  The result
//                 puts one query per lane (col = lane&31) and 32 keys of the tile in the lane's
//                 registers, so the softmax row statistics are lane-local plus ONE exchange with
//                 lane^32 -- wavefront shuffles, no LDS.
//   O^T += V^T P^T  A = V^T fragment read straight from the row-major V tile with the gfx950 LDS
//                 transpose read (ds_read_b64_tr_b16), B = P in registers.  The key order inside a
//                 k-step is a free permutation as long as A and B agree, so P needs no cross-lane
//                 movement at all: k-slot (hi, j) of step s <-> key 16s + 4hi + (j&3) + 8(j>>2).
//   Online softmax in fp32 (exp2 with log2(e) folded in), running max / sum per lane, O rescaled by
//   the lane-local alpha.  N = 1370 is not a multiple of 64: the last tile masks keys >= N to -inf.
#include "ada_common.h"

namespace {

constexpr int HD = 64;          // head dim
constexpr int QBLK = 128;       // query rows per workgroup
constexpr int KVB = 64;         // keys per tile
constexpr int K_ROW = 128;      // bytes per K row in LDS
constexpr int V_ROW = 192;      // bytes per V row in LDS (64 B pad: 4 consecutive rows hit disjoint banks)
constexpr int K_TILE = KVB * K_ROW;
constexpr int V_TILE = KVB * V_ROW;
constexpr int STAGE = K_TILE + V_TILE;

typedef __attribute__((ext_vector_type(4))) short s16x4;

ADA_DEV opx4 lds_tr_read(const char* p) {
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(opx4, v);
}

__global__ __launch_bounds__(256, 2) void attention_kernel(const op_t* __restrict__ qkv, op_t* __restrict__ out,
                                                           int n_tok, int heads, int nqb, int n_bh) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int l31 = lane & 31;
    const int hi = lane >> 5;

    // XCD-aware remap: the q-blocks of one (batch, head) run on one XCD so K/V stay in its L2.
    int bh, qb;
    {
        const int nblk = gridDim.x, bid = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7;
        const int xcd = bid & 7, idx = bid >> 3;
        const int logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
        bh = logical / nqb;
        qb = logical - bh * nqb;
    }
    const int b = bh / heads, h = bh - b * heads;
    const long D = (long)heads * HD;
    const long row_stride = 3 * D;  // elements between consecutive tokens in qkv
    const op_t* base = qkv + (long)b * n_tok * row_stride + (long)h * HD;
    const op_t* qbase = base;
    const op_t* kbase = base + D;
    const op_t* vbase = base + 2 * D;

    // ---- Q fragments (B operand of S^T = K Q^T): Q[q][16s + 8hi + j] ---------------------
    const int q_row = qb * QBLK + wave * 32 + l31;
    const int q_ld = q_row < n_tok ? q_row : n_tok - 1;
    opx8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const opx8*)(qbase + (long)q_ld * row_stride + 16 * s + 8 * hi);

    // ---- staging: each thread moves 2 K chunks + 2 V chunks (16 B each) per tile ------------
    const int srow = tid >> 3;  // 0..31
    const int sc = tid & 7;
    u32x4 kreg[2], vreg[2];
    auto load_tile = [&](int t) {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int key = t * KVB + it * 32 + srow;
            if (key >= n_tok) key = n_tok - 1;
            const long off = (long)key * row_stride + sc * 8;
            kreg[it] = *(const u32x4*)(kbase + off);
            vreg[it] = *(const u32x4*)(vbase + off);
        }
    };
    auto write_tile = [&](int buf) {
        char* ks = smem + buf * STAGE;
        char* vs = ks + K_TILE;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            const int row = it * 32 + srow;
            *(u32x4*)(ks + row * K_ROW + ((sc ^ ((row >> 1) & 7)) * 16)) = kreg[it];
            *(u32x4*)(vs + row * V_ROW + sc * 16) = vreg[it];
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m_run = -INFINITY, l_run = 0.0f;
    constexpr float LOG2E = 1.4426950408889634f;

    const int swz = (l31 >> 1) & 7;
    const int k_frag_off = l31 * K_ROW;
    // transpose-read address: lane i of a 16-lane group supplies row (i>>2), 4 columns at 4*(i&3)
    const int i16 = lane & 15;
    const int g1 = (lane >> 4) & 1;
    const int v_frag_off = (4 * hi + (i16 >> 2)) * V_ROW + (16 * g1 + 4 * (i16 & 3)) * 2;

    const int nt = (n_tok + KVB - 1) / KVB;
    load_tile(0);
    write_tile(0);
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) load_tile(t + 1);
        const char* ks = smem + cur * STAGE;
        const char* vs = ks + K_TILE;

        // ---- S^T = K Q^T -------------------------------------------------------------------
        f32x16 sT[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[kb][r] = 0.0f;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const opx8 kf = *(const opx8*)(ks + kb * 32 * K_ROW + k_frag_off + (((2 * s + hi) ^ swz) * 16));
                sT[kb] = mfma32(kf, qf[s], sT[kb]);
            }
        }
        if (t == nt - 1) {  // mask keys beyond the sequence (wave-uniform branch)
            const int kv0 = t * KVB;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + crow32(r, hi) >= n_tok) sT[kb][r] = -INFINITY;
        }

        // ---- online softmax ------------------------------------------------------------------
        float mx = sT[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = __builtin_fmaxf(mx, sT[kb][r]);
        mx = __builtin_fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = __builtin_fmaxf(m_run, mx);
        const float alpha = __builtin_amdgcn_exp2f((m_run - m_new) * LOG2E);
        const float mb = m_new * LOG2E;
        float psum = 0.0f;
        opx8 pf[2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float pv = __builtin_amdgcn_exp2f(__builtin_fmaf(sT[kb][r], LOG2E, -mb));
                psum += pv;
                pf[kb][r >> 3][r & 7] = (op_t)pv;
            }
        }
        l_run = l_run * alpha + psum;
        m_run = m_new;
#pragma unroll
        for (int db = 0; db < 2; ++db)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[db][r] *= alpha;

        // ---- O^T += V^T P^T --------------------------------------------------------------------
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const char* vp = vs + (kb * 32 + s * 16) * V_ROW + db * 64 + v_frag_off;
                    const opx4 lo = lds_tr_read(vp);
                    const opx4 hi4 = lds_tr_read(vp + 8 * V_ROW);
                    opx8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
                    o[db] = mfma32(vf, pf[kb][s], o[db]);
                }
            }
        }

        if (t + 1 < nt) write_tile(cur ^ 1);
        __syncthreads();
    }

    // ---- normalise and store: lane holds q = l31, d = 32db + crow32(r, hi) -----------------------
    const float l_tot = l_run + __shfl_xor(l_run, 32);
    const float inv = 1.0f / l_tot;
    if (q_row < n_tok) {
        op_t* orow = out + ((long)b * n_tok + q_row) * D + (long)h * HD;
#pragma unroll
        for (int db = 0; db < 2; ++db) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                opx4 v4;
#pragma unroll
                for (int e = 0; e < 4; ++e) v4[e] = to_op(o[db][g * 4 + e] * inv);
                *(opx4*)(orow + db * 32 + 8 * g + 4 * hi) = v4;
            }
        }
    }
}

}  // namespace

extern "C" int ada_attention_fwd(const void* qkv, void* out, int32_t batch, int32_t n_tokens, int32_t heads, void* stream) {
    ADA_REQUIRE(qkv && out, ADA_EINVAL, "ada_attention_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && n_tokens > 0 && heads > 0, ADA_EINVAL, "ada_attention_fwd: bad shape B=%d N=%d H=%d", batch, n_tokens, heads);
    ADA_REQUIRE(((uintptr_t)qkv % 16) == 0 && ((uintptr_t)out % 16) == 0, ADA_EINVAL, "ada_attention_fwd: buffers must be 16-byte aligned");
    const int nqb = (n_tokens + QBLK - 1) / QBLK;
    const long nblk = (long)nqb * batch * heads;
    ADA_REQUIRE(nblk < (1L << 31), ADA_EUNSUPPORTED, "ada_attention_fwd: grid too large");
    hipLaunchKernelGGL(attention_kernel, dim3((unsigned)nblk), dim3(256), 0, (hipStream_t)stream, (const op_t*)qkv, (op_t*)out,
                       n_tokens, heads, nqb, batch * heads);
    return ada_check_launch("ada_attention_fwd");
}
