// Fused softmax(q k^T) v for head_dim 64 (see include/ada_hip.h: ada_attention_fwd).
//
// Work decomposition (gfx950): one workgroup = 4 waves = 128 query rows of one (batch, head); each wave owns 32
// query rows for the whole key loop.  Keys/values stream in tiles of 64 through a two-stage LDS ring filled by
// 16-byte global_load_lds (no VGPR round trip); the copies of tile t+1 are in flight during the math of tile t.
//
//   S^T = K Q^T   "swapped" product: v_mfma_f32_32x32x16 with A = K fragment (LDS rows of 128 B, XOR-swizzled through
//                 the copy's SOURCE address, ds_read_b128), B = Q fragment (registers, loaded once).  One query per
//                 lane (col = lane&31), 32 keys of the tile in the lane's registers: softmax statistics are lane-local
//                 plus ONE v_permlane32_swap with the other half-wave -- wavefront shuffles, no LDS.
//   running max   q arrives pre-scaled by head_dim^-0.5 * log2(e), so scores are in log2 units.  The running max m
//                 (kept fp16-representable) is subtracted INSIDE the MFMA chain: one extra k-step whose K fragment is
//                 the constant 1 and whose Q fragment is -m.  The per-element work is then exp2 + convert + sum only.
//   defer-max     m is only moved (O and l rescaled) when a tile's max exceeds it by 2^8 -- rare after the first tile.
//   O^T += V^T P^T  A = V^T fragment read from the row-major V tile with the gfx950 LDS transpose read
//                 (ds_read_b64_tr_b16), B = P in registers.  The key order inside a k-step is a free permutation as
//                 long as A and B agree, so P needs no cross-lane movement: k-slot (hi, j) of step s <-> key
//                 16s + 4hi + (j&3) + 8(j>>2).  Row sums of P use v_dot2_f32_f16 on the packed fragments.
//   LDS reads are inline asm with hand-placed s_waitcnt: all 8 K reads are issued ahead of the QK^T MFMAs and all 16
//   V transpose reads BEFORE the softmax so they land behind it (hipcc sinks each ds_read next to its MFMA otherwise).
//   N = 1370 is not a multiple of 64: the last tile masks keys >= N to -inf.
#include "ada_common.h"

namespace {

constexpr int HD = 64;          // head dim
constexpr int QBLK = 128;       // query rows per workgroup
constexpr int KVB = 64;         // keys per tile
constexpr int ROWB = 128;       // bytes per K / V row in LDS
constexpr int K_TILE = KVB * ROWB;
constexpr int V_TILE = KVB * ROWB;
constexpr int STAGE = K_TILE + V_TILE;

// Exchange between the two 32-lane halves of a wave: returns {value held by lane&31, value held by (lane&31)+32} on
// every lane.  v_permlane32_swap swaps row 1 of vdst with row 0 of src; feeding it two copies of x leaves x_lo in vdst
// and x_hi in src, broadcast to both halves.  Written as inline asm on purpose: with hipcc (ROCm 7.2) the builtin's
// second result is lowered as a copy of the first one when both operands carry the same value (checked in the .s), which
// silently drops the exchange.  "s_nop 1" = the two wait states a VALU write needs before v_permlane* reads it.
ADA_DEV void half_exchange(float x, float& lo, float& hi_) {
    unsigned a = __builtin_bit_cast(unsigned, x), b = a;
    asm volatile("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    lo = __builtin_bit_cast(float, a);
    hi_ = __builtin_bit_cast(float, b);
}

ADA_DEV float dot2_acc(opx2 p, float acc) {
#ifdef ADA_OPERAND_BF16
    return acc + (float)p[0] + (float)p[1];
#else
    const opx2 ones = {(op_t)1.0f, (op_t)1.0f};
    return __builtin_amdgcn_fdot2(p, ones, acc, false);
#endif
}

__global__ __launch_bounds__(256, 2) void attention_kernel(const op_t* __restrict__ qkv, op_t* __restrict__ out,
                                                           int n_tok, int heads, int nqb, int n_bh) {
    __shared__ __attribute__((aligned(16))) char smem[2 * STAGE];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
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
    const op_t* qptr = base;
    const op_t* kptr = base + D;
    const op_t* vptr = base + 2 * D;

    // ---- Q fragments (B operand of S^T = K Q^T): Q[q][16s + 8hi + j] ---------------------
    const int q_row = qb * QBLK + wave * 32 + l31;
    const int q_ld = q_row < n_tok ? q_row : n_tok - 1;
    const bool wave_active = (qb * QBLK + wave * 32) < n_tok;   // wave-uniform
    opx8 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) qf[s] = *(const opx8*)(qptr + (long)q_ld * row_stride + 16 * s + 8 * hi);

    // ---- staging: each lane copies 2 K chunks + 2 V chunks (16 B) per tile straight into LDS.  The LDS image is
    //      lane-linear (row = it*32 + tid>>3, chunk = tid&7); the bank swizzles are applied to the SOURCE chunk:
    //      K: c ^ ((row>>1)&7) (ds_read_b128 conflict-free);  V: c ^ 4*((row>>1)&1) (4 consecutive rows of a
    //      transpose read land in 4 disjoint 64-byte bank segments).
    const int srow = tid >> 3;  // 0..31
    const int sc = tid & 7;
    const int k_src_chunk = sc ^ ((srow >> 1) & 7);
    const int v_src_chunk = sc ^ (((srow >> 1) & 1) << 2);
    auto stage = [&](int buf, int t) {
        char* dst = smem + buf * STAGE + wave * 1024;
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            int key = t * KVB + it * 32 + srow;
            if (key >= n_tok) key = n_tok - 1;
            const long off = (long)key * row_stride;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(kptr + off + k_src_chunk * 8),
                                             (__attribute__((address_space(3))) void*)(dst + it * 4096), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(vptr + off + v_src_chunk * 8),
                                             (__attribute__((address_space(3))) void*)(dst + K_TILE + it * 4096), 16, 0, 0);
        }
    };

    f32x16 o[2];
#pragma unroll
    for (int db = 0; db < 2; ++db)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[db][r] = 0.0f;
    float m_run = 0.0f, l_run = 0.0f;      // m_run in log2 units, always exactly representable in the operand type
    constexpr float RESCALE_THR = 8.0f;    // P <= 2^8: comfortably inside fp16; O / l are rescaled only when the max jumps

    // extra k-step that subtracts the running max inside the MFMA chain: K' = [1, 0, ...], Q' = [-m, 0, ...]
    opx8 kx, qx;
#pragma unroll
    for (int j = 0; j < 8; ++j) { kx[j] = (op_t)0.0f; qx[j] = (op_t)0.0f; }
    if (hi == 0) kx[0] = (op_t)1.0f;

    const unsigned lds0 = (unsigned)(size_t)smem;
    const int swz = (l31 >> 1) & 7;
    unsigned kofs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kofs[s] = l31 * ROWB + (((2 * s + hi) ^ swz) * 16);
    // transpose read: lane i of a 16-lane group supplies row (i>>2), 4 columns at 4*(i&3); d-block db lives in the
    // 64-byte half (db ^ swizzle bit) of the row, swizzle bit = (row>>1)&1 = (i>>3)&1 for every row this lane touches
    const int i16 = lane & 15;
    const int g1 = (lane >> 4) & 1;
    const int vsw = (i16 >> 3) & 1;
    const unsigned v_row_part = (4 * hi + (i16 >> 2)) * ROWB + (16 * g1 + 4 * (i16 & 3)) * 2;
    const unsigned vofs[2] = {v_row_part + (vsw ? 64u : 0u), v_row_part + (vsw ? 0u : 64u)};

    const int nt = (n_tok + KVB - 1) / KVB;
    stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) stage(cur ^ 1, t + 1);
        const unsigned kbase_l = lds0 + cur * STAGE;
        const unsigned vbase0 = kbase_l + K_TILE + vofs[0], vbase1 = kbase_l + K_TILE + vofs[1];

        // waves whose 32 query rows all lie beyond the sequence (last q-block only) keep copying and synchronising
        // but skip the math
        if (wave_active) {
        // ---- S^T = K Q^T - m -------------------------------------------------------------------
        opx8 kf[2][4];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(kf[kb][s]) : "v"(kbase_l + kofs[s]), "i"(kb * 32 * ROWB));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        f32x16 sT[2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; ++r) sT[kb][r] = 0.0f;
            sT[kb] = mfma32(kx, qx, sT[kb]);
#pragma unroll
            for (int s = 0; s < 4; ++s) sT[kb] = mfma32(kf[kb][s], qf[s], sT[kb]);
        }
        __builtin_amdgcn_sched_barrier(0);
        // V^T fragments for the whole tile: in flight while the softmax below runs
        opx4 vlo[2][2][2], vhi[2][2][2];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[kb][s][0]) : "v"(vbase0), "i"((kb * 32 + s * 16) * ROWB));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[kb][s][0]) : "v"(vbase0), "i"((kb * 32 + s * 16 + 8) * ROWB));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vlo[kb][s][1]) : "v"(vbase1), "i"((kb * 32 + s * 16) * ROWB));
                asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(vhi[kb][s][1]) : "v"(vbase1), "i"((kb * 32 + s * 16 + 8) * ROWB));
            }
        if (t == nt - 1) {  // mask keys beyond the sequence (wave-uniform branch)
            const int kv0 = t * KVB;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (kv0 + kb * 32 + crow32(r, hi) >= n_tok) sT[kb][r] = -INFINITY;
        }

        // ---- online softmax in base 2; sT already holds s - m_run ------------------------------------
        float mx = sT[0][0];
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = __builtin_fmaxf(mx, sT[kb][r]);
        {
            float a, b2;
            half_exchange(mx, a, b2);   // the other half-wave holds the other 32 keys of the same query
            mx = __builtin_fmaxf(a, b2);
        }
        if (t == 0 || __any(mx > RESCALE_THR)) {   // wave-uniform; per lane only rows that need it move their max
            const bool mv = (t == 0) || (mx > RESCALE_THR);
            const op_t m16 = (op_t)(m_run + mx);
            const float m_new = mv ? (float)m16 : m_run;
            const float delta = m_new - m_run;
            const float alpha = __builtin_amdgcn_exp2f(-delta);
            l_run *= alpha;
            m_run = m_new;
            if (hi == 0) qx[0] = (op_t)(-m_new);
#pragma unroll
            for (int db = 0; db < 2; ++db)
#pragma unroll
                for (int r = 0; r < 16; ++r) o[db][r] *= alpha;
#pragma unroll
            for (int kb = 0; kb < 2; ++kb)
#pragma unroll
                for (int r = 0; r < 16; ++r) sT[kb][r] -= delta;
        }
        opx8 pf[2][2];
        float ps0 = 0.0f, ps1 = 0.0f;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int r = 0; r < 16; r += 2) {
                opx2 pp;
                pp[0] = (op_t)__builtin_amdgcn_exp2f(sT[kb][r]);
                pp[1] = (op_t)__builtin_amdgcn_exp2f(sT[kb][r + 1]);
                pf[kb][r >> 3][r & 7] = pp[0];
                pf[kb][r >> 3][(r & 7) + 1] = pp[1];
                if (r & 2) ps1 = dot2_acc(pp, ps1);
                else ps0 = dot2_acc(pp, ps0);
            }
        }
        l_run += ps0 + ps1;

        // ---- O^T += V^T P^T --------------------------------------------------------------------
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
#pragma unroll
                for (int db = 0; db < 2; ++db) {
                    const opx4 lo = vlo[kb][s][db], hi4 = vhi[kb][s][db];
                    opx8 vf;
                    vf[0] = lo[0]; vf[1] = lo[1]; vf[2] = lo[2]; vf[3] = lo[3];
                    vf[4] = hi4[0]; vf[5] = hi4[1]; vf[6] = hi4[2]; vf[7] = hi4[3];
                    o[db] = mfma32(vf, pf[kb][s], o[db]);
                }
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        }  // wave_active

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    // ---- normalise and store: lane holds q = l31, d = 32db + crow32(r, hi) -----------------------
    float l_lo, l_hi;
    half_exchange(l_run, l_lo, l_hi);
    const float inv = 1.0f / (l_lo + l_hi);
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
