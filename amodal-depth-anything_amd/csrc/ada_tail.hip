// Fused DPT tail (see include/ada_hip.h: ada_dpt_tail_fwd):
//     out = act( sum_n relu( conv3x3( bilinear_ac(in) )[.., n] + bias[n] ) * tail_w[n] + tail_b )
// i.e. reference DA2/dpt.py:194-195 -- F.interpolate(out, (14 ph, 14 pw), bilinear, align_corners=True) followed by
// scratch.output_conv2 = Conv2d(C, 32, 3, padding 1) -> ReLU -> Conv2d(32, 1, 1) -> Sigmoid / ReLU / Identity -- in ONE kernel.
//
// The two-kernel path materialises the up-sampled map ("fin": [B, H+2, W+2, C] operand-typed, 2.2 GB at ViT-L bs=32), writes it
// once and re-reads it nine times through L2 from a 32-column GEMM whose A tile is all traffic and no reuse (0.8 ms + 1.3 ms).
// Here nothing but `in`, the weights and the output touches HBM / L2.
//
// Persistent launch, one 12-wave workgroup per CU walking tiles of 8 x 30 output pixels; a tile is processed as `passes` units of 64
// input channels.  The waves of a SIMD have different jobs, so that its VALU work and its matrix pipe overlap their latencies:
//   * waves 4-11 (producers, two per SIMD) build the unit's 10 x 32 halo tile in LDS, five rows per group of four waves: thread (halo
//     column, 8 channels) fetches the two source pixels of its column for the 5 source rows under its halo rows straight into registers
//     (fp32, two 16-byte pieces per pixel; the fetches of the next unit are re-issued row by row as this unit's rows are used up, so
//     they land under a whole unit of work), interpolates horizontally once per source row, then walks down its halo rows interpolating
//     vertically, and stores operand-typed 16-byte chunks -- zero outside the image (= the convolution's padding) -- XOR-swizzled (chunk ^ (column & 7)) so
//     that the consumers' fragment reads are bank-conflict free for every tap shift;
//   * waves 0-3 (consumers, 4 rows x 16 columns of the tile each) run the 3 x 3 convolution of the previous unit from the other halo
//     buffer: 6 (dx, k half) steps of 24 v_mfma_f32_16x16x32 (a tap is a constant row / column shift of the fragment address) against
//     the weights, which stay resident in LDS for the life of the workgroup; after the last unit of a tile: bias, ReLU, 32 -> 1 (DPP
//     row reduction), activation, store.
// One barrier per unit hands the buffers over.
#include <mutex>
#include "ada_common.h"
#ifndef TAIL_ABL
#define TAIL_ABL 0   // timing experiments: 1 no convolution, 2 no interpolation arithmetic, 4 no fetches
#endif

namespace {

constexpr int T_TH = 8, T_TW = 30;                 // output pixels per tile (the fragments cover 32 columns: two are discarded)
constexpr int T_HH = T_TH + 2, T_HW = 32;          // halo tile
constexpr int T_CH = 64;                           // channels per unit
constexpr int T_PIXB = T_CH * 2;                   // bytes per halo pixel (128)
constexpr int T_N = 32;                            // output channels of the 3x3 convolution
constexpr int T_WROW = 9 * T_PIXB;                 // bytes per weight row in LDS (9 taps x 64 channels)
constexpr int T_W_BYTES = T_N * T_WROW;            // 36864 per pass
constexpr int T_HALO_BYTES = T_HH * T_HW * T_PIXB + 2 * T_PIXB;   // 41216: the fragments of the two discarded columns read 2 pixels on
constexpr int T_MAX_PASSES = 2;                    // weights of all passes are LDS-resident: 2 * 41216 + 2 * 36864 = 156160 B
#ifndef TAIL_PRODUCER_WAVES
#define TAIL_PRODUCER_WAVES 8
#endif
constexpr int T_PW = TAIL_PRODUCER_WAVES;          // producer waves: 4 (one per SIMD, a thread builds all 10 halo rows of its column) or 8 (5 rows each)
constexpr int T_PH = T_PW / 4;                     // halves of the halo tile, one per group of 4 producer waves
constexpr int T_HROWS = T_HH / T_PH;               // halo rows per producer thread
constexpr int T_ROWS = T_PH == 1 ? 8 : 5;          // source rows fetched per thread: its halo rows must span <= T_ROWS - 1 row intervals
constexpr int T_INFLIGHT = 4 * (T_ROWS - 1);       // fetches that may be outstanding when a row is needed (see interpolate)
constexpr int T_THREADS = 256 + 64 * T_PW;
static_assert(T_PW == 4 || T_PW == 8, "4 or 8 producer waves");

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
    int ntx, nty, ntiles;
};

// sum over the 16 lanes of a DPP row, result in every lane (row_ror 8, 4, 2, 1)
ADA_DEV float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
    return v;
}

struct TileAt {
    int b, ty0, tx0;
};

__global__ __launch_bounds__(T_THREADS, 1) void dpt_tail_kernel(TailArgs p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const wbuf = smem + 2 * T_HALO_BYTES;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wave >= 4;
    const int l15 = lane & 15, q4 = lane >> 4;
    const int G = gridDim.x, bid = blockIdx.x;
    const int nunits = ((p.ntiles - bid + G - 1) / G) * p.passes;

    auto tile_of = [&](int u) -> TileAt {
        const int ti = bid + (u / p.passes) * G;
        const int per_img = p.ntx * p.nty;
        const int b = ti / per_img, rem = ti - b * per_img;
        const int tyi = rem / p.ntx;
        return TileAt{b, tyi * T_TH, (rem - tyi * p.ntx) * T_TW};
    };

    // ---- weights of every pass -> LDS, once: [pass][n][tap][64 channels], 16-byte chunks XOR-swizzled with (n >> 1) & 7 ----
    // k order inside a 64-channel block: chunk c = channels 4c..4c+3 and 32+4c..32+4c+3 (the same permutation on both MFMA operands).
    // A producer lane then fetches two 16-byte pieces that are contiguous with its neighbours': a fetch instruction touches 8 full
    // 128-byte lines instead of 16 half-used ones (the L1 moves whole lines: this halved the producers' fetch time).
    for (int q = tid; q < p.passes * (T_N * 72); q += T_THREADS) {   // chunks of 16 B: (pass, n, tap, chunk)
        const int pass = q / (T_N * 72), r0 = q - pass * (T_N * 72);
        const int n = r0 / 72, rem = r0 - n * 72, t = rem >> 3, c = rem & 7;
        const op_t* g = p.w + (long)n * (9 * p.cp) + t * p.cp + pass * T_CH + 4 * c;
        const u32x2 lo = *(const u32x2*)g, hi = *(const u32x2*)(g + 32);
        *(u32x4*)(wbuf + pass * T_W_BYTES + n * T_WROW + t * T_PIXB + ((c ^ ((n >> 1) & 7)) * 16)) = u32x4{lo[0], lo[1], hi[0], hi[1]};
    }

    // ---- producer state: thread (halo column hx, chunk pc) ----
    const int pt = tid & 255;
    const int half = T_PH == 2 ? __builtin_amdgcn_readfirstlane((tid - 256) >> 8) : 0;   // which T_HROWS rows of the halo tile
    const int hx = pt >> 3, pc = pt & 7;
    const unsigned pdst = (unsigned)(hx * T_PIXB + ((pc ^ (hx & 7)) * 16));
    f32x4 v[T_ROWS][4];
#if TAIL_ABL & 4
#pragma unroll
    for (int r = 0; r < T_ROWS; ++r)
#pragma unroll
        for (int i = 0; i < 4; ++i) v[r][i] = f32x4{0.f, 0.f, 0.f, 0.f};
#endif
    // v: source row r: pixel x0 (channels 4 pc .. 4 pc + 3 and 32 + 4 pc .. 32 + 4 pc + 3 of the unit), pixel x1 (same)
    struct Src {               // where a unit's source rows are: per-lane element offsets of the two pixels, first row, image
        long o0, o1;
        int py0, b;
    };
    auto src_of = [&](int u) -> Src {
        const TileAt t = tile_of(u);
        const int pass = u % p.passes;
        int x = t.tx0 - 1 + hx;
        x = x < 0 ? 0 : (x > p.wo - 1 ? p.wo - 1 : x);
        const int x0 = (int)(p.sx * (float)x);
        const int x1 = x0 + (x0 < p.wi - 1 ? 1 : 0);
        const int yf = t.ty0 - 1 + half * T_HROWS;
        const int yv = yf < 0 ? 0 : (yf > p.ho - 1 ? p.ho - 1 : yf);
        return Src{(long)x0 * p.ld_in + pass * T_CH + pc * 4, (long)x1 * p.ld_in + pass * T_CH + pc * 4, (int)(p.sy * (float)yv), t.b};
    };
    // The fetches are inline assembly with hand-counted s_waitcnt: written as C++ loads the compiler gives the rows of the next unit fresh
    // registers and copies them into place at the loop's back edge -- behind an "s_waitcnt vmcnt(0)" that puts the whole fetch latency
    // back on the critical path.  Every wait names the row's registers as in/out operands, so no use of a row can be scheduled above it.
    auto fetch_row = [&](const Src& q, int r) {
        const int gy = q.py0 + r < p.hi - 1 ? q.py0 + r : p.hi - 1;
        const float* row = p.in + ((long)q.b * p.hi + gy) * p.wi * p.ld_in;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[r][0]) : "v"(row + q.o0));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[r][1]) : "v"(row + q.o0 + 32));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[r][2]) : "v"(row + q.o1));
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[r][3]) : "v"(row + q.o1 + 32));
    };
#define TAIL_WAIT_ROW(r) asm volatile("s_waitcnt vmcnt(%4)" : "+v"(v[r][0]), "+v"(v[r][1]), "+v"(v[r][2]), "+v"(v[r][3]) : "n"(T_INFLIGHT))
    // Builds unit u's halo tile from the rows in v and, row by row as they are used up, re-issues the fetches of unit u + 1 into the same
    // registers: every row then has a whole unit of interpolation arithmetic plus the barrier wait to land in.  (Fetching all of unit u + 1
    // after unit u is done leaves only the barrier wait: ~2 us of exposed latency per unit, profiles/r03_p_fused_tail.txt.)
    auto interpolate = [&](int u, bool more) {
        const TileAt t = tile_of(u);
        const Src nx = src_of(more ? u + 1 : u);
        char* const dst = smem + (u & 1) * T_HALO_BYTES + pdst;
        const int x = t.tx0 - 1 + hx;
        const bool xin = x >= 0 && x < p.wo;
        const int xc = x < 0 ? 0 : (x > p.wo - 1 ? p.wo - 1 : x);
        const float fx = p.sx * (float)xc;
        const float lx1 = fx - (float)(int)fx, lx0 = 1.0f - lx1;
        const int yf = t.ty0 - 1 + half * T_HROWS;
        const int yv = yf < 0 ? 0 : (yf > p.ho - 1 ? p.ho - 1 : yf);
        const int py0 = (int)(p.sy * (float)yv);
        // horizontal interpolation of source row r (computed when the walk below reaches it: the fetched pixels die as they are used)
        // Fetches retire in issue order.  When row r of this unit is needed, the rows after it (4 fetches each) and the rows 0 .. r - 1 of the
        // next unit, issued since, may still be in flight: always 4 (T_ROWS - 1) fetches -- the last unit of a workgroup fetches itself again rather
        // than take a second code path (two paths make the row registers phi nodes, which the compiler copies while they are in flight).
        auto hrow = [&](int r, float* h) {
            h[0] = lx0 * v[r][0][0] + lx1 * v[r][2][0]; h[1] = lx0 * v[r][0][1] + lx1 * v[r][2][1];
            h[2] = lx0 * v[r][0][2] + lx1 * v[r][2][2]; h[3] = lx0 * v[r][0][3] + lx1 * v[r][2][3];
            h[4] = lx0 * v[r][1][0] + lx1 * v[r][3][0]; h[5] = lx0 * v[r][1][1] + lx1 * v[r][3][1];
            h[6] = lx0 * v[r][1][2] + lx1 * v[r][3][2]; h[7] = lx0 * v[r][1][3] + lx1 * v[r][3][3];
        };
        opx8 zero;
#pragma unroll
        for (int e = 0; e < 8; ++e) zero[e] = (op_t)0.0f;
        int hy = half * T_HROWS;
        const int hy_end = hy + T_HROWS;
        if (yf < 0) {   // the row above the image
            *(opx8*)dst = zero;
            hy = 1;
        }
        float ha[8], hb[8];
#if !(TAIL_ABL & 4)
        TAIL_WAIT_ROW(0);
#endif
        hrow(0, hb);
        if (!(TAIL_ABL & 4)) fetch_row(nx, 0);
#pragma unroll
        for (int r = 0; r < T_ROWS - 1; ++r) {
#pragma unroll
            for (int e = 0; e < 8; ++e) ha[e] = hb[e];
#if !(TAIL_ABL & 4)
            TAIL_WAIT_ROW(r + 1);
#endif
            hrow(r + 1, hb);
            if (!(TAIL_ABL & 4)) fetch_row(nx, r + 1);
            while (hy < hy_end) {   // halo rows whose upper source row is py0 + r (wave-uniform: 0, 1 or 2 of them when up-sampling)
                const int y = t.ty0 - 1 + hy;
                if (y >= p.ho) break;
                const float fy = p.sy * (float)y;
                const int y0 = (int)fy;
                if (y0 != py0 + r) break;
                const float ly1 = fy - (float)y0, ly0 = 1.0f - ly1;
                opx8 o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = to_op(ly0 * ha[e] + ly1 * hb[e]);
                *(opx8*)(dst + hy * (T_HW * T_PIXB)) = xin ? o : zero;
                ++hy;
            }
        }
        for (; hy < hy_end; ++hy) *(opx8*)(dst + hy * (T_HW * T_PIXB)) = zero;   // rows below the image
    };

    // ---- consumer state: wave w owns output rows 4 (w >> 1) .. + 3 and the column half w & 1 of the tile; fragment f = row in the group ----
    // A rows are halo pixels (4 (w >> 1) + f + dy, 16 (w & 1) + dx + l15); the 16-byte chunk c of a pixel is stored at chunk c ^ (column & 7).
    // A ds_read_b128 is serviced in the lane groups {0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... (MI355X_MICROARCH.md, LDS): a group is 16
    // consecutive pixels, eight of them reading chunk c and eight chunk c ^ 1, and pixels l and l + 8 always differ in which.  Same-parity
    // columns have same-parity keys, so (column & 1, chunk ^ key) is distinct over the group for ANY starting column.  (The GEMM tiles'
    // key (row >> 1) & 7 is conflict-free only for windows that start on a multiple of 16: with it the dx = 1, 2 taps here were 2-way
    // conflicts, 25 % of all LDS cycles of the kernel -- SQ_LDS_BANK_CONFLICT 28.8 M -> 0 per launch, profiles/r03_p_fused_tail.txt.)  The kernel is bound by these fragment
    // reads (a 32-column GEMM: every MFMA wants a fresh 1 KB), so a wave walks (dx, k half) outermost and reads the six halo rows its
    // four output rows touch ONCE for all three dy taps: 6 A + 6 B fragments per 24 MFMAs (rows 2 w, 2 w + 1 with dy outermost: 18 per 24).
    const int rg = (wave & 3) >> 1, ch = wave & 1;
    unsigned abase[3][2], bbase[2][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ax = 16 * ch + dx + l15;
            abase[dx][k] = (unsigned)((4 * rg * T_HW + ax) * T_PIXB + (((4 * k + q4) ^ (ax & 7)) * 16));
        }
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int n = 16 * j + l15;
            bbase[j][k] = (unsigned)(2 * T_HALO_BYTES + n * T_WROW + (((4 * k + q4) ^ ((n >> 1) & 7)) * 16));
        }
    f32x4 acc[4][2];
    float b0 = 0.f, b1 = 0.f, w0 = 0.f, w1 = 0.f;   // fetched by the consumers only: a pending load in a producer wave would be waited for with vmcnt(0) inside its loop

    auto convolve = [&](int u) {
        const int pass = u % p.passes;
        if (pass == 0) {
#pragma unroll
            for (int f = 0; f < 4; ++f)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[f][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        const char* const ha = smem + (u & 1) * T_HALO_BYTES;
        const char* const wb = smem + pass * T_W_BYTES;
        // step i = (dx, k half).  The 12 fragment reads of step i + 1 are slotted one behind every second MFMA of step i: left alone the compiler
        // reads each fragment one or two MFMAs before its use, and the one MFMA-issuing wave of the SIMD sits out the LDS latency some 36 times
        // per unit.  (Consumers alone, per launch at ViT-L bs=32: 0.82 ms with rows 2w / 2w+1 and dy outermost, 0.75 ms with the six-row reuse,
        // 0.69 ms with the reads of the next step in one block ahead of the MFMAs, 0.66 ms interleaved -- profiles/r03_p_fused_tail.txt.)
        opx8 af[2][6], bf[2][3][2];
        auto read_step = [&](int i, opx8* a, opx8 (*bq)[2]) {
            const int dx = i >> 1, k = i & 1;
#pragma unroll
            for (int hr = 0; hr < 6; ++hr) a[hr] = *(const opx8*)(ha + abase[dx][k] + hr * (T_HW * T_PIXB));
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int j = 0; j < 2; ++j) bq[dy][j] = *(const opx8*)(wb + bbase[j][k] + (dy * 3 + dx) * T_PIXB);
        };
        read_step(0, af[0], bf[0]);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i + 1 < 6) read_step(i + 1, af[(i + 1) & 1], bf[(i + 1) & 1]);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int f = 0; f < 4; ++f)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[f][j] = mfma16(af[i & 1][f + dy], bf[i & 1][dy][j], acc[f][j]);
            if (i + 1 < 6) {
#pragma unroll
                for (int q = 0; q < 12; ++q) {   // one read of step i + 1 in the shadow of every second MFMA of step i
                    __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (pass != p.passes - 1) return;
        // bias, ReLU, 32 -> 1, activation.  D[4 * q4 + r][l15]: row = pixel inside the 16-pixel fragment, column = output channel
        const TileAt t = tile_of(u);
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            float d[4];
#pragma unroll
            for (int r = 0; r < 4; ++r)
                d[r] = row16_sum(__builtin_fmaxf(acc[f][0][r] + b0, 0.0f) * w0 + __builtin_fmaxf(acc[f][1][r] + b1, 0.0f) * w1) + p.tail_b;
            const int y = t.ty0 + 4 * rg + f;
            if (l15 < 4 && y < p.ho) {   // lane l15 = r stores pixel r of its quarter: four consecutive floats per 16-lane row
                const int cx = 16 * ch + 4 * q4 + l15, x = t.tx0 + cx;
                float o = l15 == 0 ? d[0] : (l15 == 1 ? d[1] : (l15 == 2 ? d[2] : d[3]));
                if (p.tail_act == ADA_ACT_SIGMOID) o = 1.0f / (1.0f + __expf(-o));
                else if (p.tail_act == ADA_ACT_RELU) o = __builtin_fmaxf(o, 0.0f);
                if (cx < T_TW && x < p.wo) p.out[((long)t.b * p.ho + y) * p.wo + x] = o;
            }
        }
    };

    // ---- the pipeline: during step u the producers build unit u while the consumers convolve unit u - 1 ----
    // Two loops with the same number of barriers (the branch is wave-uniform; s_barrier counts waves, whichever loop they are in): the
    // producers' 80 registers of pixels in flight are not live in the consumers' code and the other way round.
    if (producer) {
        if (nunits > 0 && !(TAIL_ABL & 4)) {
            const Src q = src_of(0);
#pragma unroll
            for (int r = 0; r < T_ROWS; ++r) fetch_row(q, r);
        }
        for (int u = 0; u <= nunits; ++u) {
            if (u < nunits && !(TAIL_ABL & 2)) interpolate(u, u + 1 < nunits);
            __syncthreads();
        }
        asm volatile("s_waitcnt vmcnt(0)");   // the last unit's spare fetches
    } else {
        const float* bias = p.bias;
        const float* tail_w = p.tail_w;
        asm volatile("" : "+s"(bias), "+s"(tail_w));   // keeps the four loads in this branch ...
        b0 = bias[l15]; b1 = bias[16 + l15]; w0 = tail_w[l15]; w1 = tail_w[16 + l15];
        // ... and waited for here: the structurised CFG has a (never taken) path from this branch into the producers' one, and a load the compiler
        // believes pending there costs an "s_waitcnt vmcnt(0)" in the middle of the producers' loop, i.e. their whole fetch latency, every unit
        asm volatile("" ::"v"(b0), "v"(b1), "v"(w0), "v"(w1));
        for (int u = 0; u <= nunits; ++u) {
            if (u >= 1 && !(TAIL_ABL & 1)) convolve(u - 1);
            __syncthreads();
        }
    }
}

int g_tail_cus = 0;

}  // namespace

extern "C" int ada_dpt_tail_fwd(const float* in, int64_t ld_in, int32_t batch, int32_t hi, int32_t wi, int32_t ho, int32_t wo, int32_t cp,
                                const void* w, const float* bias, const float* tail_w, float tail_b, int32_t tail_act, float* out,
                                void* stream) {
    ADA_REQUIRE(in && w && bias && tail_w && out, ADA_EINVAL, "ada_dpt_tail_fwd: null pointer");
    ADA_REQUIRE(batch > 0 && hi > 0 && wi > 0 && ho > 0 && wo > 0, ADA_EINVAL, "ada_dpt_tail_fwd: bad shape");
    ADA_REQUIRE(cp > 0 && cp % T_CH == 0 && cp / T_CH <= T_MAX_PASSES && ld_in >= cp && ld_in % 4 == 0, ADA_EUNSUPPORTED,
                "ada_dpt_tail_fwd: the padded channel count (%d) must be 64 or 128 and fit the input row (ld_in=%ld)", cp, (long)ld_in);
    ADA_REQUIRE(((uintptr_t)in % 16) == 0 && ((uintptr_t)w % 16) == 0, ADA_EINVAL, "ada_dpt_tail_fwd: in / w must be 16-byte aligned");
    TailArgs p;
    p.in = in; p.ld_in = ld_in; p.batch = batch; p.hi = hi; p.wi = wi; p.ho = ho; p.wo = wo; p.passes = cp / T_CH;
    p.sy = ho > 1 ? (float)(hi - 1) / (float)(ho - 1) : 0.0f;
    p.sx = wo > 1 ? (float)(wi - 1) / (float)(wo - 1) : 0.0f;
    // a producer thread holds T_ROWS source rows in registers: its T_HROWS halo rows may span at most T_ROWS - 1 source row intervals.  The
    // limit is stated for the widest variant (10 rows within 8) so that it does not depend on the build: an up-sampling by at least 1.5
    ADA_REQUIRE((int)(p.sy * (float)(T_HH - 1)) + 2 <= 7 && (int)(p.sy * (float)(T_HROWS - 1)) + 2 <= T_ROWS - 1, ADA_EUNSUPPORTED,
                "ada_dpt_tail_fwd: vertical scale %d -> %d is not supported (needs ho >= 1.5 hi; the model's ratio is 14 / 8)", hi, ho);
    p.w = (const op_t*)w; p.cp = cp; p.bias = bias; p.tail_w = tail_w; p.tail_b = tail_b; p.tail_act = tail_act; p.out = out;
    p.ntx = (wo + T_TW - 1) / T_TW;
    p.nty = (ho + T_TH - 1) / T_TH;
    const long ntiles = (long)p.ntx * p.nty * batch;
    ADA_REQUIRE(ntiles < (1L << 30), ADA_EUNSUPPORTED, "ada_dpt_tail_fwd: too many tiles");
    p.ntiles = (int)ntiles;
    static std::once_flag once;
    std::call_once(once, []() {
        if (hipFuncSetAttribute((const void*)dpt_tail_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) (void)hipGetLastError();
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) g_tail_cus = n;
        if (g_tail_cus <= 0) g_tail_cus = 256;
        (void)hipGetLastError();
    });
    const size_t smem = 2 * (size_t)T_HALO_BYTES + (size_t)p.passes * T_W_BYTES;
    const int grid = p.ntiles < g_tail_cus ? p.ntiles : g_tail_cus;
    hipLaunchKernelGGL(dpt_tail_kernel, dim3(grid), dim3(T_THREADS), smem, (hipStream_t)stream, p);
    return ada_check_launch("ada_dpt_tail_fwd");
}
