/*
 * ada_hip.h -- C ABI of libada_hip.so, the MI355X (gfx950) kernel library behind the
 * Amodal-Depth-Anything forward pass.
 *
 * The reference (zhyever/Amodal-Depth-Anything) has no FFI of its own: its lowest boundary is
 * the torch.nn / ATen call made by each L1 module.  Every entry point below
 * therefore names the reference call sites (file:line under /root/reference, DA2 =
 * src/models/amodalsynthdrive/depth_anything_v2) whose arithmetic it replaces.
 *
 * Conventions
 *  - plain C, raw device pointers, explicit sizes/strides; no torch types.
 *  - the caller owns every buffer (inputs, outputs, workspaces); the library allocates nothing.
 *    State kept by the library: a thread-local last-error string, a thread-local "last tile" code,
 *    and the process-global tuning switches of the ada_debug_* hooks at the end of this header
 *    (defaults = the shipped configuration; they select between kernels that all compute the same
 *    result, so a caller that never touches them sees a stateless library).
 *  - every launcher is asynchronous on `stream` (a hipStream_t passed as void*), re-entrant,
 *    and returns 0 on success or a negative ADA_E* code; it never throws or exits.
 *  - "op" = the contraction-operand type the library was built for: IEEE fp16 by default
 *    (ada_operand_dtype() == ADA_DT_F16), bf16 with -DADA_OPERAND_BF16.  All accumulation,
 *    normalisation, softmax, activation and residual arithmetic is fp32.
 *  - activations of the DPT head are NHWC ("pixel rows"): row = (b*H + y)*W + x, ld = channels
 *    rounded up to a multiple of 64 (pad columns are zero).  "padded NHWC" additionally has a
 *    one-pixel zero border: [B, H+2, W+2, ld].
 */
#ifndef ADA_HIP_H
#define ADA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADA_ABI_VERSION 8   /* 8 (round 6): the default-off experiment paths that lost are gone -- ada_igemm_args without ln_stats / ln_colsum / rowstat_out (EP_LNFOLD / EP_ROWSTATS) and the LayerNorm tail (ln_*), no ada_rowstats_finalize.  History: 6 (round 5): + ada_depth_stats_fwd, ada_token_diversity_fwd; 7: ada_igemm_args grows f8_from / f8_mid / f8_scales at its end and split_seg < 0 names the fp8 form of a split output (a zero-filled tail = off: every ABI-6 call means what it meant) */

/* status codes */
#define ADA_OK 0
#define ADA_EINVAL (-1)      /* bad argument (null pointer, negative size, misaligned ld ...) */
#define ADA_EUNSUPPORTED (-2) /* shape outside what the kernels implement */
#define ADA_ELAUNCH (-3)     /* HIP reported a launch failure */

/* dtypes */
#define ADA_DT_F32 0
#define ADA_DT_F16 1
#define ADA_DT_BF16 2

int ada_abi_version(void);
/* ADA_DT_F16 or ADA_DT_BF16: the operand type this build of the library contracts in. */
int ada_operand_dtype(void);
/* Text of the last error raised on the calling thread ("" if none). */
const char* ada_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM contraction with fused epilogue:   C[m, n] = sum_k A(m, k) * W[n, k]
 *
 * One MFMA kernel serves every dense contraction of the path:
 *   nn.Linear            DA2/dinov2_layers/attention.py:51,60 (qkv, proj); mlp.py:36,39 (fc1, fc2);
 *                        swiglu_ffn.py:30,33 (w12, w3)
 *   Conv2d 14x14 s14     DA2/dinov2_layers/patch_embed.py:66,76 after ada_patchify (a_mode PLAIN)
 *   Conv2d 1x1           DA2/dpt.py:172 (projects), DA2/util/blocks.py:146 (out_conv), dpt.py:149
 *   ConvTranspose2d k=s  DA2/dpt.py:88-100,173 (a_mode PLAIN + o_map SHUFFLE)
 *   Conv2d 3x3 s1/s2 p1  DA2/dpt.py:101-106,156,135,147; DA2/util/blocks.py:20-24,45-47 (a_mode CONV3)
 * with epilogues
 *   + bias                                  (every nn.Linear / Conv2d bias)
 *   GELU (exact erf)                        DA2/dinov2_layers/mlp.py:37
 *   * gamma + residual                      layer_scale.py:28 + block.py:105-106
 *   + residual                              DA2/util/blocks.py:80 (skip_add), dinov2.py:246 (pos_embed)
 *   ReLU on the operand copy                DA2/util/blocks.py:67,72 (activation before conv1/conv2)
 *   SiLU(x1)*x2                             swiglu_ffn.py:31-32 (columns interleaved by the packer)
 *   ReLU -> 1x1 conv (N -> 1) -> Sigmoid/ReLU   DA2/dpt.py:146-151 / RAW dpt.py:109-115 (tail)
 * ---------------------------------------------------------------------------------------- */

/* a_mode */
#define ADA_A_PLAIN 0 /* A(m, k) = A[m * lda + k] */
#define ADA_A_CONV3 1 /* 3x3 window gather from padded NHWC, k = (dy*3+dx)*lda + c */

/* output row maps: GEMM row m -> row of the output buffer */
#define ADA_MAP_PLAIN 0   /* row = m */
#define ADA_MAP_PAD 1     /* m=(b,y,x) in [B,Ho,Wo] -> interior of padded NHWC [B,Ho+2,Wo+2] */
#define ADA_MAP_TOKEN 2   /* m=(b,p) in [B,Np] -> token row b*(Np+1) + 1 + p (skips the cls row) */
#define ADA_MAP_SHUFFLE 3 /* ConvTranspose k=s: m=(b,y,x) in [B,Ho,Wo], n=(i,j,co) ->
                             padded NHWC [B, s*Ho+2, s*Wo+2], pixel (s*y+i, s*x+j), column co */

/* epilogue flags */
#define ADA_EP_BIAS 0x1
#define ADA_EP_GELU 0x2
#define ADA_EP_GAMMA 0x4     /* v *= gamma[n] (LayerScale) */
#define ADA_EP_RESIDUAL 0x8  /* v += res[res_row(m) * ldr + n] (fp32) */
#define ADA_EP_RELU_OP 0x10  /* ReLU applied to the operand-typed copy only */
#define ADA_EP_SWIGLU 0x20   /* columns come in (x1, x2) 32-wide groups: out = silu(x1) * x2 */
#define ADA_EP_TAIL 0x40     /* v = relu(v); d = sum_n v*tail_w[n] + tail_b; out_f32[m] = act(d) */
#define ADA_EP_RELU_F32 0x80 /* ReLU applied to the fp32 output as well */
/* (0x100, 0x200: ADA_EP_ROWSTATS / ADA_EP_LNFOLD of ABI <= 7 -- the LayerNorm folded into its consumer, a measured loss -- are gone; the bits stay unused) */

/* tail activations */
#define ADA_ACT_NONE 0
#define ADA_ACT_SIGMOID 1
#define ADA_ACT_RELU 2

typedef struct ada_igemm_args {
    int32_t M, N, K;        /* K = padded contraction length (multiple of 64); for CONV3, K = 9*lda */
    int32_t a_mode;
    const void* A;          /* op-typed */
    int64_t lda;            /* elements; multiple of 64 */
    /* CONV3 geometry: output grid [B, Ho, Wo]; input padded NHWC [B, Hp, Wp, lda]; stride 1 or 2 */
    int32_t Ho, Wo, Hp, Wp, stride;
    const void* W;          /* op-typed [N, K] row-major, K contiguous, zero padded */
    const float* bias;      /* [N] or NULL */
    const float* gamma;     /* [N] or NULL */
    const float* res;       /* fp32 residual or NULL */
    int64_t ldr;
    int32_t res_row_mod;    /* >0: res row = (m % res_row_mod) + res_row_off; 0: res row = out_f32 row */
    int32_t res_row_off;
    int32_t flags;          /* ADA_EP_* */
    float* out_f32;         /* fp32 output or NULL (may alias res) */
    int64_t ldo_f32;
    int32_t map_f32;        /* ADA_MAP_PLAIN / TOKEN */
    void* out_op;           /* op-typed output or NULL */
    int64_t ldo_op;
    int32_t map_op;         /* ADA_MAP_* */
    int32_t map_h, map_w;   /* Ho, Wo (PAD / SHUFFLE) or Np in map_h (TOKEN) */
    int32_t shuffle_s;      /* SHUFFLE: kernel = stride s; N = s*s*shuffle_c */
    int32_t shuffle_c;
    const float* tail_w;    /* TAIL: [N] fp32 */
    float tail_b;
    int32_t tail_act;
    int32_t split_seg;      /* > 0: split-precision op output -- out_op receives [hi | lo] in two column segments of
                               split_seg elements (hi = round(v), lo = round(v - hi)); 0 = plain;
                               < 0: the fp8 form of the same, seg = -split_seg: [hi: seg elements | lo8: seg bytes | hi8: seg bytes], see f8_from */
    int32_t a_dup_seg;      /* > 0: the A operand is such a split tensor: [hi | lo] segments of a_dup_seg elements per row (per tap of a
                               3x3 conv) and the contraction runs over THREE segments (hi, lo, hi -- the third re-reads the first) against
                               weights packed [w_hi | w_hi | w_lo]: x_hi w_hi + x_lo w_hi + x_hi w_lo.  K = 3 * a_dup_seg (x 9 for CONV3),
                               lda >= 2 * a_dup_seg.  0 = plain operand */
    int32_t tap_cols;       /* CONV3, > 0: sub-pixel convolution.  A stride-s transposed convolution (DA2/dpt.py:88-100 resize_layers[0/1]) followed,
                               with nothing in between, by a 3x3 convolution (DA2/dpt.py:153-159 input_projection[i][0]) is ONE 3x3 convolution
                               over the COARSE grid with s*s*Cout output columns: column block p = py*s + px (tap_cols = Cout columns each) is
                               output phase (py, px) and touches only the coarse taps its 3x3 fine window reaches -- 1, 2 or 4 of the 9.  The
                               weights are composed by the caller ([N, 9*lda] tap-major as for any CONV3, zero in the untouched blocks) and
                               tap_mask[p] (bit 3*(dy+1) + (dx+1)) names the touched taps; the kernel's k-loop for an N-tile walks the union of
                               its blocks' taps only: 36 C^2 MACs per coarse pixel instead of 160 C^2 for s = 4.  N / tap_cols <= 16.  0 = off */
    uint16_t tap_mask[16];
    int32_t a_wrap;         /* PLAIN, > 0: weight-only split precision.  The A row holds a_wrap operand-typed elements and the contraction runs over
                               K = 2 * a_wrap with the A walk starting over at k = a_wrap, against weights packed [w_hi | w_lo] (w_hi = round(w),
                               w_lo = round(w - w_hi)):  x w_hi + x w_lo  -- the weight's rounding error is gone, the activation's stays
                               (used where the activation is produced in the operand type anyway: attention output, SwiGLU hidden).  0 = off */
    int32_t bias_row_mod;   /* > 0: one bias vector per group of bias_row_mod consecutive rows: row m uses bias[(m / bias_row_mod) * N + n].  The
                               use_clstoken read-out (DA2/dpt.py:110-117,164-167: Linear(2D, D) on [patch | class token] + GELU) is a D -> D GEMM over
                               the patch tokens whose bias W_cls cls_b + b differs per image: one launch for the whole batch.  Operand-typed output
                               only (bias / GELU epilogues).  0 = one bias vector [N] */
    /* fp8 correction terms of a split-precision product (gfx950 v_mfma_scale_f32_16x16x128_f8f6f4, twice the fp16 matrix rate):
         x w  ~  x_hi w_hi  +  2^-10 x_lo8 w_hi8  +  x_hi8 w_lo8          (instead of three fp16 products, a_dup_seg)
       f8_from > 0: inside each period of the k-walk -- all K operand slots of a PLAIN operand, one tap (lda slots) of a CONV3 one -- the slots from
       f8_from on hold BYTES, two codes per slot: e5m2 for A, e4m3 for W, contracted 128 codes per k-step into the same fp32 accumulators.
       Slots [f8_from, f8_mid) use the scale pair in bits 0-15 of f8_scales (A byte, then W byte), [f8_mid, period) the pair in bits 16-31; a scale
       byte e multiplies its operand by 2^(e - 127) (E8M0).  The A layout [hi: seg slots | lo8: seg bytes | hi8: seg bytes] is what a producer writes
       for split_seg = -seg (lo8 = e5m2((v - hi) 2^10), hi8 = e5m2(v)); the weights are packed [w_hi | w_hi8 | w_lo8] with per-tensor power-of-two
       scales: K = 2 seg (x 9), f8_from = seg, f8_mid = 3 seg / 2, f8_scales = 117 | sb_hi << 8 | 127 << 16 | sb_lo << 24.  f8_from, f8_mid: multiples
       of 64.  Excludes a_dup_seg / a_wrap.  0 = off */
    int32_t f8_from, f8_mid;
    uint32_t f8_scales;
} ada_igemm_args;

int ada_igemm(const ada_igemm_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused scaled-dot-product attention, head_dim 64 (all of ViT-S/B/L/G):
 *   softmax(q k^T) v per (batch, head).  q must arrive pre-scaled by head_dim^-0.5 * log2(e) (folded into
 *   the qkv weights by the packer): the kernel evaluates the softmax in base 2, softmax_e(s) = softmax_2(s log2 e).  Replaces DA2/dinov2_layers/attention.py:53-59 (and the xformers
 *   memory_efficient_attention call at :76).  qkv is the packed output of the qkv linear:
 *   [B*N, 3*heads*64] with column = which*heads*64 + head*64 + d (attention.py:51 reshape).
 *   out: [B*N, heads*64] op-typed (the "transpose(1,2).reshape(B,N,C)" layout of :59).
 *   The N x N score matrix is never materialised: K/V tiles of 64 keys are staged through LDS,
 *   softmax runs online in fp32 registers.
 * ---------------------------------------------------------------------------------------- */
int ada_attention_fwd(const void* qkv, void* out, int32_t batch, int32_t n_tokens, int32_t heads,
                      void* stream);
/* ABI 8: the same with a row stride for `out` (ld_out elements; 0 = heads * 64) and a split-precision output for the blocks whose linear layers run in split
 * precision -- the attention output feeds attn.proj (attention.py:60) and exists in the operand type only.  split_seg > 0: row = [hi | lo], lo = round(v - hi) at
 * column + split_seg;  split_seg < 0, seg = -split_seg: row = [hi: seg elements | lo8: seg bytes | hi8: seg bytes] (e5m2((v - hi) 2^10), e5m2(v)) -- the forms
 * ada_igemm reads through a_dup_seg / f8_from.  |split_seg| >= heads * 64, ld_out >= 2 |split_seg|.  0 = the plain row. */
int ada_attention_ex(const void* qkv, void* out, int32_t batch, int32_t n_tokens, int32_t heads, int64_t ld_out, int32_t split_seg,
                     void* stream);

/* ------------------------------------------------------------------------------------------
 * LayerNorm over the last dimension, eps inside the sqrt, biased variance, fp32 statistics:
 *   nn.LayerNorm(D, eps=1e-6)   DA2/dinov2.py:96,185 used at block.py:84,87 and dinov2.py:337-338
 *   channels-first LayerNorm    DA2/dpt.py:55-61 (on NHWC rows it is the same row-wise op)
 * in: fp32 [rows, ld_in]; normalises the first `dim` columns of each row.
 * Row selection: group_in > 0 treats the input as groups of `group_in` rows and skips the first
 * `skip` rows of each group (drops the cls token: dinov2.py:339-340); output rows are compacted.
 * Output (either may be NULL): op-typed with row map (PLAIN or PAD) and optional ReLU
 * (DA2/dpt.py:158), and/or fp32 plain.  split_seg > 0 writes the op-typed output in split precision
 * ([hi | lo] column segments, see ada_igemm_args.split_seg).
 * ---------------------------------------------------------------------------------------- */
int ada_layernorm_fwd(const float* in, int64_t ld_in, int32_t rows_out, int32_t dim,
                      int32_t group_in, int32_t skip,
                      const float* weight, const float* bias, float eps,
                      void* out_op, int64_t ld_op, int32_t map_op, int32_t map_h, int32_t map_w,
                      int32_t relu,
                      float* out_f32, int64_t ld_f32, int32_t split_seg, void* stream);

/* The same kernel with its two extensions (struct arguments; zero-initialise what is not used):
 *  - a SECOND operand-typed output of the same rows with the same statistics but its own gain / bias: the four tap LayerNorms of
 *    get_intermediate_layers (DA2/dinov2.py:337-340: shared final norm, cls token dropped) read the very rows the next block's norm1
 *    (block.py:84) reads, so one pass emits both -- rows are taken in groups of out2_group, the first out2_skip rows of a group are
 *    left out of the second output and the others compacted;
 *  - sub-pixel input (unshuffle_s = s > 0): `in` is the [coarse pixel, s*s*dim] fp32 output of a sub-pixel convolution
 *    (ada_igemm_args.tap_cols) and output row r is FINE pixel (b, Y, X) of the map_h x map_w grid, whose values are columns
 *    [(Y%s * s + X%s) * dim, +dim) of coarse pixel (b, Y/s, X/s) -- the channels-first LayerNorm + ReLU of input_projection
 *    (DA2/dpt.py:153-159) reads the merged convolution's output in place.  tap_bias [s*s*dim, 9] (optional) holds, per output column
 *    and coarse tap, the transposed convolution's bias as seen through the 3x3 filter; on the outermost ring of the fine grid the taps
 *    that fall outside the image are subtracted (zero padding applies to the transposed conv's OUTPUT, bias included). */
typedef struct ada_layernorm_args {
    const float* in;
    int64_t ld_in;
    int32_t rows_out, dim, group_in, skip;
    const float* weight;
    const float* bias;
    float eps;
    void* out_op;
    int64_t ld_op;
    int32_t map_op, map_h, map_w, relu;
    float* out_f32;
    int64_t ld_f32;
    int32_t split_seg;
    const float* weight2;
    const float* bias2;
    void* out2_op;
    int64_t ld2_op;
    int32_t out2_group, out2_skip, split_seg2;
    int32_t unshuffle_s;
    const float* tap_bias;
    int32_t identity;       /* 1: no normalisation, y = x (weight / bias may be NULL): with unshuffle_s the kernel is the re-layout pass behind a sub-pixel
                               convolution whose consumer is not a LayerNorm (raw head: resize_layers[i] + layerN_rn).  relu: 0 none, 1 both outputs,
                               2 the operand-typed output only (the fp32 copy is the pre-activation the ResidualConvUnit adds back, util/blocks.py:57-80) */
} ada_layernorm_args;
int ada_layernorm_ex(const ada_layernorm_args* args, void* stream);

/* ------------------------------------------------------------------------------------------
 * 3x3 convolution of an align-corners bilinear up-sampling with the channel mixing moved in front of the resize:
 *   scratch.output_conv1( F.interpolate(refinenet1 output, x2, bilinear, align_corners=True) )      DA2/dpt.py:192-193, util/blocks.py:144-146
 * A 1x1 channel mix commutes with a per-channel resample, so
 *   conv3x3(resize(z))[Y, X, co] = b[co] + sum over the taps t = (dy, dx) whose position (Y + dy - 1, X + dx - 1) lies inside the fine grid of
 *                                  resize(W_t z)[Y + dy - 1, X + dx - 1, co]
 * in: [B * hi * wi, ld_in], fp32 or operand-typed (in_dtype = ADA_DT_F32 / the library's operand type), column t * channels + co = (W_t z)[co] on the COARSE grid (one ada_igemm with N = 9 * channels; the
 *     caller composes W_t with whatever 1x1 convolution precedes the resize -- refinenet1.out_conv -- and puts that convolution's bias, seen
 *     through W_t, into the GEMM's bias: a tap that falls into the zero padding then drops out whole, as it must).
 * out: fp32 [B * ho * wo, ld_out].  channels in {32, 64, 128}; an up-sampling is expected (the patch under an 8 x 16 tile must fit LDS).
 * A quarter of the 3x3 convolution's MACs on the 2x finer grid, and the up-sampled operand map is never written.
 * ---------------------------------------------------------------------------------------- */
int ada_tapsum_resize_fwd(const void* in, int32_t in_dtype, int64_t ld_in, int32_t batch, int32_t hi, int32_t wi, int32_t ho, int32_t wo, int32_t channels,
                          const float* bias, float* out, int64_t ld_out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Patchify: the im2col half of the 14x14/stride-14 patch-embed convolutions
 * (DA2/dinov2_layers/patch_embed.py:76 for RGB, DA2/dinov2.py:239 for the guidance embed) with
 * the ImageNet normalisation of src/models/amodalsynthdrive/dav2.py:65 fused in.
 * x: fp32 NCHW [B,3,H,W]; guide: fp32 NCHW [B,Cg,H,W] or NULL (Cg = 0).
 * out: op-typed [B*(H/14)*(W/14), ld] with column (c*196 + dy*14 + dx), c over RGB then guide
 * channels; columns >= (3+Cg)*196 are written as zero.  mean/inv_std: [3] fp32 HOST pointers or NULL (raw
 * model, already normalised by the caller: infer.py:19).
 * split = 1 (split-precision embed): ld = 2 * segment and the row holds [hi | lo], hi = round_op(x),
 * lo = round_op(x - hi); with weights packed as [w_hi | w_hi | w_lo] the following GEMM evaluates the patch
 * embedding to ~fp32 accuracy on the fp16 matrix cores (3x the MACs of a layer that is 0.2 % of the model).
 * ---------------------------------------------------------------------------------------- */
int ada_patchify(const float* x, const float* guide, int32_t batch, int32_t cg, int32_t height,
                 int32_t width, const float* mean, const float* inv_std, void* out, int64_t ld,
                 int32_t split, void* stream);

/* Position table for a patch grid other than the native sq x sq one (DA2/dinov2.py:199-230): bicubic resample (ATen
 * upsample_bicubic2d semantics: A = -0.75, align_corners = False, the given scale factors, antialias off) of
 * pos[1:, :] viewed as [sq, sq, dim]; row 0 (cls) is copied.  pos: fp32 [1 + sq*sq, dim]; out: fp32 [1 + ph*pw, dim];
 * scale_h / scale_w: the scale_factor pair the reference passes ((ph + 0.1) / sq, (pw + 0.1) / sq).  Runs once per grid. */
int ada_pos_embed_resize(const float* pos, int32_t sq, int32_t dim, int32_t ph, int32_t pw, double scale_h,
                         double scale_w, float* out, void* stream);

/* cls row of the token matrix: tokens[b, 0, :] = cls + pos[0]  (DA2/dinov2.py:245-246). */
int ada_write_cls(float* tokens, int32_t batch, int32_t n_tokens, int32_t dim, const float* cls,
                  const float* pos0, void* stream);

/* ------------------------------------------------------------------------------------------
 * Bilinear resize, align_corners=True, on NHWC fp32 rows, with optional fused skip add:
 *   out(b,y,x,:) = bilinear(in)(b,y,x,:) [+ add(b,y,x,:)]
 * Replaces F.interpolate at DA2/util/blocks.py:144 and DA2/dpt.py:194, and the skip_add of
 * DA2/util/blocks.py:133.  Outputs (either may be NULL): fp32 plain [B*Ho*Wo, ld_f32]; op-typed
 * with row map PLAIN or PAD and optional ReLU (the activation opening the next ResidualConvUnit,
 * DA2/util/blocks.py:67).  split_seg > 0: split-precision op output as in ada_igemm_args.split_seg.
 * ---------------------------------------------------------------------------------------- */
int ada_bilinear_fwd(const float* in, int64_t ld_in, int32_t batch, int32_t hi, int32_t wi,
                     int32_t ho, int32_t wo, int32_t channels, const float* add, int64_t ld_add,
                     float* out_f32, int64_t ld_f32, void* out_op, int64_t ld_op, int32_t map_op,
                     int32_t relu, int32_t split_seg, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused DPT tail: bilinear resize (align_corners=True) of the fp32 NHWC map `in` [B, hi, wi, >= cp channels] to [B, ho, wo],
 * 3x3 convolution (padding 1) to 32 channels + bias, ReLU, 1x1 convolution to one channel + tail_b, activation -- reference
 * DA2/dpt.py:194-195 (F.interpolate + scratch.output_conv2; raw model RAW/dpt.py:148-150) in one kernel; the up-sampled
 * map is never written to memory.  w: op-typed [32, 9 * cp] tap-major (the ada_igemm CONV3 packing), cp = channels padded to
 * a multiple of 64 (weights of pad channels zero; `in` rows must hold cp readable floats: ld_in >= cp).  out: fp32 [B, ho, wo].
 * Limits (ADA_EUNSUPPORTED otherwise; callers fall back to ada_bilinear_fwd + the EP_TAIL ada_igemm): cp is 64 or 128 (the weights of
 * every 64-channel pass stay in LDS), and the vertical scale is an up-sampling by at least 1.5 (a 10-row halo tile of the output must
 * lie within 8 source rows; the model's ratio is always 14 / 8).  Any horizontal scale.
 * ---------------------------------------------------------------------------------------- */
int ada_dpt_tail_fwd(const float* in, int64_t ld_in, int32_t batch, int32_t hi, int32_t wi, int32_t ho, int32_t wo,
                     int32_t cp, const void* w, const float* bias, const float* tail_w, float tail_b, int32_t tail_act,
                     float* out, void* stream);

/* ------------------------------------------------------------------------------------------
 * Depth evaluation sums.  Replaces the host-side numpy/torch evaluation of the reference:
 * the masked reductions behind src/util/metric.py:37-160 (abs_relative_difference, squared_relative_difference,
 * rmse_linear, rmse_log, log10, threshold_percentage = delta1/2/3, i_rmse, silog_rmse) and the normal-equation sums of
 * align_depth_least_square (src/util/alignment.py:7-54), all from ONE pass over the maps.
 *   pred, gt   fp32 [B, n_per_image];  mask uint8 [B, n_per_image] (non-zero = valid) or NULL (all valid)
 *   scale_shift fp32 [B, 2] or NULL: p = pred * scale + shift is what gets evaluated (the aligned prediction);
 *               clip_lo < clip_hi additionally clamps p to [clip_lo, clip_hi] (dataset depth range)
 *   sums       fp64 [B, ADA_EVAL_NSUM] (overwritten), indexed by the ADA_EVAL_* constants; p, gt must be > 0 where valid
 *               for the log / ratio / inverse terms to be meaningful (as in the reference)
 * ---------------------------------------------------------------------------------------- */
#define ADA_EVAL_N 0          /* valid pixels */
#define ADA_EVAL_SUM_P 1      /* sum p */
#define ADA_EVAL_SUM_G 2      /* sum gt */
#define ADA_EVAL_SUM_PP 3     /* sum p*p */
#define ADA_EVAL_SUM_PG 4     /* sum p*gt */
#define ADA_EVAL_ABS_REL 5    /* sum |p-gt|/gt */
#define ADA_EVAL_SQ_REL 6     /* sum (p-gt)^2/gt */
#define ADA_EVAL_SQ 7         /* sum (p-gt)^2 */
#define ADA_EVAL_LOG_SQ 8     /* sum (ln p - ln gt)^2 */
#define ADA_EVAL_LOG 9        /* sum (ln p - ln gt) */
#define ADA_EVAL_LOG10_ABS 10 /* sum |log10 p - log10 gt| */
#define ADA_EVAL_D1 11        /* #{max(p/gt, gt/p) < 1.25} */
#define ADA_EVAL_D2 12        /* ... < 1.25^2 */
#define ADA_EVAL_D3 13        /* ... < 1.25^3 */
#define ADA_EVAL_INV_SQ 14    /* sum (1/p - 1/gt)^2 */
#define ADA_EVAL_NSUM 16
int ada_depth_eval_fwd(const float* pred, const float* gt, const uint8_t* mask, int32_t batch,
                       int64_t n_per_image, const float* scale_shift, float clip_lo, float clip_hi,
                       double* sums, void* stream);

/* ------------------------------------------------------------------------------------------
 * Device-side glue of the two-model pipeline of infer.py (kept in HBM instead of bouncing through numpy):
 *   ada_minmax_fwd     per-image min / max of a [B, n] fp32 map -> minmax[B, 2]          (infer.py:22)
 *   ada_normalize_fwd  norm = (d - min) / (max - min) and/or obs = norm * 2 - 1          (infer.py:22,92)
 *   ada_blend_fwd      out = mask > 0 ? amodal : base, then a 3x3 box blur (reflect-101 borders, = cv2.blur)
 *                      on the pixels whose 3x3 mask neighbourhood is mixed                (infer.py:30-44)
 * all tensors fp32, [B, H, W] contiguous; mask is 0/1 (anything > 0 counts as inside).
 * ---------------------------------------------------------------------------------------- */
int ada_minmax_fwd(const float* in, int32_t batch, int64_t n_per_image, float* minmax, void* stream);
int ada_normalize_fwd(const float* in, const float* minmax, int32_t batch, int64_t n_per_image, float* norm,
                      float* obs, void* stream);
int ada_blend_fwd(const float* amodal, const float* base, const float* mask, int32_t batch, int32_t height,
                  int32_t width, float* out, void* stream);

/* Per-image moments of a sigmoid-head depth map (output of nn.Sigmoid(), reference DA2/dpt.py:146-151): sums[(b * chunks + c) * 2 + {0, 1}] =
 * (sum s, sum s (1 - s)) over chunk c of image b; the caller adds the chunk sums (fixed order, no atomics: bit-reproducible).  sum s(1-s) / sum s
 * is the factor by which the sigmoid compresses the head's logit error in mean|a - b| / mean|b| for this image: hip_ext/engine.py's precision
 * ladder re-runs the DPT head in split precision for the images where it is large (depth maps concentrated near 0).  No reference counterpart.
 * ABI 8: `act` (ADA_ACT_*) names the final activation of the head and with it the pair that plays this role -- the sensitivity of the metric to a logit
 * error of mean size eps is eps * sums[1] / sums[0]:  SIGMOID (sum s, sum s (1 - s));  RELU (sum out, number of positive outputs) -- a clipped pixel
 * carries no error, and a map that is mostly clipped with the rest just above the kink has a small denominator (RAW/dpt.py:109-115,182-184);  NONE
 * ('ssi' logits: sum |out|, number of outputs). */
int ada_depth_stats_fwd(const float* in, int32_t batch, int64_t n_per_image, int32_t chunks, int32_t act, float* sums, void* stream);
/* Token diversity of an encoder tap (operand-typed [batch * rows_per_image, ld], the first `dim` columns of a row): for image b and column chunk j
 * (64 columns; ceil(dim / 64) chunks)  sums[(b * chunks + j) * 2 + {0, 1}] = (sum over the chunk's columns of Var_rows, sum of E_rows[t^2]).  The caller adds the
 * chunks; sum Var / sum E[t^2] ~ 0.02 marks inputs whose patch tokens are all alike (constant images), where the head's operand rounding errors add
 * coherently: the second trigger of hip_ext/engine.py's precision ladder.  No reference counterpart. */
int ada_token_diversity_fwd(const void* tap, int64_t ld, int32_t batch, int32_t rows_per_image, int32_t dim, float* sums, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tiled inference for inputs larger than the network's native 518 x 518 (the reference squashes every
 * input to 518 x 518, infer.py:17,84).  ada_tile_blend_fwd merges the per-tile predictions:
 *   tiles     fp32 [B, T, tile_h, tile_w]; tile t covers rows origin_y[t].. and columns origin_x[t].. of the full map
 *             (origin_* are DEVICE int32 [T]); every output pixel must be covered by at least one tile
 *   weight    separable feather min(i + 1, n - i, ramp) / ramp per axis: linear cross-fade over `ramp` pixels in overlaps
 *   out       fp32 [B, height, width] = sum_t w_t tile_t / sum_t w_t
 * ---------------------------------------------------------------------------------------- */
int ada_tile_blend_fwd(const float* tiles, int32_t batch, int32_t n_tiles, int32_t tile_h, int32_t tile_w,
                       const int32_t* origin_y, const int32_t* origin_x, int32_t height, int32_t width,
                       int32_t ramp, float* out, void* stream);

/* Hardware self-test used by the GPU test-suite: checks the MFMA / LDS-transpose fragment layouts the
 * kernels assume against a scalar computation on the device.  Returns 0 when they hold,
 * a positive bit mask of failed probes otherwise.  scratch: >= 1 MiB of device memory. */
int ada_selftest(void* scratch, int64_t scratch_bytes, void* stream);

/* ------------------------------------------------------------------------------------------
 * Tuning / diagnostic hooks.  Process-global switches (atomics) used by the test-suite (to run every tile
 * configuration and both main loops of ada_igemm against the same references), by tools/ (A/B
 * timing) and by bench.py (to report which kernel ran).  They never change results beyond fp32
 * summation order.  The environment variables ADA_IGEMM_TILE / _VARIANT / _GROUP and ADA_ATTN_VARIANT preset
 * the same switches once, at the first ada_igemm / ada_attention_fwd call.  One default and at most one alternate per
 * kernel family are built; what was measured and removed is recorded in DESIGN.md section 8 and profiles/.
 *   ada_debug_set_tile(cfg)      force the ada_igemm tile: 0 256x32, 1 128x64, 2 256x128, 3 256x256,
 *                                4 128x128; -1 = heuristic (default)
 *   ada_debug_set_variant(v)     main loop of the 256x256 tile: 0 = chosen by shape (default: the hand-scheduled 4-wave loop for
 *                                k-loops of >= 128 k-tiles, the single-barrier 8-wave loop otherwise); 4 = always the 8-wave loop;
 *                                16 = always the hand-scheduled 4-wave loop (generated assembly, csrc/ada_igemm_pipe4.inc)
 *   ada_debug_set_group(g)       force the column-group width of the tile order (0 = traffic model)
 *   ada_debug_last_tile()        tile code of the calling thread's most recent ada_igemm launch
 *                                (+200 when the hand-scheduled 4-wave main loop ran), -1 before the first launch
 *   ada_debug_set_timestamps(p)  device buffer of 8 x u64 per workgroup receiving s_memtime stamps
 *                                of the single-barrier loop, or NULL (default)
 *   ada_debug_set_attention_variant(v)  5 = 4-wave kernel with the softmax interleaved between its MFMAs (default),
 *                                3 = the 4-wave round-1 kernel (load, QK^T, softmax, PV in sequence)
 * ---------------------------------------------------------------------------------------- */
void ada_debug_set_tile(int cfg);
void ada_debug_set_variant(int v);
void ada_debug_set_group(int g);
int ada_debug_last_tile(void);
void ada_debug_set_timestamps(void* dev_buf);
void ada_debug_set_attention_variant(int v);
/* Saturation probe.  fp32 -> operand conversions clamp to the largest finite fp16 (+-65504) instead of overflowing to inf
 * (csrc/ada_common.h to_op); that is silent in the forward.  This adds, to the device counter *count (uint64, caller-zeroed), the
 * number of elements of an operand-typed buffer of n elements that sit AT the clamp (or are inf / NaN), so a caller can sweep the
 * activation buffers after a forward (hip_ext.engine.DepthEngine.saturation_report).  Asynchronous on `stream`; not on the hot path. */
int ada_debug_count_saturated(const void* buf, int64_t n, void* count, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* ADA_HIP_H */
