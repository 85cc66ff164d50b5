/* libvista_hip.so -- C ABI of the MI355X (gfx950) kernels behind the Vista denoising hot path.
 *
 * Contract: every entry point borrows device pointers, enqueues on the given HIP stream, performs
 * no allocation and no synchronisation, and returns 0 or a negative errno-style code (-22 bad argument, -5 launch
 * failure). The reference has no FFI of its own (it is pure Python); each function below names the reference
 * call site (file:line under the Vista tree) whose third-party ATen / xformers arithmetic it replaces. The Python
 * boundary classes in vista_amd/modules mirror the reference classes and are the only callers.
 *
 * Activation layout everywhere: token-major "NHWC" bf16, tensor[(b*T + t)][y*W + x][c], c contiguous.
 * bf16 is passed as uint16_t bit patterns. "stream" is a hipStream_t passed as void*.
 */
#ifndef VISTA_HIP_H
#define VISTA_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ GEMM / implicit-GEMM convolution */
enum { VK_AMODE_DENSE = 0, VK_AMODE_CONV3X3 = 1, VK_AMODE_TEMPORAL3 = 2, VK_AMODE_CONV3D = 3 };
enum { VK_EPI_LINEAR = 0, VK_EPI_GEGLU = 1, VK_EPI_TRANS = 2 };

typedef struct VkGemmDesc {
    const void* A;       /* bf16 activations: [M][lda] (DENSE) or NHWC source image stack (conv modes)            */
    const void* Wt;      /* bf16 weights [pad(N)][K] (pad: multiple of 256 and >= ceil320(N)), K contiguous; conv: [Cout][Cin/64][tap][64]                    */
    void* out;           /* bf16 or f32 [M][ldc]; EPI_TRANS: bf16 [M/S][N][S]                                     */
    const float* bias;   /* [ceil256(N)] f32 or NULL (EPI_GEGLU: in packed row order)                             */
    const float* rowvec; /* f32 [M/rows_per_vec][ldv] added per image, or NULL                                    */
    const void* res1;    /* bf16 [M][ld_res1] residual added before alpha, or NULL                                */
    const void* res2;    /* bf16 [M][ld_res2] residual added with beta, or NULL                                   */
    int32_t M, N, K;
    int32_t lda, ldc, ld_res1, ld_res2, ldv, rows_per_vec;
    float alpha, beta;   /* out = alpha*(acc + bias + rowvec + res1) + beta*res2                                  */
    int32_t amode, epi, out_f32;
    int32_t H, Wd, Cin, Hout, Wout, stride, ups; /* CONV3X3: source H x Wd (before the x`ups` nearest upsample)   */
    int32_t T, S;        /* TEMPORAL3: frames per clip, tokens per frame. EPI_TRANS: S = tokens per image.
                            CONV3D (3x3x3, pad 1 over [frames][H][Wd][Cin], K = 27*Cin): T = frames per clip        */
    int32_t tile_cfg;    /* 0 = auto; forced variants (tests / tuning): 1 = 128x128, 2 = 256x128, 3 = 256x256, 4 = 256x320 (sixteen waves),
                            5 = 128x160 (two workgroups per CU; DENSE LINEAR), 6 = the weight-stationary streaming kernel (K = 320, N = 320 / 640 /
                            960), 7 = 256x320 eight-wave pipelined kernel (DENSE / CONV3X3 / TEMPORAL3 without halos x LINEAR, bf16 out,
                            DENSE x GEGLU; bitwise equal to 4). A variant that does not take the problem falls back to the launcher's choice.
                            Weight rows are zero-padded to max(ceil256(N), ceil320(N)) so every variant reads whole tiles.
                            + 64 (with variant 0): apply the tail-split rule of vk_gemm_tail_split (an A/B option, off by default).
                            + 16: the four-wave pipelined 128x320 kernel, two workgroups per CU (DENSE x LINEAR / GEGLU, 16-bit out; bitwise equal
                            to 7; an A/B option of round 6, off by default) wherever it takes the problem and leaves the row-sum slabs alone.        */
    const void* halo_prev; /* TEMPORAL3, frame-sharded runs: bf16 [clips][S][Cin] frame preceding / following the local frame range   */
    const void* halo_next; /* (from the neighbour rank); NULL = the conv's zero padding at the window ends                               */
    void* splitk_ws;     /* optional fp32 workspace for split-K of small-M, deep-K LINEAR problems (NULL = never split); must not be  */
    int64_t splitk_ws_bytes; /* shared by GEMMs in flight on different streams. A split needs slices * M * N * 4 bytes.              */
    int32_t asym_pad;    /* CONV3X3: 0 = zero padding 1 on all sides; 1 = padding (0,1,0,1) = bottom/right only, the VAE encoder's
                            Downsample (vwm/modules/diffusionmodules/model.py:77-81: F.pad(x, (0,1,0,1)) then conv stride 2 pad 0)  */
    /* ---- DENSE two-source A: the channel concat of the UNet skip (video_model.py:493) folded into the loader ---- */
    int32_t k_split;     /* columns k >= k_split of A come from A2[m][k - k_split]; a multiple of 64; used only when A2 != NULL    */
    const void* A2;      /* bf16 [M][lda2] or NULL                                                                              */
    int32_t lda2;
    /* ---- LayerNorm folded into the consumer GEMM (attention.py:514-524, video_attention.py:119-137: `Linear(LayerNorm(x))`):
     *   LN(x) W^T + b = rstd_m * (x W'^T - mean_m * s_n) + t_n,  W' = gamma (.) W,  s_n = sum_k W'[n][k],  t_n = (W beta)_n + b_n.
     * Wt holds W', bias holds t, ln_colsum holds s (of the bf16-rounded W' so that the mean term cancels exactly); mean_m / rstd_m
     * come from per-row partial sums (sum x, sum x^2) over column blocks of x, summed here in a fixed order. DENSE only, K = LN width. */
    int32_t ln_parts;    /* number of partial-sum slabs                                                                         */
    const float* ln_stats;  /* f32 [ln_parts][M][2] or NULL (= no LayerNorm fold)                                               */
    const float* ln_colsum; /* f32 [pad(N)] (EPI_GEGLU: packed row order)                                                       */
    float ln_eps;
    /* ---- producer side of the same fold: emit the row sums of the bf16-rounded OUTPUT of this GEMM (EPI_LINEAR, bf16 out) ---- */
    float* rowstat_out;  /* f32 [vk_gemm_rowstat_parts(desc)][M][2] or NULL; one slab per (column tile, wave column)               */
    const float* rowvec2; /* f32 [M/rows_per_vec][ldv] or NULL, added with beta: out = alpha*(...) + beta*(res2 + rowvec2[row/rows_per_vec]) */
    int32_t act;         /* EPI_LINEAR: 0 = none; 1 = exact-erf GELU applied to (acc + bias + rowvec) BEFORE res1 / alpha -- the MLP of the
                            conditioner's OpenCLIP image tower: c_fc -> nn.GELU -> c_proj (vwm/modules/encoders/modules.py:273-279)           */
    /* ---- BASELINE config 5: MX-fp8 output of the LEADING columns of a LINEAR GEMM (no quantisation pass): output columns [0, mx8_cols) are
     *   written as e4m3 bytes to mx8_out[m][n] with one E8M0 scale per row and 32 columns in mx8_scales[m][n / 32] (2^e >= max|v| / 448,
     *   chosen in the epilogue); columns n >= mx8_cols go to out[m][n - mx8_cols] (ldc counts `out`'s own columns). The q | k blocks of the fused
     *   q|k|v projection feeding the fp8 QK^T of vk_attn_spatial_fp8qk (attention.py:344-346,391-407), and the FeedForward output feeding an
     *   fp8 proj_out. N % 320 == 0, mx8_cols % 320 == 0, ld_mx8 % 16 == 0; bf16 out, no rowstat_out on those columns.                       */
    void* mx8_out;
    void* mx8_scales;
    int32_t mx8_cols, ld_mx8, ld_mx8s;
    /* ---- row range (ABI v5): this call computes output rows [m_begin, m_end) of the M-row problem only; 0 / 0 = all rows. Everything else
     *   (M as the extent of the operands, of the row-sum slabs and of the conv / temporal sources) is unchanged, so a problem may be covered by
     *   several calls with bitwise the same result as one call. vk_gemm_bf16 uses it itself: when the last round of a one-tile-per-workgroup
     *   launch would leave most of the chip idle (tiles mod 256 small), the whole rounds run on the 256x320 pipelined kernel and the remaining
     *   rows as a second launch of 128x160 tiles. EPI_LINEAR / EPI_GEGLU, no split-K. ---- */
    int32_t m_begin, m_end;
    /* ---- GroupNorm statistics of the OUTPUT from the epilogue (ABI v6): the GroupNorm32 that follows a ResBlock convolution (openaimodel.py:195-199,
     *   227-234; video_model.py:38-52) needs (sum, sum of squares) per image and group of N/32 channels; instead of a read pass of its own over the
     *   tensor this GEMM is writing, the epilogue emits the stage-1 partials of vk_groupnorm_stats_bf16 -- one slot of 64 floats [32 sums | 32 sums of
     *   squares] per 64 consecutive output rows, taken from the bf16-rounded values it stores, in a fixed order (bitwise reproducible) --
     *   and vk_groupnorm_finalize_partials folds them per image group (nchunks = gn_rows / 64). Only launches for which vk_gemm_gnstat_fit() > 0
     *   may set gnstat_out (vk_gemm_bf16 returns an error otherwise): EPI_LINEAR, bf16 out, CONV3X3 / TEMPORAL3 without halos, N = 320 / 640 / 1280,
     *   gn_rows % 64 == 0, the whole row range, on the 256x320 pipelined kernel in one launch (no split-K). ---- */
    float* gnstat_out;   /* f32 [M / 64][64] or NULL */
    int32_t gn_rows;     /* output rows per image (H*W of the output) */
    /* ---- fp16 storage build only (ABI v7; libvista_hip_f16.so, vk_act_dtype() == 1; the bf16 build ignores it): output columns n >= alt_cols_from
     *   of an EPI_LINEAR 16-bit-out launch are written as bf16 instead of fp16. 0 = none; a multiple of 32. The V column block of the fused q|k|v
     *   projections (attention.py:344-346): the attention kernels keep the P.V product in bf16 in both builds (their zero-base softmax needs bf16's
     *   exponent range for the numerators), so V must arrive as bf16 while q and k are fp16. No row sums / GroupNorm statistics on such a launch. ---- */
    int32_t alt_cols_from;
} VkGemmDesc;

/* nn.Linear / nn.Conv2d / nn.Conv3d call sites of the UNet:
 *   vwm/modules/attention.py:85-92,117-121 (GEGLU FF), :344-346,421 (q,k,v,out), :579,602 (proj_in/out)
 *   vwm/modules/diffusionmodules/openaimodel.py:136 (Downsample), :84,100-102 (Upsample), :195-199,227-234 (ResBlock convs), :241 (skip)
 *   vwm/modules/diffusionmodules/video_model.py:38-52 (3x1x1 temporal conv), :148-157,176-182 (embedding MLPs), :189,438 (in/out conv) */
int vk_gemm_bf16(const VkGemmDesc* d, void* stream);

/* Number of partial-sum slabs vk_gemm_bf16 would write to d->rowstat_out for this problem (depends on the block tile the launcher
 * picks: column tiles x wave columns); > 0, or a negative error code. Pure host function, no launch. */
int vk_gemm_rowstat_parts(const VkGemmDesc* d);
/* The launcher's decision for `desc` without launching anything (host arithmetic only): block-tile variant (1 = 128x128, 2 = 256x128,
 * 3 = 256x256, 4 = 256x320, 5 = 128x160) * 16 + number of K slices; negative = the error vk_gemm_bf16 would return. */
int vk_gemm_tile_choice(const VkGemmDesc* d);
/* The output row at which vk_gemm_bf16 would split this problem into two launches (whole rounds of 256x320 tiles on the pipelined kernel +
 * the remaining rows as 128x160 tiles; VkGemmDesc.m_begin / m_end), 0 = a single launch; negative = the error vk_gemm_bf16 would return.
 * Pure host function, no launch. */
int vk_gemm_tail_split(const VkGemmDesc* d);
/* ABI v6: > 0 (= M / 64, the number of 64-float slots gnstat_out needs) when vk_gemm_bf16 would take `d` with gnstat_out set, 0 when this
 * problem cannot emit GroupNorm statistics (the caller then runs vk_groupnorm_stats_bf16 as before); negative = the error vk_gemm_bf16 would
 * return. Evaluated with d->gnstat_out ignored (the launcher's kernel choice does not depend on it). Pure host function, no launch. */
int vk_gemm_gnstat_fit(const VkGemmDesc* d);

/* fp8 (OCP e4m3) variant of the DENSE GEMM for the UNet's Linear / 1x1 projections (BASELINE.json config 5: "fp8 1x1
 * conv-as-GEMM path"; same reference call sites as vk_gemm_bf16's DENSE mode: attention.py:268-285,97-128, video_attention.py).
 *   out = epilogue( a_scale[m] * w_scale[n] * sum_k Aq[m][k] * Wq[n][k] )
 * d->A = fp8 activations [M][lda] (bytes), d->Wt = fp8 weights [ceil-tile(N)][d->K] with d->K = row stride, a multiple of 128,
 * zero-filled past k_real; a_scale [M] per-row, w_scale [ceil-tile(N)] per-output-channel (GEGLU: packed row order);
 * amode must be DENSE, epi LINEAR or GEGLU; the epilogue fields (bias, rowvec, res1/res2, alpha/beta, out_f32) as for bf16. */
int vk_gemm_fp8(const VkGemmDesc* desc, const float* a_scale, const float* w_scale, int32_t k_real, void* stream);

/* The same GEMM with the round-2 options (all optional; vk_gemm_fp8(d, a_scale, w_scale, k_real) == {a_scale, w_scale, k_real, 0...}):
 *   a_mx / ld_mx     : MX block scales of the activations INSTEAD of a_scale: E8M0 bytes (2^(e-127)) [M][ld_mx], one per 32 consecutive
 *                      K-elements, applied inside v_mfma_scale_f32_32x32x64_f8f6f4; ld_mx % 4 == 0 and ld_mx * 32 >= d->K.
 *   mx_out / ld_mx_out: EPI_GEGLU only: write the gated output as MX fp8 -- d->out = e4m3 bytes [M][d->ldc], mx_out = E8M0 [M][ld_mx_out],
 *                      one scale per 32 output columns, chosen in the epilogue (2^e >= max|h| / 448) -- so that the FeedForward's second
 *                      GEMM (a_mx = this mx_out) needs no quantisation pass (attention.py:85-128). (N/2) % 32 == 0, ldc % 16 == 0.
 * d->rowstat_out is honoured (vk_gemm_fp8_rowstat_parts sizes it).
 * Implicit-GEMM convolutions (d->amode = CONV3X3 / TEMPORAL3, the ResBlock convolutions of openaimodel.py:300-318 / video_model.py:38-52 on
 * the e4m3 output of vk_groupnorm_silu_fp8): stride 1, pad 1, no upsample / halo, weights [Cout][Cin/64][tap][64] bytes padded to a multiple
 * of 128 per row (d->K), k_real = taps * Cin, Cin % 64 == 0, bf16 output, a_scale + a_scale_rows. */
typedef struct VkFp8Args {
    const float* a_scale;
    const float* w_scale;
    int32_t k_real;
    const void* a_mx;
    int32_t ld_mx;
    void* mx_out;
    int32_t ld_mx_out;
    int32_t a_scale_rows; /* 0 / 1: a_scale is per row; n > 1: one scale per n consecutive rows (the per-image scales of
                             vk_groupnorm_silu_fp8: n = H*W, or T*H*W for the temporal norm) */
} VkFp8Args;
int vk_gemm_fp8_mx(const VkGemmDesc* desc, const VkFp8Args* args, void* stream);
int vk_gemm_fp8_rowstat_parts(const VkGemmDesc* d);

/* Fused FeedForward of the level-0 (width 320) transformers: net = [GEGLU, Dropout, Linear] (vwm/modules/attention.py:85-128) applied as
 * `x = ff(norm3(x)) + x` (attention.py:524) and as ff_in / ff of the temporal block with its AlphaBlender mix (video_attention.py:119-121,
 * 137-141) -- ONE kernel, the hidden activation (M x 1280 bf16, 1.18 GB per call at the BASELINE shape) never goes to HBM.
 *   geglu    : the in-projection exactly as vk_gemm_bf16 would take it (amode DENSE, epi GEGLU: A = x [M][lda], Wt = packed GEGLU weight
 *              [2 Hd][320], bias, optional folded LayerNorm ln_*); K must be 320. `out` / `ldc` are ignored.
 *   out_proj : the out-projection as vk_gemm_bf16 would take it (amode DENSE, epi LINEAR, bf16 out: bias, rowvec / rowvec2, res1 / res2,
 *              alpha / beta, rowstat_out), N = 320, K = Hd (a multiple of 64, 128 <= Hd <= 1280), with ONE difference: Wt's K axis is
 *              permuted inside every group of 16 hidden units to [0-3, 8-11, 4-7, 12-15] (the order in which a lane of the
 *              in-projection's MFMA accumulator holds them, so the hidden values feed the second MFMA straight from registers) and the
 *              matrix is stored chunk-major, bf16 [Hd / 32][320][32] (a 32-wide hidden chunk of all rows = one contiguous 20 KB block;
 *              bias f32 [320]). `A` / `lda` are ignored.
 * The hidden activation is rounded to bf16 exactly where the two-kernel form rounds it; rowstat_out gets vk_ff_fused_rowstat_parts()
 * slabs ([parts][M][2]). Same return codes as vk_gemm_bf16. */
int vk_ff_fused_bf16(const VkGemmDesc* geglu, const VkGemmDesc* out_proj, void* stream);
int vk_ff_fused_rowstat_parts(void);

/* GroupNorm(32)[+SiLU] with e4m3 output and ONE scale per image group (of frames_per_group images): y8 = e4m3(y / scale[g]), scale[g] =
 * (max_c |a_c| max|x| + |b_c|) / 448 with y = a_c x + b_c the folded affine form -- an upper bound from the statistics pass (which also
 * tracks max|x|), so no extra pass over the data; a conv output pixel sums taps of its own image only, so the scale factors out of the
 * fp8 convolution (VkFp8Args.a_scale_rows). x2 != NULL: the input is the channel concat [x1 | x2] (C1 + C2 channels), as
 * vk_groupnorm_silu_cat_bf16. stats_ws: floats, groups * 64 + n_img * ceil(S / 32) * 65.
 * Reference: GroupNorm32 + SiLU of openaimodel.py:281-298, video_model.py:38-52. */
int vk_groupnorm_silu_fp8(const void* x1, const void* x2, void* y8, float* scale_out, const float* gamma, const float* beta, float* stats_ws,
                          int32_t n_img, int32_t S, int32_t C1, int32_t C2, int32_t frames_per_group, float eps, int32_t silu, void* stream);

/* LayerNorm fused with per-row dynamic fp8 quantisation: y = LN(x)*gamma + beta (as vk_layernorm_bf16), scale[r] = max|y[r]| / 448,
 * q[r] = e4m3(y[r] / scale[r]) -- one pass, the normalised tensor is never written in bf16 (FeedForward input of the fp8 configuration:
 * attention.py:524, video_attention.py:119-137). C % 8 == 0, C <= 1536. */
int vk_layernorm_quant_fp8(const void* x, void* q, float* scale, const float* gamma, const float* beta, int32_t rows, int32_t C, float eps,
                           void* stream);

/* Per-row dynamic quantisation bf16 -> fp8 e4m3: scale[m] = max|x[m][:]| / 448, q = e4m3(x / scale). K % 8 == 0, K <= 5120. */
int vk_quantize_rows_fp8(const void* x, void* q, float* scale, int32_t M, int32_t K, int64_t ldx, int64_t ldq, void* stream);

/* ------------------------------------------------------------------ attention */
/* Spatial self-attention, softmax(q k^T * scale) v per (image, head), head dim 64, no mask.
 * Replaces xformers.ops.memory_efficient_attention at vwm/modules/attention.py:400-407 for attn1 of
 * BasicTransformerBlock (attention.py:514-518).
 *   q, k : bf16, row (image*S + token) at q + row*ldq + head*64 (same for k / ldk)
 *   vt   : bf16 [n_img][heads][64][S]   (V transposed, produced by vk_gemm_bf16 EPI_TRANS)
 *   o    : bf16, row at o + row*ldo + head*64
 * S must be a multiple of 8. */
int vk_attn_spatial_bf16(const void* q, const void* k, const void* vt, void* o, int32_t n_img, int32_t heads,
                         int32_t S, int32_t ldq, int32_t ldk, int32_t ldo, float scale, void* stream);

/* The same attention with V as a token-major column block [row][ldv] of the fused q|k|v projection (vwm/modules/attention.py:344-346 as ONE
 * GEMM): no V^T tensor and no second pass over x -- the kernel transposes the V tile on its way out of LDS (ds_read_b64_tr_b16).
 *   v : bf16, row (image*S + token) at v + row*ldv + head*64;  ldv % 8 == 0. Everything else as vk_attn_spatial_bf16. */
int vk_attn_spatial_qkv_bf16(const void* q, const void* k, const void* v, void* o, int32_t n_img, int32_t heads,
                             int32_t S, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, float scale, void* stream);

/* The same attention for a query that ALREADY carries softmax_scale * log2(e) -- folded into the to_q weight rows when they are packed
 * (vwm/modules/attention.py:344,400-407: q = to_q(x), scale = dim_head ** -0.5; one bf16 rounding of the scaled projection instead of one of the
 * unscaled one: no extra error). q . k is then the base-2 exponent itself; rows whose running maximum lies within +-60 octaves use the base 0
 * (P = exp2(q . k), no per-score scale / base arithmetic), others re-base on their true maximum exactly like vk_attn_spatial_qkv_bf16. */
int vk_attn_spatial_qkv_log2_bf16(const void* q, const void* k, const void* v, void* o, int32_t n_img, int32_t heads,
                                  int32_t S, int32_t ldq, int32_t ldk, int32_t ldv, int32_t ldo, void* stream);

/* Small dense attention, any head dim D in {64, 80, 128}: softmax(q k^T * scale) v per (image, head); q / k / v are column blocks of one
 * row-major buffer (q at qkv + row*ld + head*D, k at + k_off, v at + v_off; row = image*S + token). The nn.MultiheadAttention of the
 * conditioner's OpenCLIP ViT-H/14 image tower (FrozenOpenCLIPImageEmbedder, vwm/modules/encoders/modules.py:251-399: 257 tokens, 16 heads
 * of dim 80). 4*S*D bytes of LDS (<= 160 KiB). */
int vk_attn_small_bf16(const void* qkv, void* o, int32_t n_img, int32_t heads, int32_t S, int32_t D, int32_t ld, int32_t k_off,
                       int32_t v_off, int32_t ldo, float scale, void* stream);

/* OpenCLIP image preprocessing + patchify (FrozenOpenCLIPImageEmbedder.preprocess, modules.py:304-315, and the im2col of the tower's
 * patch convolution): kornia-0.6.9 antialiased bicubic resize of img f32 [n][3][H][W] in [-1, 1] to out_hw x out_hw (Gaussian blur with
 * (sigma_y, sigma_x) / odd kernel sizes (ks_y, ks_x) <= 63, reflect border, then bicubic align_corners=True), (x + 1) / 2, CLIP mean / std
 * (HOST arrays of 3 floats), written as the bf16 A operand of the patch-embedding GEMM:
 *   out[(img * (1 + g*g) + 1 + py*g + px) * ldo + c*patch*patch + ky*patch + kx],  g = out_hw / patch.
 * Only those columns are written: the caller zero-fills `out` (class-token row of every image, K padding) once. */
int vk_clip_preprocess_patches(const float* img, void* out, int32_t n_img, int32_t H, int32_t W, int32_t out_hw, int32_t patch, int32_t ldo,
                               float sigma_y, float sigma_x, int32_t ks_y, int32_t ks_x, const float* mean3, const float* std3, void* stream);

/* BASELINE config 5: the same attention with the score product in fp8. q8 / k8: e4m3 bytes, row (image*S + token) at q8 + row*ldq8 + head*64;
 * q_scales / k_scales: E8M0 bytes, one per row and 32 head-dim elements, at + row*ldqs + 2*head + block -- the MX-fp8 q | k blocks written by the
 * fused q|k|v projection's epilogue (VkGemmDesc.mx8_out). S^T = K . Q^T runs as v_mfma_scale_f32_32x32x64_f8f6f4 with both block scales applied
 * inside the instruction; softmax, bf16 P and P.V on the bf16 V rows (v / ldv as in vk_attn_spatial_qkv_bf16) are unchanged.
 * Output: bf16 `o` (ldo), or -- o8 != NULL -- MX fp8: e4m3 bytes at o8 + row*ldo8 + head*64 and E8M0 scales at o_scales + row*ldos + 2*head + block,
 * which an fp8 attention-out projection consumes through VkFp8Args.a_mx (attention.py:391-421).
 * scale == 0: q already carries softmax_scale * log2(e) (the pre-scaled query of vk_attn_spatial_qkv_log2_bf16; zero-base softmax). */
int vk_attn_spatial_fp8qk(const void* q8, const void* k8, const void* q_scales, const void* k_scales, const void* v, void* o, void* o8,
                          void* o_scales, int32_t n_img, int32_t heads, int32_t S, int32_t ldq8, int32_t ldk8, int32_t ldqs, int32_t ldks,
                          int32_t ldv, int32_t ldo, int32_t ldo8, int32_t ldos, float scale, void* stream);

/* Temporal (cross-frame) self-attention over the T frames of every pixel: sequence length T <= 32, head dim 64.
 * Replaces the batchified xformers call at vwm/modules/attention.py:384-399 for VideoTransformerBlock.attn1
 * (vwm/modules/video_attention.py:116-127). Token rows are (b*T + t)*S + s; q,k,v are column blocks of one
 * row-major buffer: q at qkv + row*ld + head*64, k at + k_off, v at + v_off. */
int vk_attn_temporal_bf16(const void* qkv, void* o, int32_t B, int32_t T, int32_t S, int32_t heads, int32_t ld,
                          int32_t k_off, int32_t v_off, int32_t ldo, float scale, void* stream);

/* Row softmax, fp32 scores -> bf16 probabilities: y[r][c] = softmax_c(x[r][:cols]). cols % 4 == 0, cols <= 16384.
 * Replaces `torch.nn.functional.softmax(w_, dim=2)` of the VAE decoder's single-head AttnBlock
 * (vwm/modules/diffusionmodules/model.py:160-166); q.k^T (scaled) and P.v run as vk_gemm_bf16 calls either side. */
int vk_softmax_rows_f32_bf16(const float* x, void* y, int64_t rows, int32_t cols, int64_t ldx, int64_t ldy, void* stream);

/* ------------------------------------------------------------------ normalisation */
/* GroupNorm(32 groups) [+ SiLU] over token-major x[n_img][S][C]; statistics span `frames_per_group` consecutive
 * images (1 = per-image GroupNorm32 of the 2-D ResBlock / transformer entry norm; T = the 5-D norm of the
 * temporal ResBlock whose statistics cover (C/32, T, H, W)).
 * Replaces GroupNorm32/Normalize + nn.SiLU: vwm/modules/diffusionmodules/util.py:196-216, attention.py:141-142,
 * openaimodel.py:195-199,227-230, video_model.py:434-436.
 * stats_ws: f32 workspace of 64*(n_img/frames_per_group) + 64*n_img*ceil(S/32) floats (fixed-order partial sums:
 * results are bitwise reproducible, no atomics). */
int vk_groupnorm_silu_bf16(const void* x, void* y, const float* gamma, const float* beta, float* stats_ws,
                           int32_t n_img, int32_t S, int32_t C, int32_t frames_per_group, float eps, int32_t silu,
                           void* stream);

/* The two halves of vk_groupnorm_silu_bf16, for pixel-sharded multi-GPU runs of the temporal ResBlock: `sums` is
 * [n_img/frames_per_group][64] = raw [32 sums | 32 sums of squares] over the LOCAL elements (all-reduce them across ranks),
 * `partial_ws` needs 64*n_img*ceil(S/32) floats, `count` is the GLOBAL element count per (image-group, channel-group). */
int vk_groupnorm_stats_bf16(const void* x, float* sums, float* partial_ws, int32_t n_img, int32_t S, int32_t C,
                            int32_t frames_per_group, void* stream);
int vk_groupnorm_apply_bf16(const void* x, void* y, const float* gamma, const float* beta, const float* sums, int32_t n_img,
                            int32_t S, int32_t C, int32_t frames_per_group, float count, float eps, int32_t silu, void* stream);
/* ABI v6: the second stage of vk_groupnorm_stats_bf16 on partials somebody else produced (VkGemmDesc.gnstat_out): `partial` is
 * [n_img][nchunks][64] stage-1 slots (consumed: the fold of large groups works in place), `sums` [n_img/frames_per_group][64] as above.
 * Followed by vk_groupnorm_apply_bf16 (after the all-reduce of `sums` in a pixel-sharded run) it replaces vk_groupnorm_silu_bf16 without the
 * statistics pass over x. */
int vk_groupnorm_finalize_partials(float* partial, float* sums, int32_t n_img, int32_t nchunks, int32_t frames_per_group, void* stream);
/* ABI v7: the apply pass straight on stage-1 slots (`partial` as above, NOT consumed): every workgroup folds its image group's
 * frames_per_group * nchunks slots itself, in gn_finalize's summation order, so the output is bitwise that of vk_groupnorm_finalize_partials +
 * vk_groupnorm_apply_bf16 with one launch less (GroupNorm32 of openaimodel.py:195-199,227-234 = producer epilogue + ONE pass). Taken when
 * frames_per_group * nchunks <= vk_groupnorm_fold_max() (256; 0 when VISTA_GN_FOLD=0), else VK_EINVAL. Not for pixel-sharded norms (their raw
 * sums are all-reduced between the two stages). vk_groupnorm_silu_bf16 / _cat_bf16 apply the same rule to their own statistics pass. */
int vk_groupnorm_fold_max(void);
int vk_groupnorm_apply_partials_bf16(const void* x, void* y, const float* gamma, const float* beta, const float* partial, int32_t n_img,
                                     int32_t S, int32_t C, int32_t nchunks, int32_t frames_per_group, float count, float eps, int32_t silu,
                                     void* stream);

/* LayerNorm over C of x[rows][C] (+ optional per-image pre-add vector):  u = x + addvec[row / rows_per_vec];
 * if sum_out: sum_out = u (bf16);  y = LN(u)*gamma + beta.
 * Replaces nn.LayerNorm at attention.py:514-524 (norm1-3) and video_attention.py:119-137 (norm_in, norm1-3),
 * fused with the `x_mix = x + emb` add of video_attention.py:283-284. */
int vk_layernorm_bf16(const void* x, void* y, void* sum_out, const float* gamma, const float* beta,
                      const float* addvec, int32_t rows, int32_t C, int32_t rows_per_vec, int32_t ldv, float eps,
                      void* stream);

/* Row sums for a LayerNorm folded into the consumer GEMM (VkGemmDesc.ln_stats with ln_parts = 1) when the tensor was not produced
 * by a GEMM epilogue on this rank (pixel-sharded temporal block after the all-to-all): stats[row] = (sum_c x, sum_c x^2) of
 * x[rows][ldx] (first C columns), fixed order. Read-only pass: half the traffic of vk_layernorm_bf16. C % 8 == 0, C <= 1536. */
int vk_rowstats_bf16(const void* x, float* stats, int32_t rows, int32_t C, int64_t ldx, void* stream);

/* vk_groupnorm_silu_bf16 over the channel concat [x1 | x2] (C1 + C2 channels, both token-major, C1 % 8 == C2 % 8 == 0) without
 * materialising it: the `torch.cat([h, hs.pop()], dim=1)` of the UNet's output blocks (video_model.py:493) feeding the ResBlock's
 * first GroupNorm32 (openaimodel.py:195-199). y is [n_img][S][C1+C2]. stats_ws as for vk_groupnorm_silu_bf16. */
int vk_groupnorm_silu_cat_bf16(const void* x1, const void* x2, void* y, const float* gamma, const float* beta, float* stats_ws,
                               int32_t n_img, int32_t S, int32_t C1, int32_t C2, int32_t frames_per_group, float eps, int32_t silu,
                               void* stream);

/* ------------------------------------------------------------------ elementwise / layout */
/* out[m][0:C1] = a[m][:], out[m][C1:C1+C2] = b[m][:]  (channel concat of the UNet skip: video_model.py:493) */
int vk_concat_channels_bf16(const void* a, const void* b, void* out, int64_t rows, int32_t C1, int32_t C2, void* stream);

/* NCHW float (f32) -> token-major bf16 with the channel dim zero-padded to Cpad (UNet entry; video_model.py:473) */
int vk_nchw_to_tokens_bf16(const float* x, void* out, int32_t n_img, int32_t C, int32_t HW, int32_t Cpad, void* stream);
/* token-major f32 [n_img][HW][ldx] (first C columns) -> NCHW f32 (UNet exit; video_model.py:502-503) */
int vk_tokens_to_nchw_f32(const float* x, float* out, int32_t n_img, int32_t C, int32_t HW, int32_t ldx, void* stream);

/* sinusoidal timestep embedding: out[n][0:half]=cos(t[n]*f_j), out[n][half:]=sin(t[n]*f_j), f_j=exp(-ln(max_period)*j/half)
 * (vwm/modules/diffusionmodules/util.py:141-165); bf16 output feeds the embedding MLPs */
int vk_timestep_embedding_bf16(const float* t, void* out, int32_t n, int32_t dim, float max_period, void* stream);
/* The same sinusoid in fp32: ConcatTimestepEmbedderND of the conditioner (vwm/modules/encoders/modules.py:402-425 over openaimodel.py Timestep). */
int vk_timestep_embedding_f32(const float* t, float* out, int32_t n, int32_t dim, float max_period, void* stream);

/* emb = a*mask[n] + b*(1-mask[n]) + c  (video_model.py:457-471); writes emb (f32) and silu(emb) (bf16, the input
 * of every ResBlock emb_layers: openaimodel.py:222-225). a may be NULL (no cond-frame mask: emb = b + c). */
int vk_emb_combine(const float* a, const float* b, const float* c, const float* mask, float* emb, void* silu_out,
                   int32_t n, int32_t dim, void* stream);

/* y = silu(x) f32 -> bf16 (time_pos_embed MLP hidden: video_attention.py:227-231) */
int vk_silu_f32_to_bf16(const float* x, void* y, int64_t count, void* stream);
/* y = bf16(x) */
int vk_cast_f32_to_bf16(const float* x, void* y, int64_t count, void* stream);

/* ---- sampler-side fused elementwise (EulerEDMSampler step; sampling.py:78-89,104-122, guiders.py:23-36,
 *      denoiser.py:30-35, denoiser_scaling.py:51-59, wrappers.py:28-31) ----
 * x, cond_frame: f32 NCHW [T][4][HW]; mask: f32 [T].
 * vk_sampler_prepare: xr = x*(1-mask) + cond_frame*mask (written back to x when replace != 0) and builds the
 *   CFG-doubled UNet input in token-major bf16 [2T][HW][Cpad]: ch 0-3 = xr * c_in, ch 4-7 = concat latent
 *   (uncond half: concat_uc, cond half: concat_c; f32 NCHW [T][4][HW]), remaining channels 0. */
int vk_sampler_prepare(float* x, const float* cond_frame, const float* mask, const float* concat_uc,
                       const float* concat_c, void* net_in, int32_t T, int32_t HW, int32_t Cpad, float c_in,
                       int32_t replace, void* stream);
/* vk_sampler_update: net_out f32 token-major [2T][HW][ld] (first 4 cols). den = net*c_out + x*c_skip for both
 *   halves; g = den_u + scale[t]*(den_c - den_u); d = (x - g)/sigma; x += d*(sigma_next - sigma). */
int vk_sampler_update(float* x, const float* net_out, const float* scale, int32_t T, int32_t HW, int32_t ld,
                      float c_out, float c_skip, float sigma, float sigma_next, void* stream);

/* generic pieces of the same maths for callers that use the reference's Denoiser / guider API directly */
/* out = net*c_out[n] + x*c_skip[n]  (NCHW f32, per-image coefficients; denoiser.py:35) */
int vk_denoiser_combine(const float* net, const float* x, const float* c_out, const float* c_skip, float* out,
                        int32_t n_img, int32_t chw, void* stream);
/* out[t] = u[t] + scale[t]*(c[t]-u[t]) with u = x[0:T], c = x[T:2T]  (guiders.py:23-26,54-61) */
int vk_cfg_combine(const float* x2, const float* scale, float* out, int32_t T, int32_t chw, void* stream);
/* x_next = x + (x - den)/sigma[n] * (sigma_next[n]-sigma[n])  (sampling_utils.py:46-47; sampling.py:66-67,85-88) */
int vk_euler_step(const float* x, const float* den, const float* sigma, const float* sigma_next, float* out,
                  int32_t n_img, int32_t chw, void* stream);
/* out = x*(1-mask[n]) + cond*mask[n]  (sampling.py:106,122) */
int vk_mask_replace(const float* x, const float* cond, const float* mask, float* out, int32_t n_img, int32_t chw,
                    void* stream);
/* out = x * s[n]  (per-image scale, NCHW f32; denoiser.py:35 `noised_input * c_in`) */
int vk_scale_rows(const float* x, const float* s, float* out, int32_t n_img, int32_t chw, void* stream);

/* Diagonal-Gaussian posterior of the first-stage encoder (vwm/modules/distributions/distributions.py:24-37 and
 * regularizers/__init__.py:30-39), fused with encode_first_stage's `z * scale_factor` (vwm/models/diffusion.py:194):
 * moments f32 NCHW [n][2C][hw] = (mean | logvar); out[n][C][hw] = (mean + exp(0.5*clamp(logvar,-30,20)) * noise) * scale;
 * noise NULL = the posterior mode (mean * scale). */
int vk_gaussian_sample(const float* moments, const float* noise, float* out, int32_t n_img, int32_t C, int32_t hw, float scale,
                       void* stream);

/* Reward estimation (reward_utils.py:329-335): out[0] = sum_i sum_e (x[e][i] - mean_e)^2 / (E - 1) over an ensemble x[E][n] of f32
 * latents, reduced in a fixed order (f64 partials; partial_ws: 512 doubles). The caller forms exp(-out/n). */
int vk_ensemble_variance_sum(const float* x, double* out, double* partial_ws, int32_t E, int64_t n, void* stream);

/* library info */
int vk_abi_version(void);
/* ABI v7: the 16-bit storage type this library was built for: 0 = bf16 (libvista_hip.so, the default and the BASELINE config's dtype), 1 = IEEE fp16
 * (libvista_hip_f16.so, -DVK_F16=1: the reference's autocast width, sample_utils.py:301-303). Every "bf16" in the entry-point names and comments of
 * this header reads "the storage type" for the fp16 build; fp32 interfaces are unchanged. The fp8 entry points (BASELINE config 5) exist in the bf16
 * build only and return VK_EINVAL in the other. */
int vk_act_dtype(void);

#ifdef __cplusplus
}
#endif
#endif
