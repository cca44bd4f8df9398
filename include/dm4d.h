/*
 * dm4d.h -- C ABI of libdm4d.so: the MI355X (gfx950) operator library underneath the
 * Diffuman4D sliding iterative denoiser.
 *
 * The reference (zju3dv/Diffuman4D) has no native layer and no FFI: every device op on its hot
 * path is an implicit torch / diffusers call.  Each entry point below
 * therefore cites the *call site* in the reference whose arithmetic it replaces.  A reference
 * maintainer binds these with ctypes; diffuman4d_amd/host/lib.py is that
 * binding.
 *
 * Conventions
 *   - all tensors are device pointers owned by the caller; bf16 unless noted; activations are
 *     token-major / NHWC: [B, H*W, C] with C contiguous
 *   - `stream` is a hipStream_t passed as void* (0 = default stream); every call is asynchronous
 *   - return value: 0 = ok, <0 = error; dm4d_last_error() returns a thread-local message
 *   - no entry point allocates, synchronises or touches host memory
 */
#ifndef DM4D_H
#define DM4D_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DM4D_OK 0
#define DM4D_ERR_ARG -1
#define DM4D_ERR_LAUNCH -2

/* gemm / conv epilogue flags */
#define DM4D_EPI_GEGLU 1u /* W holds [2*N, K]: out = (x W_h^T + b_h) * gelu(x W_g^T + b_g)   */
#define DM4D_EPI_SILU 2u  /* out = silu(acc + bias ...) (time embedding MLP)                 */
#define DM4D_EPI_F32OUT 4u /* C is float* [M, ldc] (ldc in floats): the result is stored unrounded */
/* parity-precision launches (two-term bf16 operands, fp32 tensors between kernels; see "Parity precision" below) */
#define DM4D_EPI_F32SIDE 8u   /* rowbias and residual are float* (their strides in floats) */
#define DM4D_EPI_SPLITOUT 16u /* C is bf16 [M, ldc >= 2 N]: hi = bf16(result) at column n, lo = bf16(result - hi) at column N + n */
/* fp16-precision launches (see "fp16 precision" below): set by the *_f16 entries themselves, never passed by a caller */
#define DM4D_EPI_H16 32u      /* A, W, bias and every 16-bit side input / output are IEEE fp16; the matrix unit runs v_mfma_f32_32x32x16_f16 */

int dm4d_version(void);
const char* dm4d_last_error(void);

/* Linear layers: out[M,N] = epi( [A | A2][M,K] @ W[N,K]^T )
 *   replaces nn.Linear calls of diffusers Attention.to_q/k/v/to_out, Transformer2DModel.proj_in/out,
 *   FeedForward/GEGLU, TimestepEmbedding, ResnetBlock2D.time_emb_proj and the 1x1 conv_shortcut
 *   (reference call sites: attention.py:73-78,90,142; transformer_multiview.py:160,209;
 *   unet_multiview_condition.py:520; unet_multiview_blocks.py:342,518,697).
 *   A2 != NULL: columns [K1, K) come from A2 (channel concat of the up-block skip, :667) .
 *   epilogue: acc (+bias[n]) (GEGLU|SILU) (+rowbias[m / rows_per_rowbias, n]) (+residual[m,n]) * out_scale
 *   K, K1 multiples of 32; lda/ldw multiples of 8.                                              */
int dm4d_gemm_bf16(void* stream, const void* A, int64_t lda, const void* A2, int64_t lda2, int K1, const void* W,
                   int64_t ldw, void* C, int64_t ldc, int M, int N, int K, const void* bias, const void* rowbias,
                   int64_t ld_rowbias, int rows_per_rowbias, const void* residual, int64_t ld_res, unsigned flags,
                   float out_scale);

/* 3x3 convolution, NHWC, implicit GEMM on MFMA.
 *   replaces nn.Conv2d(k=3) in ResnetBlock2D.conv1/conv2, Downsample2D.conv (stride 2),
 *   Upsample2D (nearest x2 fused into the gather: upsample=1), conv_in, conv_out
 *   (unet_multiview_condition.py:549,593; unet_multiview_blocks.py:342,381,460,518,620,697).
 *   X [B,H,W,Cin] (Cin % 32 == 0), Wt [Cout][3][3][Cin], Y [B,Ho,Wo,Cout];
 *   Ho = upsample ? 2H : (H + 2*pad_hi... see dm4d_conv_out_size). pad = leading pad (1 for the UNet,
 *   0 for the VAE encoder's asymmetric (0,1,0,1) pad); trailing pad is implied by Ho/Wo.
 *   epilogue as in dm4d_gemm_bf16 (rowbias row = batch index: the temb projection, resnet.py)   */
int dm4d_conv3x3_nhwc_bf16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y, int Ho,
                           int Wo, int Cout, int stride, int pad, int upsample, const void* bias, const void* rowbias,
                           int64_t ld_rowbias, const void* residual, int64_t ld_res, float out_scale);

/* Same convolution with a caller-owned fp32 workspace.  Stride-1 convolutions on small images with a deep K (H*W <= 64,
 *   Cin >= 512: the 9x5 level of the UNet, unet_multiview_blocks.py:71,274,585 at the deepest resolution) are split
 *   over the three kernel rows -- three workgroups per output tile, partial sums in ws, a second launch adds them in a
 *   fixed order and applies the epilogue -- because B*45 output rows cannot fill 256 CUs against an 11520-deep K.
 *   dm4d_conv3x3_ws_bytes() returns the bytes needed for a shape (0 = never split); ws == NULL or too small runs the
 *   un-split kernel.  The split depends on the per-image geometry only, not on B.                                   */
size_t dm4d_conv3x3_ws_bytes(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int stride, int pad, int upsample);
int dm4d_conv3x3_nhwc_bf16_ws(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y, int Ho,
                              int Wo, int Cout, int stride, int pad, int upsample, const void* bias, const void* rowbias,
                              int64_t ld_rowbias, const void* residual, int64_t ld_res, float out_scale, void* ws,
                              size_t ws_bytes);

/* The same convolution with epilogue flags (DM4D_EPI_F32OUT: Y is float* [B,Ho,Wo,Cout]; DM4D_EPI_F32SIDE: rowbias / residual are
 *   float*): the parity-precision form.  X then usually carries two-term operands, [hi(Cin/2) | lo(Cin/2)] per pixel, against weights
 *   duplicated along Cin; the kernel neither knows nor cares.  Never split over the kernel rows (no workspace).             */
int dm4d_conv3x3_nhwc_bf16_flags(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y, int Ho,
                                 int Wo, int Cout, int stride, int pad, int upsample, const void* bias, const void* rowbias,
                                 int64_t ld_rowbias, const void* residual, int64_t ld_res, float out_scale, unsigned flags);

/* Upsample2D -- third-party: diffusers==0.33.1 (requirements.txt:5), instantiated by the reference at
 * src/diffusers/models/unets/unet_multiview_blocks.py:620 and by the VAE decoder; its published forward is
 * F.interpolate(x, scale_factor=2.0, mode="nearest") then conv3x3(padding=1) -- as four 2x2 convolutions of the
 * LOW-resolution input, one per output phase (restated and checked against that form in oracle/up2x.py):
 * 4/9 of the multiply-adds.  `Wp` [4][Cout][4*Cin] is made once per layer from the 3x3 weights W [Cout][9*Cin]
 * (ky, kx, ci order) by dm4d_conv_up2x_prepare_bf16 (sums of 1, 2 or 4 taps in fp32, rounded to bf16 once).
 * X [B, H, W, Cin] -> Y [B, 2H, 2W, Cout], Cin % 64 == 0, Cout % 8 == 0; bias optional.                               */
int dm4d_conv_up2x_prepare_bf16(void* stream, const void* W, void* Wp, int Cout, int Cin);
int dm4d_conv_up2x_nhwc_bf16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wp, void* Y, int Cout,
                             const void* bias);

/* Direct (fp32 FMA) convolution for thin layers, NHWC: the PoseEncoder's conv stack
 *   (pose_encoder.py:14-31: 3x3 stride 1 / 4x4 stride 2, padding 1, 3..64 input channels, SiLU after each).
 *   X [B,H,W,Cin], Wt [Cout][ksize*ksize][Cin], Y [B,Ho,Wo,Cout]; Cin, Cout multiples of 4 (zero-pad the 3-channel
 *   image and the first layer's filters); ksize <= 4; Ho = (H + 2*pad - ksize)/stride + 1.                        */
int dm4d_conv2d_direct_nhwc_bf16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt,
                                 const void* bias, void* Y, int Ho, int Wo, int Cout, int ksize, int stride, int pad,
                                 int apply_silu);
/* The same chain on fp32 tensors (image, filters, bias, result all fp32; nothing rounded to bf16): the PoseEncoder of the parity
 * precision (see "Parity precision" below).                                                                              */
int dm4d_conv2d_direct_nhwc_f32(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt,
                                const void* bias, void* Y, int Ho, int Wo, int Cout, int ksize, int stride, int pad,
                                int apply_silu);

/* GroupNorm (+SiLU) over [X1 | X2] (channel concat, X2 may be NULL), NHWC.
 *   replaces nn.GroupNorm + SiLU in ResnetBlock2D.norm1/norm2, TransformerMultiviewModel.norm
 *   (transformer_multiview.py:43-45), conv_norm_out (unet_multiview_condition.py:590-592).
 *   ws: fp32 scratch of dm4d_groupnorm_ws_bytes(B, HW) bytes.  Statistics in fp32.              */
size_t dm4d_groupnorm_ws_bytes(int B, int HW, int groups);
int dm4d_groupnorm_nhwc_bf16(void* stream, const void* X1, int C1, const void* X2, int C2, int B, int HW, int groups,
                             float eps, const void* gamma, const void* beta, void* Y, int apply_silu, void* ws);

/* LayerNorm over the last dim.  replaces BasicTransformerBlock.norm1/norm3 (attention.py:49,129) */
int dm4d_layernorm_bf16(void* stream, const void* X, int64_t ldx, const void* gamma, const void* beta, void* Y,
                        int64_t ldy, int M, int C, float eps);

/* Self-attention, head_dim 64, no mask: O = softmax(Q K^T * scale) V per (batch, head).
 *   replaces AttnProcessor2_0 / F.scaled_dot_product_attention reached from attention.py:73-78;
 *   the "(b t) hw c -> b (t hw) c" frame folding (attention.py:69-71,81-83) is expressed by
 *   batch = B / num_frames, L = num_frames * HW on the SAME memory (token-major rows).
 *   Q/K/V/O rows are tokens; element (b, t, h, d) lives at ptr[(b*L + t)*ld + h*64 + d].
 *   Pointers 16-byte aligned, row strides multiples of 8 elements.                                    */
int dm4d_attention_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq, int64_t ldk,
                        int64_t ldv, int64_t ldo, int batch, int heads, int L, float scale);

/* Same kernel with separate query / key lengths: queries are the Lq local tokens of this rank, keys/values the Lk
 *   tokens gathered from every rank (in-window frame sharding).  Element (b, t) of K/V lives at
 *   ptr[(b*Lk + t)*ld + h*64 + d]; of Q/O at ptr[(b*Lq + t)*ld + h*64 + d].                                        */
int dm4d_attention_kv_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq, int64_t ldk,
                           int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk, float scale);

/* Same attention for a Q that already carries the softmax scale: Q' = Q * scale * log2(e)  (DM4D_LOG2E), i.e.
 *   O = softmax2(Q' K^T) V with softmax2 in base 2.  The model folds the factor into the to_q rows of the fused
 *   QKV weight when the checkpoint is loaded (attention.py:73-78 computes to_q(x) then SDPA's q*scale), so the
 *   scale costs no instruction at all: the kernel starts its QK^T accumulator from -rowmax and P = exp2(S).
 *   Layout, strides and Lq/Lk as in dm4d_attention_kv_bf16.                                                    */
#define DM4D_LOG2E 1.4426950408889634
int dm4d_attention_qscaled_kv_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq,
                                   int64_t ldk, int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk);

/* Generic-head-dim attention pieces for the VAE mid block (AutoencoderKL mid_block.attentions.0: single head, d = 512,
 *   reached from pipeline_diffuman4d.py:52,65): P = softmax(S * scale) per row, P in bf16.  The f32in form takes the
 *   logits as dm4d_gemm_bf16(..., DM4D_EPI_F32OUT) leaves them (SDPA keeps its logits in fp32); lds, ldp multiples
 *   of 4 and >= N there, any N: columns behind N of a padded row are neither read nor written.                      */
int dm4d_softmax_rows_bf16(void* stream, const void* S, int64_t lds, void* P, int64_t ldp, int M, int N, float scale);
int dm4d_softmax_rows_f32in_bf16(void* stream, const float* S, int64_t lds, void* P, int64_t ldp, int M, int N, float scale);

/* sinusoidal timestep embedding, diffusers get_timestep_embedding (unet_multiview_condition.py:494):
 *   out[b, :] = [cos(t*f) | sin(t*f)] (flip_sin_to_cos) in bf16, t given as fp32                  */
int dm4d_timestep_embedding_bf16(void* stream, const float* t, void* out, int B, int dim, int flip_sin_to_cos,
                                 float freq_shift);

int dm4d_silu_bf16(void* stream, const void* X, void* Y, int64_t n);

/* Model-input assembly for one window (pipeline_diffuman4d.py:375-395, 348-357):
 *   out [cfg*F, HW, cpad] NHWC: channels [latent 4 | plucker 6 | skeleton-latent 4 (opt) | mask 1 | 0...]
 *   cond rows (is_cond[f] != 0) take the clean image latents (and +1.0 in the negative half);
 *   negative half: plucker 0, skeleton -1.  Also performs the reference's aliasing side effect
 *   (SURVEY 8a P-4 iv): latents[cond rows] <- image latents, in place.
 *   frame_idx (int32 [F], may be NULL): window frame f reads row frame_idx[f] of the task-level
 *   tensors latents/pv_lat/plucker/skel/mask -- the x[window] gathers of :522-531 without copies.     */
int dm4d_pack_model_input_bf16(void* stream, void* latents, const void* pv_lat, const void* plucker, const void* skel,
                               const void* mask, const int32_t* is_cond, const int32_t* frame_idx, void* out, int F,
                               int HW, int cpad, int use_cfg);

/* CFG combine + per-latent DDIM step (pipeline_diffuman4d.py:408-422), one launch for the window:
 *   eps = u + s (c - u);  x <- sqrt(a_prev) x0 + sqrt(1 - a_prev) eps_hat   for non-cond rows.
 *   noise_pred [cfg*F, HW, ldn]; coef [F,4] fp32 = {sqrt(a_t), sqrt(1-a_t), sqrt(a_prev), sqrt(1-a_prev)}
 *   frame_idx as above: the update is scattered straight into the task-level latents (:543).         */
int dm4d_cfg_ddim_step_bf16(void* stream, void* latents, const void* noise_pred, int64_t ldn, const float* coef,
                            const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg,
                            float guidance_scale, int v_prediction);

/* CFG combine + per-latent LINEAR MULTISTEP step: the update of any scheduler whose step is linear in the sample, the model
 * output and one stored prediction -- DPM-Solver++ of order 1 / 2, where the reference keeps one stateful scheduler object per
 * latent (pipeline_diffuman4d.py:265-271, 420, 500-501):   m = u + s (c - u);  x <- a x + b m + c p;  p <- d x + e m
 * for non-cond rows.  coef [F,8] fp32 = {a, b, c, d, e, -, -, -} per frame (host/scheduler.py::step_rows); x0_prev = the task's
 * stored x0 predictions, same shape and indexing as latents (rows whose c is 0 are not read).                                */
int dm4d_cfg_linear_step_bf16(void* stream, void* latents, void* x0_prev, const void* noise_pred, int64_t ldn,
                              const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg,
                              float guidance_scale);

/* CFG combine + per-latent GENERAL linear multistep step with up to three stored tensors per latent: UniPC (with its corrector) and DEIS,
 * where the reference keeps one stateful scheduler object per latent (pipeline_diffuman4d.py:265-271, 420, 500-501).  coef [F,16] fp32 rows
 * (k0..k10, host/scheduler.py::step_rows):   m = u + s (c - u);   conv = k0 x + k1 m;   xc = k2 x + k3 s3 + k4 s1 + k5 s2 + k6 conv;
 * x <- k7 xc + k8 conv + k9 s1 + k10 s2 + k11 s3;   s3 <- xc, s2 <- s1, s1 <- conv   for non-cond rows.  s1 / s2 / s3: same shape and
 * indexing as latents, ZERO at the start of a sliding_iterative_denoise call (s2, s3 may be NULL).  k12 selects what the stored tensors
 * become: 0 as written, 1 kept unchanged, 2 shifted as a history (s3 <- s2, s2 <- s1, s1 <- conv): PNDM's linear multistep form (PLMS),
 * whose repeated second step re-does the first from the stored sample (host/scheduler.py::PNDMScheduler).                                 */
int dm4d_cfg_multistep_step_bf16(void* stream, void* latents, void* s1, void* s2, void* s3, const void* noise_pred, int64_t ldn,
                                 const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg,
                                 float guidance_scale);

/* VAE posterior sample, DiagonalGaussianDistribution.sample() * scaling_factor
 *   (pipeline_diffuman4d.py:52,55): out[m,c] = (mean + exp(0.5*clamp(logvar,-30,20)) * noise[m,c]) * scale
 *   moments rows hold [mean(C) | logvar(C) | ...] with row stride ldm; noise/out are [M, C].          */
int dm4d_vae_sample_bf16(void* stream, const void* moments, int64_t ldm, const void* noise, void* out, int64_t M, int C,
                         float scale);

/* Y[m, 0:cpad] = [X[m, 0:C] * scale | 0]   (latents / scaling_factor before post_quant_conv, :66)     */
int dm4d_scale_pad_bf16(void* stream, const void* X, int64_t ldx, void* Y, int cpad, int64_t M, int C, float scale);

/* F.interpolate(size=(h,w), mode = bilinear (1) | nearest (0)), fp32 NCHW -> bf16 NHWC
 *   (encode_image_resizing, pipeline_diffuman4d.py:90-100: Pluecker maps and condition masks)        */
int dm4d_resize_nchw_f32_to_nhwc_bf16(void* stream, const float* X, void* Y, int B, int C, int H, int W, int h, int w,
                                      int bilinear);

/* Pluecker ray maps at latent resolution, from the cameras: calc_plucker_embeds(H, W, K, pose) (data/utils/ray_utils.py:101-112,
 *   called at spatem_dataset.py:169-176) followed by F.interpolate(size=(h, w), mode="bilinear") and the cast to bf16
 *   (encode_image_resizing, pipeline_diffuman4d.py:90-100, 218-222) in one launch: no [N, 6, H, W] fp32 map on the host, no
 *   H2D copy of it.  cams [N, 24] fp32 = per frame {K^-1 (9, row major) | R (9) | T (3) | o = -R^T T (3)} with [R | T] =
 *   inverse(camera-to-world pose)[:3]; Y [N, h*w, 6] bf16 = [ray direction | o x direction].                        */
int dm4d_plucker_latent_bf16(void* stream, const float* cams, void* Y, int N, int H, int W, int h, int w);

/* F.interpolate(size=(h, w), mode="bilinear", antialias=True) on `planes` fp32 H x W planes (NCHW with planes = N * C): the down-scale of
 *   the result writer's snapshot mosaic (samplers/utils/sampling_utils.py:70-93 -> torchvision's antialiased resize); PIL's separable
 *   triangle filter of support max(H / h, 1), normalised per axis.                                                            */
int dm4d_resize_aa_nchw_f32(void* stream, const float* X, float* Y, int64_t planes, int H, int W, int h, int w);

/* VaeImageProcessor.postprocess(do_denormalize): (x/2 + 0.5).clamp(0,1), NHWC(ldx) -> NCHW (:282-284)  */
int dm4d_postprocess_images_bf16(void* stream, const void* X, void* Y, int B, int C, int HW, int ldx);

/* Fused feed-forward of a transformer block (attention.py:129-149 `ff(norm3(x)) + x`; diffusers FeedForward with GEGLU:
 * Linear(C -> 2 hidden) -> hidden * gelu(gate) -> Linear(hidden -> C)) in ONE launch:
 *     Out[M, C] = residual + W2 (h * gelu(g)) + b2,   [h | g] = W1 Y' + b1,   Y' = LayerNorm(Y) when ln_gamma / ln_beta are given
 *     (norm3 folded in: Y is then the block's residual stream itself), Y' = Y when both are NULL.
 * The [M, hidden] intermediate stays on the chip.  Same products in the same order, same rounding points (the hidden tensor is
 * rounded to bf16 between the two products) as dm4d_gemm_bf16(GEGLU) followed by dm4d_gemm_bf16(residual): bit-identical.
 * dm4d_ff_geglu_supported(C, hidden) says whether the kernel is built for the shape (C = 320: level 0 of an SD-class UNet);
 * dm4d_ff_geglu_prepare_bf16 makes the per-step packed weight copies once per layer from the checkpoint's tensors:
 *   W1 [2 hidden, C] (rows 0..hidden-1 = hidden half, then the gate half), b1 [2 hidden] or NULL, W2 [C, hidden]
 *   -> W1p [2 hidden * C], b1p [2 hidden], W2p [C * hidden] (caller-owned, bf16).                                         */
int dm4d_ff_geglu_supported(int C, int hidden);
int dm4d_ff_geglu_prepare_bf16(void* stream, const void* W1, const void* b1, const void* W2, void* W1p, void* b1p, void* W2p,
                               int C, int hidden);
int dm4d_ff_geglu_fused_bf16(void* stream, const void* Y, int64_t ldy, const void* ln_gamma, const void* ln_beta, float ln_eps,
                             const void* W1p, const void* b1p, const void* W2p, const void* b2, const void* residual,
                             int64_t ld_res, void* Out, int64_t ldo, int M, int C, int hidden);
/* The same launch with the attention output projection in front of it -- the whole tail of a transformer block
 * (attention.py:88-90: attn1.to_out + residual; :129-149: ff(norm3(h)) + h):
 *     h = A0 Wo^T + bo + X          (A0 [M, C] = the attention output, Wo [C, C] row-major, bo [C] or NULL, X [M, C] = the block's input)
 *     Out = h + W2 (u * gelu(g)) + b2,   [u | g] = W1 LayerNorm(h) + b1
 * h is rounded to bf16 once, exactly as dm4d_gemm_bf16(A0, Wo, bias, residual) leaves it.  Since round 6 it never leaves the registers:
 * LayerNorm is taken from the accumulators it was formed in, and its bf16 values start the accumulators of the second product
 * (Out = (h + b2) + W2 (...)), so nothing of h is stored or re-read.  Same products and rounding points as that GEMM followed by
 * dm4d_ff_geglu_fused_bf16 with LayerNorm; fp32 additions in another order (isolated one-ulp differences of a bf16 rounding; rounds
 * 3-5 kept h in Out between the two halves of the launch and were bit-identical).  Out may not alias A0 or X.                      */
int dm4d_attn_out_ff_geglu_fused_bf16(void* stream, const void* A0, int64_t lda0, const void* Wo, const void* bo, const void* X,
                                      int64_t ldx, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* W1p,
                                      const void* b1p, const void* W2p, const void* b2, void* Out, int64_t ldo, int M, int C,
                                      int hidden);

/* ---- Parity precision -------------------------------------------------------------------------------------------------------
 * north_star (BASELINE.json) asks for decoded RGB within 1e-3 rel-L2 of the reference's fp32 CPU path
 * (pipeline_diffuman4d.py:439-559 run with fp32 modules).  bf16 MFMA operands alone cost 7e-3 on the judged UNet call, so the
 * pipeline has a second arithmetic, selected per model object (host: precision="parity"): tensors BETWEEN kernels are fp32 and every
 * activation that feeds the matrix unit is a TWO-TERM bf16 OPERAND  [hi(C) | lo(C)],  hi = bf16(x), lo = bf16(x - hi)  (16 mantissa
 * bits), multiplied against the checkpoint's bf16 weights duplicated along K:  x W^T = [hi | lo] [W | W]^T  -- the same GEMM /
 * convolution kernels with DM4D_EPI_F32OUT / F32SIDE / SPLITOUT.  The entries below produce and consume those operands.        */

/* fp32 -> operand.  Y[m, :] = planes of act(X[m, :]) * scale over Cp >= C1 + C2 columns (columns behind C1 + C2 are zero):
 *   pattern 0: [hi | lo] (ldy >= 2 Cp)   1: [hi | lo | hi]   2: [hi | hi | lo] (ldy >= 3 Cp; the three-term product of two
 *   activations, A in pattern 1 against B in pattern 2 = hi hi + lo hi + hi lo: VAE mid-block attention).
 *   X1 element (m, c) at X1[m * row_stride1 + c * col_stride1] (a transposed read when col_stride1 != 1); X2 (optional second
 *   source = channel concat of the up-block skip, unet_multiview_blocks.py:667) at X2[m * row_stride2 + c].  act_silu: SiLU first. */
int dm4d_split_f32(void* stream, const float* X1, int64_t row_stride1, int64_t col_stride1, int C1, const float* X2,
                   int64_t row_stride2, int C2, void* Y, int64_t ldy, int64_t M, int Cp, int act_silu, float scale, int pattern);

/* GroupNorm (+SiLU) as dm4d_groupnorm_nhwc_bf16, fp32 NHWC in (X2 optional), operand out: Y [B*HW, 2 (C1 + C2)].
 *   Statistics are summed in fp64; ws: dm4d_groupnorm_f32_ws_bytes(B, HW, groups) bytes.                                          */
size_t dm4d_groupnorm_f32_ws_bytes(int B, int HW, int groups);
int dm4d_groupnorm_nhwc_f32_split(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups,
                                  float eps, const void* gamma, const void* beta, void* Y, int apply_silu, void* ws);

/* LayerNorm as dm4d_layernorm_bf16, fp32 in, operand out: Y [M, ldy >= 2 C].                                                      */
int dm4d_layernorm_f32_split(void* stream, const float* X, int64_t ldx, const void* gamma, const void* beta, void* Y, int64_t ldy,
                             int M, int C, float eps);

/* P = softmax(S * scale) per row of fp32 logits -> three planes [p_hi | p_lo | p_hi], each Np >= N columns wide (columns N .. Np-1
 *   zero: a key axis padded to the GEMM's K granularity), ldp >= 3 Np.                                                           */
int dm4d_softmax_rows_f32_split(void* stream, const float* S, int64_t lds, void* P, int64_t ldp, int M, int N, int Np, float scale);

/* dm4d_attention_kv_bf16 on two-term Q / K / V: the hi plane at the pointer, the lo plane `*_lo` ELEMENTS behind it (the planes
 *   dm4d_gemm_bf16(DM4D_EPI_SPLITOUT) leaves for a fused QKV projection).  S = Kh Qh + Kh Ql + Kl Qh, fp32 running-max softmax,
 *   O = Vh Ph + Vl Ph + Vh Pl; O is written as an operand (hi plane at O, lo plane o_lo elements behind it).  head_dim 64.          */
int dm4d_attention_split_bf16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq, int64_t ldk,
                              int64_t ldv, int64_t ldo, int64_t q_lo, int64_t k_lo, int64_t v_lo, int64_t o_lo, int batch,
                              int heads, int Lq, int Lk, float scale);

/* fp32-tensor forms of the small kernels around the UNet / VAE calls (same arithmetic, no rounding on the way out):
 *   dm4d_pack_model_input_f32_split writes the operand of conv_in, rows [hi(cpad) | lo(cpad)].                                    */
int dm4d_timestep_embedding_f32(void* stream, const float* t, float* out, int B, int dim, int flip_sin_to_cos, float freq_shift);
int dm4d_pack_model_input_f32_split(void* stream, float* latents, const float* pv_lat, const float* plucker, const float* skel,
                                    const float* mask, const int32_t* is_cond, const int32_t* frame_idx, void* out, int F, int HW,
                                    int cpad, int use_cfg);
int dm4d_cfg_ddim_step_f32(void* stream, float* latents, const float* noise_pred, int64_t ldn, const float* coef,
                           const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg, float guidance_scale,
                           int v_prediction);
int dm4d_cfg_linear_step_f32(void* stream, float* latents, float* x0_prev, const float* noise_pred, int64_t ldn, const float* coef,
                             const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg, float guidance_scale);
int dm4d_cfg_multistep_step_f32(void* stream, float* latents, float* s1, float* s2, float* s3, const float* noise_pred, int64_t ldn,
                                const float* coef, const int32_t* is_cond, const int32_t* frame_idx, int F, int HW, int use_cfg,
                                float guidance_scale);
int dm4d_vae_sample_f32(void* stream, const float* moments, int64_t ldm, const float* noise, float* out, int64_t M, int C, float scale);
int dm4d_resize_nchw_f32_to_nhwc_f32(void* stream, const float* X, float* Y, int B, int C, int H, int W, int h, int w, int bilinear);
int dm4d_plucker_latent_f32(void* stream, const float* cams, float* Y, int N, int H, int W, int h, int w);
int dm4d_postprocess_images_f32(void* stream, const float* X, float* Y, int B, int C, int HW, int ldx);
int dm4d_nhwc_to_nchw_f32(void* stream, const float* X, float* Y, int B, int C, int HW, int ldx);

/* ---- fp16 precision ---------------------------------------------------------------------------------------------------------
 * The arithmetic that meets north_star's 1e-3 on decoded RGB at ONE MFMA per product (host: precision="fp16"; round 5).  Measured on
 * the judged UNet call (tools/error_budget.py): rounding every MFMA operand to bf16 costs 7.0e-3, rounding it to IEEE fp16 (three more
 * mantissa bits) 8.5e-4.  So tensors BETWEEN kernels are fp32 as in the parity precision, and every activation that feeds the matrix
 * unit is rounded ONCE to fp16 by the kernel that produces it (GroupNorm / LayerNorm / GEGLU / the QKV projection / attention), against
 * the checkpoint's weights held in fp16 (a bf16 or fp16 checkpoint's weights are exact in fp16 -- except bf16 values below 2^-14, which
 * keep an absolute error below 2^-25).  It is also the faithful form of the reference's torch_dtype "fp16"
 * (sampling_utils.py:27-29), with fp32 instead of fp16 storage between operators.  Same tile geometries and K order as the fast
 * precision; same MFMA rate (v_mfma_f32_32x32x16_f16).                                                                            */

/* dm4d_gemm_bf16 with fp16 A / A2 / W / bias.  flags: DM4D_EPI_GEGLU, DM4D_EPI_SILU, DM4D_EPI_F32OUT (C is float*), DM4D_EPI_F32SIDE
 *   (rowbias / residual are float*, else fp16).  Without F32OUT C is fp16 [M, ldc]: the operand of the next contraction.
 *   scale_cols / col_scale: output columns [0, scale_cols) (a multiple of 8) are multiplied by col_scale in fp32 before the one
 *   rounding -- the to_q third of a fused QKV projection takes scale * log2(e) this way (attention.py:73-78), so that the attention
 *   kernel needs no per-score multiply and Q is still rounded once.                                                               */
int dm4d_gemm_f16(void* stream, const void* A, int64_t lda, const void* A2, int64_t lda2, int K1, const void* W, int64_t ldw,
                  void* C, int64_t ldc, int M, int N, int K, const void* bias, const void* rowbias, int64_t ld_rowbias,
                  int rows_per_rowbias, const void* residual, int64_t ld_res, unsigned flags, float out_scale, int scale_cols,
                  float col_scale);

/* dm4d_conv3x3_nhwc_bf16_ws with fp16 X / Wt / bias; flags: DM4D_EPI_F32OUT, DM4D_EPI_F32SIDE.  ws as for the bf16 entry
 *   (dm4d_conv3x3_ws_bytes; NULL runs the un-split kernels).                                                                      */
int dm4d_conv3x3_nhwc_f16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wt, void* Y, int Ho, int Wo,
                          int Cout, int stride, int pad, int upsample, const void* bias, const void* rowbias, int64_t ld_rowbias,
                          const void* residual, int64_t ld_res, float out_scale, unsigned flags, void* ws, size_t ws_bytes);

/* dm4d_conv_up2x_prepare_bf16 / dm4d_conv_up2x_nhwc_bf16 (Upsample2D as four 2x2 phase convolutions) with fp16 weights and input: the
 *   phase kernels are sums of up to four taps taken in fp32 and rounded ONCE to fp16 (relative 2^-12 per weight; the model-level effect is
 *   inside the precision's budget, tests/modelcheck.py fp16_unet_sd21_*).  flags: DM4D_EPI_F32OUT (Y is float*), else Y is fp16.          */
int dm4d_conv_up2x_prepare_f16(void* stream, const void* W, void* Wp, int Cout, int Cin);
int dm4d_conv_up2x_nhwc_f16(void* stream, const void* X, int B, int H, int W, int Cin, const void* Wp, void* Y, int Cout,
                            const void* bias, unsigned flags);

/* fp32 -> fp16 operand: Y[m, :Cp] = fp16(act(X[m, :]) * scale), columns behind C1 + C2 zero (dm4d_split_f32's addressing: second
 *   source = the up-block channel concat, col_stride1 != 1 = a transposed read).                                                  */
int dm4d_to_f16_f32(void* stream, const float* X1, int64_t row_stride1, int64_t col_stride1, int C1, const float* X2,
                    int64_t row_stride2, int C2, void* Y, int64_t ldy, int64_t M, int Cp, int act_silu, float scale);

/* GroupNorm (+SiLU) / LayerNorm / row softmax with fp32 input and ONE fp16 plane out (gamma, beta fp16): Y [B*HW, C1 + C2] /
 *   Y [M, ldy >= C] / P [M, ldp >= Np] (columns N .. Np-1 zero).  GroupNorm statistics: fp32 shifted sums (the fast precision's) on the
 *   vectorised path (channel counts that are multiples of 8; ws = float[B][chunks][groups][2]), fp64 only in the `_general` fall-back kernels;
 *   dm4d_groupnorm_f32_ws_bytes is an upper bound for both.  fp16 outputs saturate at +-65504 (they never become inf).              */
int dm4d_groupnorm_nhwc_f32_f16(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups, float eps,
                                const void* gamma, const void* beta, void* Y, int apply_silu, void* ws);
/* ... and the same GroupNorm with a SECOND output Yraw [B*HW, C1 + C2] = fp16 of the un-normalised input (the channel concat of X1 | X2):
 *   the operand of a resnet's 1x1 shortcut convolution, which reads the tensor its first GroupNorm reads (unet_multiview_blocks.py:667 +
 *   ResnetBlock2D.conv_shortcut), without a pass of its own over the fp32 tensor.  C1, C2 multiples of 8, 16-byte aligned tensors.  */
int dm4d_groupnorm_nhwc_f32_f16_raw(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups, float eps,
                                    const void* gamma, const void* beta, void* Y, void* Yraw, int apply_silu, void* ws);
/* ... and with an fp16 INPUT (one or two sources): the activation a convolution already left in fp16 -- conv1 of a resnet, whose only
 *   reader is norm2 (resnet.py ResnetBlock2D.forward: conv1 -> + temb -> norm2) -- read at two bytes per element.  ws as above.    */
int dm4d_groupnorm_nhwc_f16_f16(void* stream, const void* X1, int C1, const void* X2, int C2, int B, int HW, int groups, float eps,
                                const void* gamma, const void* beta, void* Y, int apply_silu, void* ws);
int dm4d_layernorm_f32_f16(void* stream, const float* X, int64_t ldx, const void* gamma, const void* beta, void* Y, int64_t ldy, int M,
                           int C, float eps);
int dm4d_softmax_rows_f32_f16(void* stream, const float* S, int64_t lds, void* P, int64_t ldp, int M, int N, int Np, float scale);
/* The two entries above take any channel count / alignment: shapes that are not 16-byte vectors of eight channels are forwarded to these
 *   general kernels (fp64 GroupNorm statistics, one or four channels per thread; one wave per LayerNorm row), same arguments.           */
int dm4d_groupnorm_f32_f16_general(void* stream, const float* X1, int C1, const float* X2, int C2, int B, int HW, int groups, float eps,
                                   const void* gamma, const void* beta, void* Y, int apply_silu, void* ws);
int dm4d_layernorm_f32_f16_general(void* stream, const float* X, int64_t ldx, const void* gamma, const void* beta, void* Y, int64_t ldy,
                                   int M, int C, float eps);

/* dm4d_attention_qscaled_kv_bf16 on fp16 Q (carrying scale * log2 e) / K / V -> fp16 O: the same optimistic-softmax loop with
 *   v_mfma_f32_32x32x16_f16; probabilities are kept below 65504 by an offset of 2^-8 on the first tile's maximum and a row-sum check
 *   at 2^16 (attention.hip, "H16").  replaces F.scaled_dot_product_attention via attention.py:73-78.                                */
int dm4d_attention_qscaled_kv_f16(void* stream, const void* Q, const void* K, const void* V, void* O, int64_t ldq, int64_t ldk,
                                  int64_t ldv, int64_t ldo, int batch, int heads, int Lq, int Lk);

/* The tail of a transformer block in one launch, precision "fp16" (dm4d_attn_out_ff_geglu_fused_bf16 above; attention.py:88-90 and
 *   :129-149): A0 [M, C] fp16 attention output, Wo / bo / LayerNorm vectors / b2 fp16, W1p / b1p / W2p = dm4d_ff_geglu_prepare_bf16 of
 *   the fp16 matrices (a permutation of 16-bit words), X [M, C] the fp32 residual stream (ldx in floats, a multiple of 4):
 *       h = A0 Wo^T + bo + X  (fp32, never rounded and never stored: it stays in the accumulators it was formed in),
 *       Out = h + W2 (u * gelu(g)) + b2,  [u | g] = W1 fp16(LayerNorm(h)) + b1,
 *   Out fp32 (out_f32 = 1) or rounded once to fp16 (out_f32 = 0: the operand of the transformer's proj_out).  Same products and
 *   rounding points as dm4d_gemm_f16 (fp32 out + residual), dm4d_layernorm_f32_f16, dm4d_gemm_f16 (GEGLU), dm4d_gemm_f16 (+ fp32
 *   residual); only the ORDER of fp32 additions differs (h enters the output sum first, the LayerNorm row sums are added up in another
 *   order), so the two forms agree to fp32 rounding, not bit for bit (tests/opcheck.py h16_ff_proj_fused_*).  C = 320 only
 *   (dm4d_ff_geglu_supported).  Out may not alias A0 or X.                                                                           */
int dm4d_attn_out_ff_geglu_fused_f16(void* stream, const void* A0, int64_t lda0, const void* Wo, const void* bo, const float* X,
                                     int64_t ldx, const void* ln_gamma, const void* ln_beta, float ln_eps, const void* W1p,
                                     const void* b1p, const void* W2p, const void* b2, void* Out, int64_t ldo, int out_f32, int M, int C,
                                     int hidden);

/* dm4d_pack_model_input_f32_split with ONE fp16 plane out: rows of cpad fp16 channels (the operand of conv_in).                    */
int dm4d_pack_model_input_f32_f16(void* stream, float* latents, const float* pv_lat, const float* plucker, const float* skel,
                                  const float* mask, const int32_t* is_cond, const int32_t* frame_idx, void* out, int F, int HW,
                                  int cpad, int use_cfg);

/* Tuning hook (not part of the operator surface): force one GEMM/conv kernel configuration id for all
 * subsequent launches of this process; 0 restores the built-in heuristic.  Used by tools/gemm_tune.py. */
int dm4d_tune_set_gemm_config(int id);
/* Tuning hook: 0 forces the two-launch GroupNorm (statistics, apply) for every shape; 1 (default) lets maps that fit
 * in registers take the single-launch kernel.  Used by tests/opcheck.py and tests/opbench.py for the A/B.            */
int dm4d_tune_set_groupnorm_resident(int on);

/* layout converters at the pipeline boundary (NCHW <-> NHWC, any C) */
int dm4d_nchw_to_nhwc_bf16(void* stream, const void* X, void* Y, int B, int C, int HW, int cpad);
int dm4d_nhwc_to_nchw_bf16(void* stream, const void* X, void* Y, int B, int C, int HW, int ldx);

#ifdef __cplusplus
}
#endif
#endif /* DM4D_H */
