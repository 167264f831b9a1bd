/* segk -- C ABI of the MI355X (gfx950) segmentation hot-path kernels.
 *
 * This is the drop-in boundary of the project: a shared library
 * (image_segmentation_amd/csrc/libsegk.so) with plain-C entry points -- raw device pointers, ints and a
 * HIP stream handle; no torch/C++ types cross it.  The reference (in5omnia/Image_Segmentation) has no
 * FFI of its own: its hot path is a chain of stock torch.nn layers.  Each entry below names the
 * reference layer(s) it replaces (file:line under the reference repo) so a maintainer can bind it from
 * any host (ctypes stub in INTEGRATION.md; image_segmentation_amd/_lib.py is the binding we ship).
 *
 * Conventions
 *  - every function returns 0 on success, a negative code on failure; segk_last_error() returns a
 *    thread-local message.  Nothing throws, allocates device memory, or synchronises the device:
 *    all launches are asynchronous on `stream` (graph-capture safe); all buffers are caller-owned.
 *  - activations are NHWC, channel count padded to a multiple of 32 ("Cp"); padded channels are zero.
 *  - dtype: SEGK_F32 (exact fp32 MFMA, parity mode) or SEGK_BF16 (bf16 storage/MFMA, fp32 accumulate).
 *  - parameters and their gradients stay in the reference layout and fp32 (OIHW conv weights, IOHW
 *    transposed-conv weights, per-channel vectors); segk_pack_* convert them to the MFMA layout.
 */
#ifndef SEGK_H
#define SEGK_H
#include <stddef.h>
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define SEGK_F32 0
#define SEGK_BF16 1
#define SEGK_MAX_CLASSES 8

typedef void* segk_stream_t; /* hipStream_t */

/* ABI version and the number of entry points this header declares: segk_version() / segk_entry_count() of a library
 * must equal them (image_segmentation_amd/_lib.py refuses a library whose values differ from the table it binds) */
#define SEGK_ABI_VERSION 310
#define SEGK_ENTRY_COUNT 69
int segk_version(void);
int segk_entry_count(void);
/* first 16 hex digits of the sha256 over the sources this library was built from (image_segmentation_amd/build.py:
 * source_hash) -- lets a host check that a shipped libsegk.so matches the sources beside it */
const char* segk_build_id(void);
const char* segk_last_error(void);

/* ---- layout ------------------------------------------------------------------------------------ */
/* NCHW fp32 [B,C,H,W] -> NHWC dtype [B,H,W,Cp] (zero padded).  Input edge of model(X), training.py:45-46 */
int segk_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cp, int dtype, segk_stream_t s);
/* NHWC dtype [B,H,W,Cp] -> NCHW fp32 [B,C,H,W] */
int segk_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int Cp, int dtype, segk_stream_t s);

/* Conv2d weight OIHW fp32 [Cout][CA+CB][taps] -> MFMA layout [Kp/CH][taps][Np][CH] (CH = 16 fp32 / 32 bf16).
 * The input channels may come from two NHWC sources (the skip concat of unet.py:63 / clipunet.py:102):
 * CA/CB logical, CAp/CBp padded.  mode 0: forward weights; mode 1: data-gradient weights (flipped taps,
 * in/out swapped).  taps = 9 (3x3) or 1 (1x1).  dst holds (CAp+CBp)*taps*Coutp elements. */
int segk_pack_conv_weight(const float* w, void* dst, int Cout, int CA, int CB, int Coutp, int CAp, int CBp,
                          int taps, int mode, int dtype, segk_stream_t s);
/* 3x3 weights: the forward (mode 0) and data-gradient (mode 1) layouts of segk_pack_conv_weight in one pass over the
 * fp32 parameter (training re-packs after every optimizer step); dst_dgrad may be NULL */
int segk_pack_conv3x3_both(const float* w, void* dst_fwd, void* dst_dgrad, int Cout, int CA, int CB, int Coutp, int CAp,
                           int CBp, int dtype, segk_stream_t s);
/* Up to 64 packed copies refreshed by ONE launch (everything a model re-packs after an optimizer step).
 * table: device array of n 64-byte entries { const float* w; void* dst_fwd; void* dst_dgrad; int32 Cout, CA, CB, Coutp,
 * CAp, CBp; int32 block0; int32 kind; int32 pad[2] } sorted by block0 = first block of the tensor; total_blocks = the sum
 * of the entries' block counts.  The table must stay valid until the launch has run.
 *   kind 0: segk_pack_conv3x3_both (blocks (CAp+CBp)/32 * Coutp/32);
 *   kind 1: segk_pack_convt_weight, mode 0 into dst_fwd and mode 1 into dst_dgrad (may be NULL), CA = Cin, CAp = Cinp
 *           (blocks ceil(Cinp*4*Coutp / segk_pack_convt_chunk()));
 *   kind 2: bias w [Cout] -> dst_fwd fp32 [reps][Coutp] (1 block), reps = CA, 0 meaning 4: the bias4 operand of
 *           segk_convt2x2_fwd; reps 1: the padded bias operand of segk_conv1x1;
 *   kind 3: Conv2d 1x1 weight w [Cout][CA] -> segk_pack_conv_weight(taps 1) mode 0 into dst_fwd and mode 1 into dst_dgrad
 *           (may be NULL) (blocks ceil(CAp*Coutp / segk_pack_convt_chunk())). */
int segk_pack_multi(const void* table, int n, int total_blocks, int dtype, segk_stream_t s);
int segk_pack_convt_chunk(void);
/* ConvTranspose2d(k=2,s=2) weight IOHW fp32 [Cin][Cout][2][2] -> MFMA layout; mode 0 forward, 1 data-gradient */
int segk_pack_convt_weight(const float* w, void* dst, int Cin, int Cout, int Cinp, int Coutp, int mode, int dtype,
                           segk_stream_t s);

/* ---- Conv2d 3x3 pad 1 (unet/unet.py:16,19; clip/clipunet.py:87,90), forward and data-gradient -----
 * out[B,H,W,CO1] (| out2[B,H,W,CO2]) = conv3x3( [srcA | srcB] , wpacked ) (+ bias).
 * scale/shift != NULL: the producer layer's BatchNorm+ReLU is applied to srcA on load (fused prologue).
 * stats != NULL: per-tile per-channel (sum, sumsq) partials [segk_conv_tiles()][CO1+CO2][2] for
 * training-mode BatchNorm (finish with segk_bn_finalize).  For the data gradient pass mode-1 weights. */
/* Second conv of a DoubleConv block with its fused BN+ReLU prologue, additionally writing the prologue's result
 * act_out[B,H,W,CA] = relu(srcA * scale + shift) (the hidden activation nn.Sequential would have materialised,
 * unet.py:17-18): the weight-gradient pass then reads it directly instead of re-deriving it per fragment.
 * Only for layers where segk_conv_writes_act_q(CA, CO, dtype) != 0; act_out may be NULL. */
int segk_conv_writes_act_q(int Cin, int Cout, int dtype);
int segk_conv3x3_act(const void* srcA, const void* wpacked, const float* scale, const float* shift, void* out,
                     void* act_out, float* stats, int B, int H, int W, int CA, int CO, int dtype, segk_stream_t s);

/* The stem: Conv2d(Cin <= 3, 64, k=3, p=1) (unet/unet.py:16, first conv of down1) applied directly to the NCHW fp32 input
 * batch of utils/training.py:45 -- an im2col GEMM with K = 9 Cin <= 27 (one MFMA step), bf16 operands rounded exactly as the
 * layout pass + packed weights would round them, bias-free like segk_conv3x3.  x_nchw [B,Cin,H,W] fp32, w_oihw the fp32
 * parameter itself [64][Cin][3][3]; z NHWC bf16 [B,H,W,64]; stats (may be NULL): segk_stem3x3_rows() rows of [64][2]
 * partial sums for segk_bn_finalize; x_nhwc (may be NULL): the padded NHWC bf16 copy of the input [B,H,W,32] the
 * weight-gradient pass reads (what segk_nchw_to_nhwc would have written).  segk_stem3x3_rows() is 0 where the kernel does
 * not apply (bf16 only, W % 16 == 0, Cout == 64): use segk_nchw_to_nhwc + segk_conv3x3 there. */
int segk_stem3x3_rows(int B, int H, int W, int Cin, int Cout, int dtype);
int segk_stem3x3(const float* x_nchw, const float* w_oihw, void* z, void* x_nhwc, float* stats, int B, int H, int W, int Cin,
                 int Cout, int dtype, segk_stream_t s);
/* ... and its weight gradient from the same NCHW fp32 batch and dz NHWC bf16 [B,H,W,64]: segk_stem3x3_wgrad_slabs() slabs of
 * [64][32] fp32 (column k = ci * 9 + tap, the OIHW order; columns >= 9 Cin are zero), summed by
 * segk_wgrad_reduce(slabs, S, grad, 64, 9 * Cin, 0, 64, 32, 0, 1).  0 slabs: not served (use segk_wgrad on the padded copy). */
int segk_stem3x3_wgrad_slabs(int B, int H, int W, int Cin, int Cout, int dtype);
int segk_stem3x3_wgrad(const float* x_nchw, const void* dz, float* slabs, int B, int H, int W, int Cin, int Cout, int dtype,
                       segk_stream_t s);
int segk_conv_tiles(int B, int H, int W, int Cin, int Cout, int dtype);   /* padded CA+CB and CO1+CO2 of the call */
/* floats to allocate for `stats`: the [tiles][Cp][2] partials plus the scratch segk_bn_finalize reduces through */
int segk_bn_stats_floats(int tiles, int Cp);
int segk_conv3x3(const void* srcA, const void* srcB, const void* wpacked, const float* bias, const float* scale,
                 const float* shift, void* out, void* out2, float* stats, int B, int H, int W, int CA, int CB,
                 int CO1, int CO2, int dtype, segk_stream_t s);
/* Conv2d 1x1 (clip/clipunet.py:84,122): same contract, taps = 1 */
int segk_conv1x1(const void* srcA, const void* wpacked, const float* bias, void* out, int B, int H, int W, int CA,
                 int CO, int dtype, segk_stream_t s);
/* ConvTranspose2d(k=2,s=2) forward (unet.py:59; clipunet.py:83): in [B,H,W,Cin] -> out [B,2H,2W,Cout];
 * bias4 = the layer bias tiled over the four taps (length 4*Cout, zero in padded channels) or NULL */
int segk_convt2x2_fwd(const void* in, const void* wpacked, const float* bias4, void* out, int B, int H, int W,
                      int Cin, int Cout, int dtype, segk_stream_t s);
/* ... its data gradient: dout [B,2H,2W,Cout] -> din [B,H,W,Cin]  (H,W are the INPUT grid) */
int segk_convt2x2_dgrad(const void* dout, const void* wpacked, void* din, int B, int H, int W, int Cin, int Cout,
                        int dtype, segk_stream_t s);

/* ---- weight gradients ---------------------------------------------------------------------------
 * slabs [S][CD][taps][CA+CB] fp32 = split-K partials of  sum_p dz[p][n] * a[p+tap][k]; then
 * segk_wgrad_reduce sums them in fixed order into the reference-layout gradient (fp32, overwritten; the
 * slabs are read only).
 * geo 0: Conv2d 3x3 (grad OIHW [N][CA+CB][9]);  geo 1: Conv2d 1x1;  geo 2: ConvTranspose2d(k=2,s=2) with
 * dz := layer input [B,H,W,CD], srcA := output gradient [B,2H,2W,CA], grad IOHW [CD][CA][2][2].
 * scale/shift: BatchNorm+ReLU prologue on srcA (the conv input is relu(bn(z)) of the previous conv). */
int segk_wgrad_tiles(int B, int H, int W, int geo, int dtype);
/* split-K factor S the host sizes the slab buffer with ([S][CD][taps][CA+CB] fp32) for `tiles` = segk_wgrad_tiles():
 * enough workgroups to fill the chip for this layer's (n, k) tiling, never more slabs than tiles; 0 for invalid input */
int segk_wgrad_split(int tiles, int CD, int CA, int CB, int geo, int dtype);
int segk_wgrad(const void* dz, const void* srcA, const void* srcB, const float* scale, const float* shift,
               float* slabs, const void* zeros64, int S, int B, int H, int W, int CD, int CA, int CB, int geo,
               int dtype, segk_stream_t s);   /* zeros64: >= 64 zero bytes in device memory (halo source of the
                                                  LDS-DMA path); NULL selects the register-staged kernel */
int segk_wgrad_reduce(const float* slabs, int S, float* grad, int N, int CA, int CB, int Np, int CAp, int CBp,
                      int taps, segk_stream_t s);
/* Up to four such reductions in ONE launch -- the two weight gradients of a DoubleConv block (unet.py:16,19), the
 * ConvTranspose2d weight gradient of the Up block around it (unet.py:59) and that layer's bias gradient: jobs is a HOST
 * array (read during the call).  kind 0: segk_wgrad_reduce(src = slabs, S, dst = grad, N, CA, CB, Np, CAp, CBp, taps);
 * kind 1: column sums dst[c] = sum over rows r < S of src[(r * N + CA + c) * 2], c < CB (N = channels per row of the
 * [rows][N][2] BatchNorm-style partials segk_conv3x3 writes: the bias gradient of the ConvTranspose whose output is the
 * second concat operand is the channel sum of the concat data gradient).  Results equal the single launches bit for bit. */
typedef struct segk_reduce_job {
  const float* src;
  float* dst;
  int kind, S, N, CA, CB, Np, CAp, CBp, taps, pad_;
} segk_reduce_job;
int segk_wgrad_reduce_multi(const segk_reduce_job* jobs, int n, segk_stream_t s);

/* ---- BatchNorm2d + ReLU (unet.py:17-18,20-21; clipunet.py:88-89,91-92) ---------------------------
 * training: stats partials -> scale = gamma*rstd, shift = beta - mean*scale, batch mean/rstd saved for
 * backward, running stats updated (momentum, unbiased var; conv_bias only shifts running_mean: the
 * kernels work on the bias-free conv output because the bias cancels inside BatchNorm).
 * eval: scale/shift from running statistics. */
int segk_bn_finalize(const float* stats, int tiles, int Cp, int C, double count, const float* conv_bias,
                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, int training, float* scale, float* shift, float* mean, float* rstd, segk_stream_t s);
/* y = relu(z*scale + shift) over P pixels */
int segk_bn_relu_apply(const void* z, void* y, const float* scale, const float* shift, long P, int Cp, int dtype,
                       segk_stream_t s);
/* ... and the MaxPool2d(2,2) of y in the same pass (unet.py:20-21 followed by :40 of the next Down block):
 * pooled [B,H/2,W/2,Cp]; bit-identical to segk_bn_relu_apply + segk_maxpool2x2_fwd */
int segk_bn_relu_apply_pool(const void* z, void* y, void* pooled, const float* scale, const float* shift, int B, int H, int W,
                            int Cp, int dtype, segk_stream_t s);
/* backward of y = relu(bn(z)): dz (may alias dy) and dgamma/dbeta.  part: segk_bn_bwd_blocks()*Cp*2 floats,
 * coef: 2*Cp floats of scratch. */
int segk_bn_bwd_blocks(long P, int Cp, int dtype);
int segk_bn_relu_bwd(const void* dy, const void* z, void* dz, const float* scale, const float* shift,
                     const float* mean, const float* rstd, long P, int Cp, int C, float* part, float* dgamma,
                     float* dbeta, float* coef, int dtype, segk_stream_t s);

/* per-channel sum over P pixels of an NHWC tensor: the bias gradient of ConvTranspose2d (unet.py:59) and of the
 * 1x1 convs (clipunet.py:84,122).  part: segk_bn_bwd_blocks(P,Cp,dtype)*Cp floats of scratch. */
int segk_channel_sum(const void* x, long P, int Cp, int C, float* part, float* out, int dtype, segk_stream_t s);

/* ---- MaxPool2d(2,2) (unet.py:40) ---------------------------------------------------------------- */
int segk_maxpool2x2_fwd(const void* x, void* y, int B, int H, int W, int Cp, int dtype, segk_stream_t s);
/* dx (+)= route(dy) to the first maximum of each window; accumulate=1 adds into dx (skip gradient) */
int segk_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int Cp, int accumulate,
                        int dtype, segk_stream_t s);

/* Pooling backward of a DoubleConv block's output y = relu(bn(z)) (unet.py:40 behind :20-21) that ALSO accumulates that
 * BatchNorm's backward reductions sum(g), sum(g*xhat) over the complete gradient it writes (xhat recovered from y where the
 * ReLU is active): part holds segk_maxpool_bwd_stat_blocks() rows of [Cp][2] floats, finished by
 * segk_bn_relu_bwd_from_part (finalize + apply, no reduce pass).  segk_maxpool_bwd_stat_blocks returns 0 when the shape
 * is not served (Cp / vector width must be a power of two). */
int segk_maxpool_bwd_stat_blocks(int B, int H, int W, int Cp, int dtype);
int segk_maxpool2x2_bwd_bnstat(const void* x, const void* dy, void* dx, int B, int H, int W, int Cp, int accumulate,
                               const float* scale, const float* shift, const float* mean, const float* rstd, float* part,
                               const void* z, int dtype, segk_stream_t s);
/* z (may be NULL): the block's pre-activation [B,H,W,Cp].  xhat is recovered from x = relu(z*scale+shift) itself except
 * for channels where that is impossible or ill-conditioned (scale == 0, |scale| < |shift|/16): the threads owning such a
 * channel read z and take xhat = (z - mean) * rstd, exactly like segk_bn_relu_bwd's own reduction; with z == NULL those
 * channels get xhat = -mean * rstd (wrong dgamma for them). */
int segk_bn_relu_bwd_from_part(const void* dy, const void* z, void* dz, const float* scale, const float* shift,
                               const float* mean, const float* rstd, long P, int Cp, int C, const float* part, int nb,
                               float* dgamma, float* dbeta, float* coef, int dtype, segk_stream_t s);

/* ---- bilinear resize, align_corners=False (clip/clipunet.py:99-100: skip features 14x14 -> decoder grid) ----
 * x [B,IH,IW,Cp] -> y [B,OH,OW,Cp]; backward is a deterministic gather dy -> dx */
int segk_bilinear_fwd(const void* x, void* y, int B, int IH, int IW, int OH, int OW, int Cp, int dtype,
                      segk_stream_t s);
/* scratch: B*OH*IW*Cp floats for the separable two-pass form (x then y; about 3x the up-sampling factor of work per
 * element instead of its square), or NULL for the single-pass 2-D gather */
int segk_bilinear_bwd(const void* dy, void* dx, float* scratch, int B, int IH, int IW, int OH, int OW, int Cp, int dtype,
                      segk_stream_t s);

/* ---- CLIP vision transformer, frozen feature extractor (clip/clipunet.py:25-46,48-63 drive transformers'
 * CLIPVisionModel -- third party; algorithm: modeling_clip.py CLIPVisionEmbeddings / CLIPEncoderLayer / CLIPAttention /
 * CLIPMLP).  Token tensors are row matrices [Mp][C], Mp = B*T rounded up to 16 rows; the residual stream is fp32. */
/* nn.Linear / patch-projection GEMM on MFMA: out[M][N] = rows[M][K] . W^T (+ bias) (act 1: quick_gelu, CLIPMLP).
 * wpacked = segk_pack_conv_weight(W viewed as [N][K][1][1], taps 1, mode 0); K, N multiples of 32; M multiple of 16. */
int segk_linear(const void* rows, const void* wpacked, const float* bias, void* out, long M, int K, int N, int act,
                int dtype, segk_stream_t s);
/* image NCHW fp32 [B,C,H,W] -> patch rows [B*(H/ps)*(W/ps)][Kp], k = c*ps*ps + i*ps + j (zero beyond C*ps*ps):
 * the im2col of CLIPVisionEmbeddings.patch_embedding (Conv2d(C, D, ps, stride ps, bias=False)) */
int segk_vit_patchify(const float* x, void* rows, int B, int C, int H, int W, int ps, int Kp, int dtype, segk_stream_t s);
/* h[b][t] = LayerNorm_pre( (t == 0 ? class_embedding : proj[b][t-1]) + position_embedding[t] ); h fp32 [B*T][D] */
int segk_vit_embed_ln(const void* proj, const float* cls, const float* pos, const float* gamma, const float* beta,
                      float eps, float* h, int B, int T, int D, int Dp, int dtype, segk_stream_t s);
/* residual add + LayerNorm: h[M][D] += delta[M][Dp] (delta may be NULL); out[M][Dp] = LN(h)*gamma+beta (out may be
 * NULL: add only) -- CLIPEncoderLayer's residual connections fused with the next layer_norm1/2 */
int segk_add_layernorm(float* h, const void* delta, const float* gamma, const float* beta, float eps, void* out, long M,
                       int D, int Dp, int dtype, segk_stream_t s);
/* split-K forms for GEMMs too small to fill 256 CUs (ViT out_proj / fc2 at B = 16: 78 output tiles): `ksplit` partial
 * products out_parts[ksplit][M][N] (bf16, bias on split 0), summed in fixed order by the residual add that follows */
int segk_linear_splitk(const void* rows, const void* wpacked, const float* bias, void* out_parts, long M, int K, int N,
                       int ksplit, int dtype, segk_stream_t s);
int segk_add_layernorm_parts(float* h, const void* delta, int nparts, long part_stride, const float* gamma,
                             const float* beta, float eps, void* out, long M, int D, int Dp, int dtype, segk_stream_t s);
/* CLIPAttention: qkv [B*T][ldq] = [q | k | v] (heads*head_dim each) -> ctx [B*T][ldo] = softmax(q k^T * scale) v per
 * head; head_dim 32 or 64; the head's K and V must fit the 160 KiB LDS (T <= 320 fp32 / 640 bf16 at head_dim 64) */
int segk_attention(const void* qkv, void* ctx, int B, int T, int heads, int head_dim, int ldq, int ldo, float scale,
                   int dtype, segk_stream_t s);
/* drop CLS, residual stream -> NHWC feature grid [B,G,G,Dp] in dtype (clipunet.py:48-51,54-63) */
int segk_vit_tokens_to_grid(const float* h, void* out, int B, int T, int D, int Dp, int dtype, segk_stream_t s);

/* ---- eval-time pre/post-processing on device (utils/utils.py:13-115; training.py:87-99) ---------------------------
 * one image [C,H,W] -> its slot [C,T,T] of the network batch: resize to (nh,nw) + zero padding (utils.py:13-49).
 * mode 0: anti-aliased bilinear = F.interpolate(bilinear, align_corners=False, antialias=True), what torchvision's
 * tensor TF.resize computes from 0.17 on; mode 2: plain two-tap bilinear (antialias=False: torchvision < 0.17 on
 * tensors; the reference pins no version); mode 1: nearest.  elem 0: float32, 1: int64 (labels; mode 1 only). */
int segk_resize_pad(const void* img, void* out, int C, int H, int W, int nh, int nw, int T, int pad_top, int pad_left,
                    int mode, int elem, segk_stream_t s);
/* slot [C,T,T] fp32 -> crop the (nh,nw) window at (pad_top,pad_left) -> [C,oh,ow]: F.interpolate bilinear
 * (align_corners=False; mode 0) or nearest (mode 1) (utils.py:51-75) */
int segk_crop_resize(const float* slot, float* out, int C, int T, int pad_top, int pad_left, int nh, int nw, int oh,
                     int ow, int mode, segk_stream_t s);

/* ---- output head: Conv2d(C, ncls, 1) (unet.py:91,105; clipunet.py:181,187) ----------------------- */
/* y NHWC [B,H,W,Cp] -> logits NCHW fp32 [B,ncls,H,W];  w fp32 [ncls][C], bias [ncls] */
int segk_head_fwd(const void* y, const float* w, const float* bias, float* logits, int B, int H, int W, int Cp,
                  int C, int ncls, int dtype, segk_stream_t s);
int segk_head_part_floats(long P, int Cp);
int segk_head_bwd(const float* dlogits, const void* y, const float* w, void* dy, float* part, float* dw, float* db,
                  int B, int H, int W, int Cp, int C, int ncls, int dtype, segk_stream_t s);

/* segk_head_bwd that also accumulates the BatchNorm backward reductions of the DoubleConv block whose output y the head
 * reads (unet.py:103-105: up4 -> output): bnpart holds segk_head_bwd_blocks(P) rows of [Cp][2] floats for
 * segk_bn_relu_bwd_from_part */
int segk_head_bwd_blocks(long P);
int segk_head_bwd_bnstat(const float* dlogits, const void* y, const float* w, void* dy, float* part, float* dw, float* db,
                         int B, int H, int W, int Cp, int C, int ncls, const float* scale, const float* shift,
                         const float* mean, const float* rstd, float* bnpart, int dtype, segk_stream_t s);

/* The same head reading the block's PRE-ACTIVATION z instead of its output (unet.py:103-105: the output of up4 is
 * consumed by the head alone, so relu(z * scale + shift) -- rounded to `dtype` exactly as segk_bn_relu_apply stores it --
 * is re-formed inside both kernels and never written): forward, and backward writing the gradient of that (virtual)
 * output to dy; bnpart (may be NULL) receives the BatchNorm backward reductions with xhat = (z - mean) * rstd. */
int segk_head_fwd_bn(const void* z, const float* scale, const float* shift, const float* w, const float* bias, float* logits,
                     int B, int H, int W, int Cp, int C, int ncls, int dtype, segk_stream_t s);
int segk_head_bwd_bn(const float* dlogits, const void* z, const float* w, void* dy, float* part, float* dw, float* db,
                     int B, int H, int W, int Cp, int C, int ncls, const float* scale, const float* shift,
                     const float* mean, const float* rstd, float* bnpart, int dtype, segk_stream_t s);

/* ---- per-pixel CrossEntropy + soft Dice (training.py:47; utils/weighted_loss.py:31-98,140-166) ----
 * logits NCHW fp32 [N,C,HW], labels int64 [N,HW].  state (segk_loss_state_floats() floats):
 * [0]=dice_weight*dice+ce_weight*ce, [1]=ce, [2]=dice, rest = saved statistics for backward.
 * ignore_index < 0: none.  class_weights may be NULL. */
int segk_loss_part_floats(long P);
int segk_loss_state_floats(void);
int segk_loss_fwd(const float* logits, const int64_t* labels, const float* class_weights, int N, int C, long HW,
                  int ignore_index, float smooth, float dice_weight, float ce_weight, float* part, float* state,
                  float* loss_out, segk_stream_t s);   /* loss_out (may be NULL): one float, a copy of state[0] (the value the
                                                          nn.Module returns, in a buffer of its own); the block of the
                                                          launch that finishes last turns the partials into the state */
int segk_loss_bwd(const float* logits, const int64_t* labels, const float* class_weights, const float* state,
                  const float* grad_out, int N, int C, long HW, int ignore_index, float dice_weight, float ce_weight,
                  float* dlogits, segk_stream_t s);

/* ---- prompt model (prompt_based/prompt.py:33-56; utils/weighted_loss.py:170-343) ---------------------------
 * remix of the frozen 4-class CLIP-UNet softmax with the sigmoid of the 1-channel mask U-Net, fp32 NCHW:
 * final[0] = 1-m; final[1] = m*(p0+p3); final[2] = m*p1; final[3] = m*p2.  Backward: gradient of the mask logit only
 * (the CLIP branch is frozen, prompt.py:30-31). */
int segk_prompt_mix_fwd(const float* clip_logits, const float* mask_logit, float* final_probs, int N, long HW,
                        segk_stream_t s);
int segk_prompt_mix_bwd(const float* clip_logits, const float* mask_logit, const float* dfinal, float* dmask_logit, int N,
                        long HW, segk_stream_t s);
/* WeightedDiceNLLLoss on class probabilities (apply_softmax=False): soft Dice on the values themselves +
 * NLLLoss(weight, ignore_index) of log(p + eps) (nll_log = 1; prompt.ipynb's stable_log) or of p itself (nll_log = 0).
 * part/state sized like segk_loss_fwd's. */
int segk_prob_loss_fwd(const float* probs, const int64_t* labels, const float* class_weights, int N, int C, long HW,
                       int ignore_index, float smooth, float dice_weight, float nll_weight, int nll_log, float eps,
                       float* part, float* state, float* loss_out, segk_stream_t s);
int segk_prob_loss_bwd(const float* probs, const int64_t* labels, const float* class_weights, const float* state,
                       const float* grad_out, int N, int C, long HW, int ignore_index, float dice_weight,
                       float nll_weight, int nll_log, float eps, float* dprobs, segk_stream_t s);

/* ---- diagnostics (not on the product path) ---------------------------------------------------------
 * The shader clock held under a dense bf16 MFMA load: `blocks` workgroups of four waves (one per SIMD) run `iters` rounds of
 * 16 v_mfma_f32_32x32x16_bf16 (shape 0; shape 1: the same work as 32 v_mfma_f32_16x16x32_bf16) each and write, per wave w, out[2w] = elapsed shader cycles (s_memtime) and out[2w+1] =
 * elapsed ticks of the constant 100 MHz counter (s_memrealtime): clock = out[2w] / out[2w+1] x 100 MHz.  bench.py puts
 * the median into its line so that box-to-box spread is explained by a number. */
int segk_clock_probe(uint64_t* out, int blocks, int iters, int shape, segk_stream_t s);
/* Overwrites every counter of the device's ticket ring with the low 32 bits of `pattern` and waits for the copy.  The ring
 * holds the arrival counters of the kernels that finish a reduction in the launch that produced its partials
 * (segk_bn_finalize above 1024 partial rows, segk_loss_fwd / segk_prob_loss_fwd); the host zeroes the counters a launch will
 * use on that launch's stream right before it, so ANY content is a valid starting state.  Tests use this entry to leave
 * behind what an aborted launch or a stray store would (tests/test_gpu_kernels.py). */
int segk_debug_poison_tickets(uint64_t pattern, segk_stream_t s);

/* ---- metric: argmax + confusion matrix (utils/MetricsHistory.py:65-75) ---------------------------
 * M[pred*8 + label] += count (uint64, caller zeroes); TP/FP/FN/TN follow on the host. */
int segk_confusion(const float* logits, const int64_t* labels, int N, int C, long HW, uint64_t* M, segk_stream_t s);

#ifdef __cplusplus
}
#endif
#endif
