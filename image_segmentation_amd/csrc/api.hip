// extern "C" entry points declared in include/segk.h: argument validation + dispatch to the kernels.
#include "common.hpp"
#include "segk_internal.h"
#include "../../include/segk.h"

thread_local char g_segk_err[512] = "";

// impl functions defined in the kernel translation units
int segk_bn_finalize_impl(const float*, int, int, int, double, const float*, const float*, const float*, float*, float*,
                          float, float, int, float*, float*, float*, float*, hipStream_t);
int segk_bn_relu_apply_impl(const void*, void*, const float*, const float*, long, int, int, hipStream_t);
int segk_bn_relu_apply_pool_impl(const void*, void*, void*, const float*, const float*, int, int, int, int, int, hipStream_t);
int segk_bn_bwd_impl(const void*, const void*, void*, const float*, const float*, const float*, const float*, long, int,
                     int, float*, float*, float*, float*, int, hipStream_t);
int segk_channel_sum_impl(const void*, long, int, int, float*, float*, int, hipStream_t);
int segk_maxpool_fwd_impl(const void*, void*, int, int, int, int, int, hipStream_t);
int segk_maxpool_bwd_impl(const void*, const void*, void*, int, int, int, int, int, int, hipStream_t);
int segk_maxpool_bwd_bnstat_impl(const void*, const void*, void*, int, int, int, int, int, const float*, const float*,
                                 const float*, const float*, float*, const void*, int, hipStream_t);
int segk_bn_bwd_from_part_impl(const void*, const void*, void*, const float*, const float*, const float*, const float*, long,
                               int, int, const float*, int, float*, float*, float*, int, hipStream_t);
int segk_nchw_to_nhwc_impl(const float*, void*, int, int, int, int, int, int, hipStream_t);
int segk_nhwc_to_nchw_impl(const void*, float*, int, int, int, int, int, int, hipStream_t);
int segk_pack_conv_weight_impl(const float*, void*, int, int, int, int, int, int, int, int, int, hipStream_t);
int segk_pack_convt_weight_impl(const float*, void*, int, int, int, int, int, int, hipStream_t);
int segk_pack_conv3x3_both_impl(const float*, void*, void*, int, int, int, int, int, int, int, hipStream_t);
int segk_pack_multi_impl(const void*, int, int, int, hipStream_t);
int segk_pack_convt_chunk_impl();
int segk_wgrad_reduce_impl(const float*, int, float*, int, int, int, int, int, int, int, hipStream_t);
int segk_wgrad_reduce_multi_impl(const segk_reduce_job*, int, hipStream_t);
int segk_head_fwd_impl(const void*, const float*, const float*, float*, int, int, int, int, int, int, const float*, const float*,
                       int, hipStream_t);
int segk_head_bwd_impl(const float*, const void*, const float*, void*, float*, float*, float*, int, int, int, int, int,
                       int, const float*, const float*, const float*, const float*, float*, int, int, hipStream_t);
int segk_head_blocks_q(long);
int segk_loss_fwd_impl(const float*, const long long*, const float*, int, int, long, int, float, float, float, float*,
                       float*, float*, int, int, float, hipStream_t);
int segk_loss_bwd_impl(const float*, const long long*, const float*, const float*, const float*, int, int, long, int,
                       float, float, float*, int, int, float, hipStream_t);
int segk_prompt_mix_impl(const float*, const float*, const float*, float*, int, long, hipStream_t);
int segk_confusion_impl(const float*, const long long*, int, int, long, unsigned long long*, hipStream_t);

int segk_clock_probe_impl(unsigned long long*, int, int, int, hipStream_t);
int segk_debug_poison_tickets_impl(unsigned long long, hipStream_t);

int segk_device_index() {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) dev = 0;
  return dev < SEGK_MAX_DEVICES ? dev : SEGK_MAX_DEVICES - 1;
}
int segk_num_cus() {
  static int n[SEGK_MAX_DEVICES] = {};    // racing first calls write the same value
  const int dev = segk_device_index();
  if (n[dev] == 0) {
    hipDeviceProp_t p;
    int v = 0;
    if (hipGetDeviceProperties(&p, dev) == hipSuccess) v = p.multiProcessorCount;
    n[dev] = v >= 8 ? v : 256;
  }
  return n[dev];
}

static void fill_tiles(ConvArgs&) {}   // tile geometry is chosen per kernel configuration by the launcher

extern "C" {

#ifndef SEGK_BUILD_ID
#define SEGK_BUILD_ID "unknown"
#endif
int segk_version(void) { return SEGK_ABI_VERSION; }
int segk_entry_count(void) { return SEGK_ENTRY_COUNT; }
int segk_debug_poison_tickets(uint64_t pattern, segk_stream_t s) {
  return segk_debug_poison_tickets_impl((unsigned long long)pattern, (hipStream_t)s);
}
int segk_clock_probe(uint64_t* out, int blocks, int iters, int shape, segk_stream_t s) {
  return segk_clock_probe_impl((unsigned long long*)out, blocks, iters, shape, (hipStream_t)s);
}
const char* segk_build_id(void) { return SEGK_BUILD_ID; }
const char* segk_last_error(void) { return g_segk_err; }

int segk_nchw_to_nhwc(const float* src, void* dst, int B, int C, int H, int W, int Cp, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "nchw_to_nhwc: bad dtype %d", dtype);
  return segk_nchw_to_nhwc_impl(src, dst, B, C, H, W, Cp, dtype, (hipStream_t)s);
}
int segk_nhwc_to_nchw(const void* src, float* dst, int B, int C, int H, int W, int Cp, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "nhwc_to_nchw: bad dtype %d", dtype);
  return segk_nhwc_to_nchw_impl(src, dst, B, C, H, W, Cp, dtype, (hipStream_t)s);
}
int segk_pack_conv_weight(const float* w, void* dst, int Cout, int CA, int CB, int Coutp, int CAp, int CBp, int taps,
                          int mode, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "pack_conv_weight: bad dtype %d", dtype);
  return segk_pack_conv_weight_impl(w, dst, Cout, CA, CB, Coutp, CAp, CBp, taps, mode, dtype, (hipStream_t)s);
}
int segk_pack_conv3x3_both(const float* w, void* dst_fwd, void* dst_dgrad, int Cout, int CA, int CB, int Coutp, int CAp,
                            int CBp, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "pack_conv3x3_both: bad dtype %d", dtype);
  return segk_pack_conv3x3_both_impl(w, dst_fwd, dst_dgrad, Cout, CA, CB, Coutp, CAp, CBp, dtype, (hipStream_t)s);
}
int segk_pack_convt_chunk(void) { return segk_pack_convt_chunk_impl(); }
int segk_pack_multi(const void* table, int n, int total_blocks, int dtype, segk_stream_t s) {
  return segk_pack_multi_impl(table, n, total_blocks, dtype, (hipStream_t)s);
}
int segk_pack_convt_weight(const float* w, void* dst, int Cin, int Cout, int Cinp, int Coutp, int mode, int dtype,
                           segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "pack_convt_weight: bad dtype %d", dtype);
  return segk_pack_convt_weight_impl(w, dst, Cin, Cout, Cinp, Coutp, mode, dtype, (hipStream_t)s);
}

int segk_conv_tiles(int B, int H, int W, int Cin, int Cout, int dtype) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (long long)B * H * W > 0x1fffffffLL) return 0;
  if (segk_conv_use_rs(Cin, Cout, dtype, W)) return segk_conv_rs_rows(B, H, W, Cout);   // one row per wave slab
  const int pk = segk_conv_use_pipe(Cin, Cout, dtype);
  const int bm = segk_conv_use_ws(Cin, Cout, dtype) ? 256 : pk ? 32768 / pk : segk_conv_bm(0, Cout);
  const int twl = segk_conv_twl(bm, W);
  return B * cdiv(W, 1 << twl) * cdiv(H, bm >> twl);
}

int segk_conv3x3(const void* srcA, const void* srcB, const void* wpacked, const float* bias, const float* scale,
                 const float* shift, void* out, void* out2, float* stats, int B, int H, int W, int CA, int CB, int CO1,
                 int CO2, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "conv3x3: bad dtype %d", dtype);
  ConvArgs a{};
  a.srcA = srcA; a.srcB = srcB; a.w = wpacked; a.bias = bias; a.scale = scale; a.shift = shift;
  a.out = out; a.out2 = out2; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.CA = CA; a.CB = CB; a.Ntot = CO1 + CO2; a.CO1 = CO1; a.CO2 = CO2;
  fill_tiles(a);
  return segk_conv_igemm_launch(a, 0, dtype, (hipStream_t)s);
}

int segk_conv_writes_act_q(int Cin, int Cout, int dtype) { return segk_conv_writes_act(Cin, Cout, dtype); }

int segk_stem3x3_rows(int B, int H, int W, int Cin, int Cout, int dtype) { return segk_stem_rows(B, H, W, Cin, Cout, dtype); }
int segk_stem3x3_wgrad_slabs(int B, int H, int W, int Cin, int Cout, int dtype) {
  return segk_stem_wgrad_slabs(B, H, W, Cin, Cout, dtype);
}
int segk_stem3x3_wgrad(const float* x_nchw, const void* dz, float* slabs, int B, int H, int W, int Cin, int Cout, int dtype,
                       segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_BF16, "stem3x3_wgrad: bf16 only (dtype %d)", dtype);
  return segk_stem_wgrad_launch(x_nchw, dz, slabs, B, H, W, Cin, Cout, (hipStream_t)s);
}
int segk_stem3x3(const float* x_nchw, const float* w_oihw, void* z, void* x_nhwc, float* stats, int B, int H, int W, int Cin,
                 int Cout, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_BF16, "stem3x3: bf16 only (dtype %d)", dtype);
  return segk_stem_launch(x_nchw, w_oihw, z, x_nhwc, stats, B, H, W, Cin, Cout, (hipStream_t)s);
}

int segk_conv3x3_act(const void* srcA, const void* wpacked, const float* scale, const float* shift, void* out,
                     void* act_out, float* stats, int B, int H, int W, int CA, int CO, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "conv3x3_act: bad dtype %d", dtype);
  ConvArgs a{};
  a.srcA = srcA; a.w = wpacked; a.scale = scale; a.shift = shift; a.out = out; a.act_out = act_out; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.CA = CA; a.Ntot = CO; a.CO1 = CO;
  fill_tiles(a);
  return segk_conv_igemm_launch(a, 0, dtype, (hipStream_t)s);
}

int segk_conv1x1(const void* srcA, const void* wpacked, const float* bias, void* out, int B, int H, int W, int CA,
                 int CO, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "conv1x1: bad dtype %d", dtype);
  ConvArgs a{};
  a.srcA = srcA; a.w = wpacked; a.bias = bias; a.out = out;
  a.B = B; a.H = H; a.W = W; a.CA = CA; a.Ntot = CO; a.CO1 = CO;
  fill_tiles(a);
  return segk_conv_igemm_launch(a, 1, dtype, (hipStream_t)s);
}

int segk_linear(const void* rows, const void* wpacked, const float* bias, void* out, long M, int K, int N, int act,
                int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "linear: bad dtype %d", dtype);
  // [M][K] x [K][N] (+ bias, optional quick_gelu): the 1x1-convolution GEMM over a 16-pixel-wide strip of M/16 rows
  SEGK_REQUIRE(M > 0 && M % 16 == 0 && M / 16 < (1 << 24), "linear: M=%ld must be a positive multiple of 16", M);
  SEGK_REQUIRE(act == 0 || act == 1, "linear: bad activation %d", act);
  ConvArgs a{};
  a.srcA = rows; a.w = wpacked; a.bias = bias; a.out = out;
  a.B = 1; a.H = (int)(M / 16); a.W = 16; a.CA = K; a.Ntot = N; a.CO1 = N; a.act = act;
  fill_tiles(a);
  return segk_conv_igemm_launch(a, 1, dtype, (hipStream_t)s);
}

int segk_linear_splitk(const void* rows, const void* wpacked, const float* bias, void* out_parts, long M, int K, int N,
                       int ksplit, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "linear_splitk: bad dtype %d", dtype);
  // bf16 only: `ksplit` partial products [ksplit][M][N] (the consumer sums them: segk_add_layernorm_parts)
  SEGK_REQUIRE(dtype == SEGK_DT_BF16, "linear_splitk: bf16 only");
  SEGK_REQUIRE(M > 0 && M % 16 == 0 && K > 0 && K % 64 == 0 && N > 0 && ksplit >= 1, "linear_splitk: bad shape");
  SEGK_REQUIRE(segk_gemm_pipe_ok(M, K / 32, K / 32, N, N, 0) && (K / 64) % ksplit == 0 && K / 64 / ksplit >= 1,
               "linear_splitk: K=%d does not split %d ways into 64-element stages (or the shape is not served)", K, ksplit);
  GemmArgs g{};
  g.A = rows; g.w = (const char*)wpacked; g.bias = bias; g.out = out_parts;
  g.M = M; g.N = N; g.nchunks = K / 32; g.nchA = K / 32; g.lda = K; g.H = 1; g.W = 1; g.Cout = N;
  g.ksplit = ksplit; g.split_stride = M * (long)N;
  return segk_gemm_pipe_launch(g, 0, (hipStream_t)s);
}

int segk_convt2x2_fwd(const void* in, const void* wpacked, const float* bias4, void* out, int B, int H, int W, int Cin,
                      int Cout, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "convt2x2_fwd: bad dtype %d", dtype);
  // bias4: per-N bias of length 4*Cout (the layer bias repeated for the four taps) or NULL
  SEGK_REQUIRE(in && wpacked && out && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "convt2x2_fwd: bad arguments");
  if (segk_convt_stream_ok(B, H, W, Cin, Cout, dtype))      // short K: weights in registers, no LDS, no unit boundary
    return segk_convt_stream_launch(in, wpacked, bias4, out, B, H, W, Cin, Cout, (hipStream_t)s);
  ConvArgs a{};
  a.srcA = in; a.w = wpacked; a.bias = bias4; a.out = out;
  a.B = B; a.H = H; a.W = W; a.CA = Cin; a.Ntot = 4 * Cout; a.CO1 = Cout; a.shuffle = 1;
  fill_tiles(a);
  return segk_conv_igemm_launch(a, 1, dtype, (hipStream_t)s);
}

int segk_convt2x2_dgrad(const void* dout, const void* wpacked, void* din, int B, int H, int W, int Cin, int Cout,
                        int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "convt2x2_dgrad: bad dtype %d", dtype);
  SEGK_REQUIRE(dout && wpacked && din && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0, "convt2x2_dgrad: bad arguments");
  if (segk_convt_stream_dgrad_ok(B, H, W, Cin, Cout, dtype))
    return segk_convt_stream_dgrad_launch(dout, wpacked, din, B, H, W, Cin, Cout, (hipStream_t)s);
  ConvArgs a{};
  a.srcA = dout; a.w = wpacked; a.out = din;
  a.B = B; a.H = H; a.W = W; a.CA = Cout; a.Ntot = Cin; a.CO1 = Cin; a.unshuf = 1;
  fill_tiles(a);
  return segk_conv_igemm_launch(a, 1, dtype, (hipStream_t)s);
}

int segk_wgrad(const void* dz, const void* srcA, const void* srcB, const float* scale, const float* shift, float* slabs,
               const void* zeros, int S, int B, int H, int W, int CD, int CA, int CB, int geo, int dtype,
               segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "wgrad: bad dtype %d", dtype);
  WgradArgs a{};
  a.dz = dz; a.srcA = srcA; a.srcB = srcB; a.scale = scale; a.shift = shift; a.slabs = slabs; a.zeros = zeros;
  a.B = B; a.H = H; a.W = W; a.CD = CD; a.CA = CA; a.CB = CB; a.S = S;
  return segk_wgrad_launch(a, geo, dtype, (hipStream_t)s);
}
int segk_wgrad_reduce(const float* slabs, int S, float* grad, int N, int CA, int CB, int Np, int CAp, int CBp, int taps,
                      segk_stream_t s) {
  return segk_wgrad_reduce_impl(slabs, S, grad, N, CA, CB, Np, CAp, CBp, taps, (hipStream_t)s);
}
int segk_wgrad_reduce_multi(const segk_reduce_job* jobs, int n, segk_stream_t s) {
  return segk_wgrad_reduce_multi_impl(jobs, n, (hipStream_t)s);
}

int segk_bn_finalize(const float* stats, int tiles, int Cp, int C, double count, const float* conv_bias,
                     const float* gamma, const float* beta, float* running_mean, float* running_var, float momentum,
                     float eps, int training, float* scale, float* shift, float* mean, float* rstd, segk_stream_t s) {
  return segk_bn_finalize_impl(stats, tiles, Cp, C, count, conv_bias, gamma, beta, running_mean, running_var, momentum,
                               eps, training, scale, shift, mean, rstd, (hipStream_t)s);
}
int segk_bn_relu_apply(const void* z, void* y, const float* scale, const float* shift, long P, int Cp, int dtype,
                       segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "bn_relu_apply: bad dtype %d", dtype);
  return segk_bn_relu_apply_impl(z, y, scale, shift, P, Cp, dtype, (hipStream_t)s);
}
int segk_bn_relu_apply_pool(const void* z, void* y, void* pooled, const float* scale, const float* shift, int B, int H, int W,
                            int Cp, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "bn_relu_apply_pool: bad dtype %d", dtype);
  return segk_bn_relu_apply_pool_impl(z, y, pooled, scale, shift, B, H, W, Cp, dtype, (hipStream_t)s);
}
int segk_bn_relu_bwd(const void* dy, const void* z, void* dz, const float* scale, const float* shift, const float* mean,
                     const float* rstd, long P, int Cp, int C, float* part, float* dgamma, float* dbeta, float* coef,
                     int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "bn_relu_bwd: bad dtype %d", dtype);
  return segk_bn_bwd_impl(dy, z, dz, scale, shift, mean, rstd, P, Cp, C, part, dgamma, dbeta, coef, dtype,
                          (hipStream_t)s);
}
int segk_channel_sum(const void* x, long P, int Cp, int C, float* part, float* out, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "channel_sum: bad dtype %d", dtype);
  return segk_channel_sum_impl(x, P, Cp, C, part, out, dtype, (hipStream_t)s);
}
int segk_maxpool2x2_fwd(const void* x, void* y, int B, int H, int W, int Cp, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "maxpool2x2_fwd: bad dtype %d", dtype);
  return segk_maxpool_fwd_impl(x, y, B, H, W, Cp, dtype, (hipStream_t)s);
}
int segk_maxpool2x2_bwd(const void* x, const void* dy, void* dx, int B, int H, int W, int Cp, int accumulate, int dtype,
                        segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "maxpool2x2_bwd: bad dtype %d", dtype);
  return segk_maxpool_bwd_impl(x, dy, dx, B, H, W, Cp, accumulate, dtype, (hipStream_t)s);
}
int segk_maxpool2x2_bwd_bnstat(const void* x, const void* dy, void* dx, int B, int H, int W, int Cp, int accumulate,
                               const float* scale, const float* shift, const float* mean, const float* rstd, float* part,
                               const void* z, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "maxpool2x2_bwd_bnstat: bad dtype %d", dtype);
  return segk_maxpool_bwd_bnstat_impl(x, dy, dx, B, H, W, Cp, accumulate, scale, shift, mean, rstd, part, z, dtype,
                                      (hipStream_t)s);
}
int segk_bn_relu_bwd_from_part(const void* dy, const void* z, void* dz, const float* scale, const float* shift,
                               const float* mean, const float* rstd, long P, int Cp, int C, const float* part, int nb,
                               float* dgamma, float* dbeta, float* coef, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "bn_relu_bwd_from_part: bad dtype %d", dtype);
  return segk_bn_bwd_from_part_impl(dy, z, dz, scale, shift, mean, rstd, P, Cp, C, part, nb, dgamma, dbeta, coef, dtype,
                                    (hipStream_t)s);
}
int segk_head_fwd(const void* y, const float* w, const float* bias, float* logits, int B, int H, int W, int Cp, int C,
                  int ncls, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "head_fwd: bad dtype %d", dtype);
  return segk_head_fwd_impl(y, w, bias, logits, B, H, W, Cp, C, ncls, nullptr, nullptr, dtype, (hipStream_t)s);
}
int segk_head_fwd_bn(const void* z, const float* scale, const float* shift, const float* w, const float* bias, float* logits,
                     int B, int H, int W, int Cp, int C, int ncls, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(scale && shift, "head_fwd_bn: null scale/shift");
  return segk_head_fwd_impl(z, w, bias, logits, B, H, W, Cp, C, ncls, scale, shift, dtype, (hipStream_t)s);
}
int segk_head_bwd(const float* dlogits, const void* y, const float* w, void* dy, float* part, float* dw, float* db,
                  int B, int H, int W, int Cp, int C, int ncls, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "head_bwd: bad dtype %d", dtype);
  return segk_head_bwd_impl(dlogits, y, w, dy, part, dw, db, B, H, W, Cp, C, ncls, nullptr, nullptr, nullptr, nullptr, nullptr,
                            0, dtype, (hipStream_t)s);
}
int segk_head_bwd_blocks(long P) { return segk_head_blocks_q(P); }
int segk_head_bwd_bnstat(const float* dlogits, const void* y, const float* w, void* dy, float* part, float* dw, float* db,
                         int B, int H, int W, int Cp, int C, int ncls, const float* scale, const float* shift,
                         const float* mean, const float* rstd, float* bnpart, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(dtype == SEGK_DT_F32 || dtype == SEGK_DT_BF16, "head_bwd_bnstat: bad dtype %d", dtype);
  SEGK_REQUIRE(bnpart, "head_bwd_bnstat: null partials");
  return segk_head_bwd_impl(dlogits, y, w, dy, part, dw, db, B, H, W, Cp, C, ncls, scale, shift, mean, rstd, bnpart, 0, dtype,
                            (hipStream_t)s);
}
int segk_head_bwd_bn(const float* dlogits, const void* z, const float* w, void* dy, float* part, float* dw, float* db,
                     int B, int H, int W, int Cp, int C, int ncls, const float* scale, const float* shift,
                     const float* mean, const float* rstd, float* bnpart, int dtype, segk_stream_t s) {
  SEGK_REQUIRE(scale && shift && mean && rstd, "head_bwd_bn: null BatchNorm vectors");
  return segk_head_bwd_impl(dlogits, z, w, dy, part, dw, db, B, H, W, Cp, C, ncls, scale, shift, mean, rstd, bnpart, 1, dtype,
                            (hipStream_t)s);
}
int segk_loss_fwd(const float* logits, const int64_t* labels, const float* cw, int N, int C, long HW, int ignore_index,
                  float smooth, float dice_weight, float ce_weight, float* part, float* state, float* loss_out,
                  segk_stream_t s) {
  return segk_loss_fwd_impl(logits, (const long long*)labels, cw, N, C, HW, ignore_index, smooth, dice_weight, ce_weight,
                            part, state, loss_out, 0, 0, 0.f, (hipStream_t)s);
}
int segk_prob_loss_fwd(const float* probs, const int64_t* labels, const float* cw, int N, int C, long HW, int ignore_index,
                       float smooth, float dice_weight, float nll_weight, int nll_log, float eps, float* part, float* state,
                       float* loss_out, segk_stream_t s) {
  return segk_loss_fwd_impl(probs, (const long long*)labels, cw, N, C, HW, ignore_index, smooth, dice_weight, nll_weight,
                            part, state, loss_out, 1, nll_log, eps, (hipStream_t)s);
}
int segk_prob_loss_bwd(const float* probs, const int64_t* labels, const float* cw, const float* state, const float* gout,
                       int N, int C, long HW, int ignore_index, float dice_weight, float nll_weight, int nll_log, float eps,
                       float* dprobs, segk_stream_t s) {
  return segk_loss_bwd_impl(probs, (const long long*)labels, cw, state, gout, N, C, HW, ignore_index, dice_weight,
                            nll_weight, dprobs, 1, nll_log, eps, (hipStream_t)s);
}
int segk_prompt_mix_fwd(const float* clip_logits, const float* mask_logit, float* final_probs, int N, long HW,
                        segk_stream_t s) {
  return segk_prompt_mix_impl(clip_logits, mask_logit, nullptr, final_probs, N, HW, (hipStream_t)s);
}
int segk_prompt_mix_bwd(const float* clip_logits, const float* mask_logit, const float* dfinal, float* dmask_logit, int N,
                        long HW, segk_stream_t s) {
  SEGK_REQUIRE(dfinal, "prompt_mix_bwd: null gradient");
  return segk_prompt_mix_impl(clip_logits, mask_logit, dfinal, dmask_logit, N, HW, (hipStream_t)s);
}
int segk_loss_bwd(const float* logits, const int64_t* labels, const float* cw, const float* state, const float* gout,
                  int N, int C, long HW, int ignore_index, float dice_weight, float ce_weight, float* dlogits,
                  segk_stream_t s) {
  return segk_loss_bwd_impl(logits, (const long long*)labels, cw, state, gout, N, C, HW, ignore_index, dice_weight,
                            ce_weight, dlogits, 0, 0, 0.f, (hipStream_t)s);
}
int segk_confusion(const float* logits, const int64_t* labels, int N, int C, long HW, uint64_t* M, segk_stream_t s) {
  return segk_confusion_impl(logits, (const long long*)labels, N, C, HW, (unsigned long long*)M, (hipStream_t)s);
}

}  // extern "C"
