// Internal (non-ABI) declarations shared between the kernel translation units and api.hip.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct ConvArgs {
  const void* srcA;     // NHWC [B,H,W,CA]  (un-shuffle gather: [B,2H,2W,CA])
  const void* srcB;     // optional second source [B,H,W,CB] (channel concat without a concat)
  const void* w;        // packed weights [K/CH][taps][Ntot][CH]
  const float* bias;    // optional per-N bias
  const float* scale;   // optional BN+ReLU prologue (per input channel)
  const float* shift;
  void* out;            // [B,H,W,CO1]  (pixel-shuffle: [B,2H,2W,CO1])
  void* out2;           // optional second destination [B,H,W,CO2]
  float* stats;         // optional [tiles][Ntot][2] BN partial sums
  void* act_out;        // optional [B,H,W,CA]: the prologue's BN+ReLU output of srcA, written once (producer/consumer and
                        // weight-stationary kernels only: segk_conv_writes_act)
  int B, H, W;
  int CA, CB;
  int Ntot, CO1, CO2;
  int twl, tiles_x, tiles_y;
  int unshuf, shuffle;
  int persistent;       // set by the launcher
  int act;              // 1x1 geometry only: 1 = quick_gelu on (acc + bias) (CLIP MLP fc1)
};
int segk_conv_igemm_launch(const ConvArgs& a, int geo, int dtype, hipStream_t st);

// per-device launch state (hipFuncSetAttribute flags, CU counts) is indexed by the current device
#define SEGK_MAX_DEVICES 64
int segk_device_index();                  // hipGetDevice(), clamped to [0, SEGK_MAX_DEVICES)
int segk_num_cus();                       // multiprocessor count of the current device (256 when no device is visible)

// register-stationary kernel for the narrow high-resolution bf16 layers (conv_rs.hip)
int segk_conv_use_rs(int cin_p, int n_p, int dtype, int W);
int segk_conv_rs_rows(int B, int H, int W, int n_p);          // rows of BatchNorm partials it writes
void segk_conv_rs_grid(int B, int H, int W, int NT, int* gw_out, int* GW_out);
int segk_conv_rs_launch(const ConvArgs& a, hipStream_t st);
// the U-Net stem on the NCHW fp32 input (stem.hip)
int segk_stem_rows(int B, int H, int W, int Cin, int Cout, int dtype);
int segk_stem_launch(const float* x, const float* w, void* z, void* xn, float* stats, int B, int H, int W, int Cin, int Cout,
                     hipStream_t st);
int segk_stem_wgrad_slabs(int B, int H, int W, int Cin, int Cout, int dtype);
int segk_stem_wgrad_launch(const float* x, const void* dz, float* slabs, int B, int H, int W, int Cin, int Cout,
                           hipStream_t st);
// register-stationary streaming kernel for the short-K ConvTranspose forward (convt_stream.hip)
int segk_convt_stream_ok(int B, int H, int W, int Cin, int Cout, int dtype);
int segk_convt_stream_launch(const void* x, const void* wp, const float* bias4, void* out, int B, int H, int W, int Cin,
                             int Cout, hipStream_t st);
int segk_convt_stream_dgrad_ok(int B, int H, int W, int Cin, int Cout, int dtype);
int segk_convt_stream_dgrad_launch(const void* dout, const void* wd, void* din, int B, int H, int W, int Cin, int Cout,
                                   hipStream_t st);
int segk_conv_use_ws(int cin_p, int n_p, int dtype);   // weight-stationary variant applies
int segk_conv_use_pipe(int cin_p, int n_p, int dtype); // producer/consumer variant: its channel tile (128 | 64) or 0
int segk_conv_writes_act(int cin_p, int n_p, int dtype); // the layer's kernel can emit ConvArgs::act_out
int segk_conv_bm(int geo, int unit);      // pixels per tile for a layer with N = unit output channels
int segk_conv_twl(int bm, int W);         // log2 tile width

// one tensor of segk_pack_multi's device table (64 bytes; the host builds it as 8 int64 words)
//   kind 0: Conv2d 3x3 weight OIHW -> forward + data-gradient layouts   (blocks: (CAp+CBp)/32 * Coutp/32)
//   kind 1: ConvTranspose2d(k=2,s=2) weight IOHW -> GEMM + un-shuffle layouts; CA = Cin, CAp = Cinp  (blocks: ceil(Cinp*4*Coutp / 2048))
//   kind 2: bias [Cout] -> fp32 [reps][Coutp] (one block), reps = CA (0: 4, the ConvTranspose2d bias4 operand; 1: a conv bias); dst_dgrad unused
//   kind 3: Conv2d 1x1 weight [Cout][CA] -> forward + data-gradient (may be null) layouts  (blocks: ceil(CAp*Coutp / 2048))
struct SegkPackEntry {
  const float* w;       // fp32 parameter
  void* dst_fwd;        // forward layout
  void* dst_dgrad;      // data-gradient layout (or null)
  int Cout, CA, CB, Coutp, CAp, CBp;
  int block0;           // first block of this tensor in the launch (entries sorted by it)
  int kind;
  int pad_[2];
};
static_assert(sizeof(SegkPackEntry) == 64, "SegkPackEntry is 64 bytes");

struct WgradArgs {
  const void* dz;       // NHWC [B,H,W,CD]        (un-shifted operand; rows of dW)
  const void* srcA;     // NHWC [B,H,W,CA]        (tap-shifted operand; columns of dW)
  const void* srcB;     // optional second source [B,H,W,CB]
  const float* scale;   // optional BN+ReLU prologue on the shifted operand
  const float* shift;
  float* slabs;         // [S][CD][taps][CA+CB] fp32 partial weight gradients
  const void* zeros;    // >= 64 bytes of zeros in device memory (source of out-of-image pixels for the DMA path)
  int B, H, W;          // grid of the un-shifted operand
  int CD, CA, CB;
  int S;                // split-K factor over spatial tiles
};
int segk_wgrad_launch(const WgradArgs& a, int geo, int dtype, hipStream_t st);
int segk_wgrad_wc(int CD, int CA, int CB, int geo, int dtype);

// producer/consumer bf16 GEMM of the 1x1 geometry (gemm.hip)
struct GemmArgs {
  const void* A;        // rows [M][lda] bf16 (un-shuffle gather: [B,2H,2W,lda])
  const char* w;        // packed weights [nchunks][N][32] bf16
  const float* bias;    // optional per-N bias
  void* out;            // [M][N] bf16 (pixel-shuffle store: [B,2H,2W,Cout])
  long M;               // rows (B*H*W)
  int N;                // multiple of 128
  int nchunks;          // K / 32 (even)
  int nchA;             // un-shuffle gather: chunks per tap (K = 4 taps x nchA chunks)
  int lda;              // elements per A row
  int H, W;             // input grid of the ConvTranspose modes
  int Cout;             // pixel-shuffle store: channels per tap (N = 4*Cout)
  int act;              // 1 = quick_gelu on (acc + bias)
  int ksplit;           // > 1: split-K, partial outputs [ksplit][M][N] (split_stride elements apart), bias on split 0
  long split_stride;
};
int segk_gemm_pipe_ok(long M, int nchunks, int nchA, int N, int cout_shuffle, int mode);   // mode 0 plain, 1 shuffle, 2 un-shuffle
int segk_gemm_pipe_launch(const GemmArgs& g, int mode, hipStream_t st);
