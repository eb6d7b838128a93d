// Kernel-side descriptors and launchers shared by engine.hip and the kernel TUs.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srcfd.h"

namespace srcfd {

// Implicit-GEMM descriptor; see the header comment of kernels_fp32.hip.
struct GemmDesc {
  int M, N, K, Npad;
  int MH, MW;          // per-image row grid
  int TY, TX, CI;      // K = TY*TX*CI
  int IH, IW;          // input spatial dims (channels = CI)
  int ay, by, cy;      // iy = my*ay + ty*by + cy
  int ax, bx, cx;
  int CO, nphx;        // n -> (phase = n / CO, co = n % CO); phase -> (py = ph / nphx, px = ph % nphx)
  int OH, OW, OC;      // output tensor dims
  int os, oy0, ox0;    // oy = my*os + oy0 + py
  int act;
};

// Training epilogues of the GEMM (train.hip): the element-wise pass that would follow the launch, done on the accumulators.
//   mode 1 (forward):        Y = v (pre-activation, kept for the backward pass),  y2 = swish(v)
//   mode 2 (data gradient):  Y = v * swish'(zaux)   (zaux: the pre-activation stored at the same offsets)
// Same expressions as the stand-alone swish_fwd_f32 / swish_bwd_f32 kernels, so fused and unfused results are identical.
struct EpiAux {
  float* y2 = nullptr;
  const float* zaux = nullptr;
  int mode = 0;
};
bool gemm_supports_epi_aux(const GemmDesc& d);

// The whole encoder_10 in one launch, f32 (kernels_enc32.hip)
constexpr int ENC32_G = 3;   // samples per workgroup
struct Enc32Params {
  const float* x;          // (n,10,10,1) f32
  const float* affine;     // (n,2) mean,std or null
  int n;
  const float* w1; const float* b1;     // conv2d: B[9][64], bias[64]
  const float* w2f; const float* b2;    // conv2d_1: A fragments [channel tile 8][tap*4+q 36][64 lanes][4] (engine.hip, plan_enc32), bias[128]
  const float* wd; const float* bd;     // dense: B[3200][128], bias[128]
  const float* wl; const float* bl;     // latent_vector: B[128][nl_pad], bias[nl]
  float* z;                             // (n, nl)
  int nl, nl_pad;
  int act1, act2, act3, act4;
};
hipError_t launch_enc32(const Enc32Params& p, hipStream_t s);

// Dense with K <= 52 and N a large multiple of 144 (decoder dense_1), f32: one workgroup per 144 output features (kernels_gemm32.hip)
bool dense_skinny32_qualifies(const GemmDesc& d);
hipError_t launch_dense_skinny32(const GemmDesc& d, const float* X, const float* B, const float* bias, float* Y, hipStream_t s);

hipError_t launch_gemm_naive(const GemmDesc& d, const float* X, const float* B, const float* bias, float* Y, hipStream_t s);
// ws: optional split-K scratch (gemm_splitk_ws_floats(d) floats); without it the launch never splits.
hipError_t launch_gemm_mfma(const GemmDesc& d, const float* X, const float* B, const float* bias, float* Y, hipStream_t s,
                            float* ws = nullptr, size_t ws_floats = 0, bool batch_invariant = true, EpiAux aux = EpiAux());
// batch_invariant (inference): the cut depends on the layer only, never on the batch size; false (training): on the tile count too.
int gemm_splitk_splits(const GemmDesc& d, int* kchunk_out, bool batch_invariant = true);
size_t gemm_splitk_ws_floats(const GemmDesc& d, bool batch_invariant = true);
// The GEMMs of one layer (the output phases of a transposed convolution with kernel != stride) as one launch + one finish;
// ws must hold gemm_group_ws_floats() floats, otherwise (or when the GEMMs do not qualify) they are launched one by one.
hipError_t launch_gemm_mfma_group(const GemmDesc* ds, int count, const float* X, const float* const* Bs, const float* const* biases, float* Y,
                                  hipStream_t s, float* ws, size_t ws_floats, bool batch_invariant = true, EpiAux aux = EpiAux());
size_t gemm_group_ws_floats(const GemmDesc* ds, int count, bool batch_invariant = true);
// Large launches of the wide decoder layers (ConvT#0's phases as one launch, ConvT#1): kernels_gemm32.hip.  Same arithmetic
// as launch_gemm_mfma / launch_gemm_mfma_group (bit-identical results), K splits summed inside the workgroup.
struct Gemm32Group {
  GemmDesc d[4];
  const float* B[4];
  const float* bias[4];
  int kchunk[4];
  int count;
};
bool gemm32_big_qualifies(const GemmDesc* ds, int count);
hipError_t launch_gemm32_big(const GemmDesc* ds, int count, const float* X, const float* const* Bs, const float* const* biases, float* Y,
                             hipStream_t s);
// Two consecutive kernel == stride == 2 transposed convolutions (CI 32 -> 16 -> 8 channels: ConvT#3 -> ConvT#4 of
// decoder_400) as one kernel: every input pixel expands to its own 4 x 4 output block, no halo, so the 16-channel
// intermediate (0.65 GB written + read per 256 samples, these two layers are HBM-bound) never leaves the registers.
struct PairDesc {
  int n, H, W;         // input (n, H, W, 32); output (n, 4H, 4W, 8)
  int act_a, act_b;
};
// wa [2 tiles][16 k-steps][64 lanes], ba [16], wb [8 k-steps][64 lanes], bb [8]: see pack_convt_pair() in engine.hip
hipError_t launch_convt_pair_f32(const PairDesc& d, const float* X, const float* wa, const float* ba, const float* wb, const float* bb,
                                 float* Y, hipStream_t s);
// The same with one more layer in front: 64 -> 32 -> 16 -> 8 channels (ConvT#2 -> ConvT#3 -> ConvT#4), input (n, H, W, 64),
// output (n, 8H, 8W, 8).  w1 [4 tiles][32 k-steps][64 lanes] (staged in LDS), b1 [32]; w2 [2][16][64], b2 [16]; w3 [8][64], b3 [8].
struct TripleDesc {
  int n, H, W;
  int act1, act2, act3;
};
hipError_t launch_convt_triple_f32(const TripleDesc& d, const float* X, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* w3, const float* b3, float* Y, hipStream_t s);
// The last four layers of decoder_400 in f32 as ONE streaming kernel (kernels_tail32.hip): ConvT 64->32 -> ConvT 32->16 ->
// ConvT 16->8 (all 2x2 stride 2) -> Conv 3x3 SAME 8->1 + de-standardise + NaN guard + output cast; the 8-channel
// full-resolution activation lives in an LDS ring only.  Input (n, H, W, 64) f32 with W <= 50; output (n, 8H, 8W).
// The three transposed convolutions are swish, the output conv linear; w1f, b1, b2, b3 carry a factor log2(e), wc 1/log2(e)
// (kernels_tail32.hip, swish_l2e).
struct Tail32Params {
  const float* in;
  void* out;
  int n, H, W;
  const float* w1f;   // [tap1 4][m-tile 2][k-step 16][64 lanes]: lane (m, kg) = W1[tap1][co 16t+m][ci 16kg+s]
  const float* b1;    // [32]
  const float* w2f;   // [tap2 4][k-step 8][64 lanes]: lane (m, kg), step 4t+i = W2[tap2][co m][ci 16t+4kg+i]
  const float* b2;    // [16]
  const float* w3f;   // [m-tile 2][k-step 4][64 lanes]: lane (m, kg) = W3[tap3 2u+(m>>3)][co m&7][ci 4kg+i]
  const float* b3;    // [8]
  const float* wc;    // [72] output conv (ty, tx, ci) + [1] bias
  const float* aff_out;   // (n, 2) mean, std or null
  int nan_guard;
  unsigned long long* nonfinite;
  int out_dtype;
  int seg;            // segments per sample, 1 .. 2H (need not divide 2H: segment s takes strips [s 2H / seg, (s + 1) 2H / seg))
  // training forward (train.hip): out = two_scale * (image - target) in f32, sum of squared errors of workgroup b in
  // sse_partial[b] (tail32_blocks() of them); aff_out / nan_guard are ignored.  null: inference.
  const float* target = nullptr;
  float two_scale = 0.f;
  double* sse_partial = nullptr;
  // SRCFD_PREC_FP32X3: the first layer's weights (x log2 e, like w1f) split exactly into three bf16 planes, as
  // v_mfma_f32_16x16x32_bf16 A fragments: [tap row ty1 2][tap column tx1 2][m-tile t 2][k-step c 2][plane 3][64 lanes] x 16 B,
  // lane (m, kg) element j = plane(W1[2 ty1 + tx1][co 16 t + m][ci 32 c + 8 kg + j]).  null: the f32 MFMAs of w1f.
  const uint16_t* w1x = nullptr;
  // ... and the second layer's (unscaled, like w2f): [tap2 4][plane 3][64 lanes] x 16 B, lane (m, kg) element j =
  // plane(W2[tap2][co m][ci j < 4 ? 4 kg + j : 16 + 4 kg + j - 4]) -- the k order of the first layer's accumulators.  Both or neither.
  const uint16_t* w2x = nullptr;
  // diagnostic (SRCFD_TAIL32_ABLATE, -DSRCFD_DIAG builds only): 1 no swish, 2 no output conv arithmetic, 4 no third-layer MFMAs,
  // 8 no first / second layer MFMAs, 16 no operand splits (X3), 32 no ring stores, 64 no window reads
  int ablate = 0;
};
hipError_t launch_tail32(const Tail32Params& p, int num_cus, hipStream_t s);
int tail32_segments(int n, int H, int num_cus);
int tail32_blocks(int n, int seg, int num_cus);   // workgroups launch_tail32 starts

// f32-grade GEMM on bf16 MFMAs, operands split exactly into three bf16 terms (kernels_x3.hip; SRCFD_PREC_FP32X3).
// Wt: [plane 3][N][Kpad] bf16 from gemm_x3_split_weights(); d.M set (the 32-bit input offsets bound the batch).
bool gemm_x3_qualifies(const GemmDesc& d);
int gemm_x3_kpad(const GemmDesc& d);
void gemm_x3_split_weights(const GemmDesc& d, const float* B, uint16_t* out);
hipError_t launch_gemm_x3(const GemmDesc& d, const float* X, const uint16_t* Wt, const float* bias, float* Y, hipStream_t s, int num_cus = 256);
// the GEMMs of one layer (output phases of a strided transposed convolution), longest first: one launch where the kernel allows it
hipError_t launch_gemm_x3_group(const GemmDesc* ds, int count, const float* X, const uint16_t* const* Wts, const float* const* biases, float* Y,
                                hipStream_t s, int num_cus = 256);

// Last layer + de-standardise + NaN guard + output cast in one kernel, for the layers gemm_fuses_finalize() accepts.
bool gemm_fuses_finalize(const GemmDesc& d);
hipError_t launch_gemm_finalize(const GemmDesc& d, const float* X, const float* B, const float* bias, void* out, int out_dtype,
                                const float* affine, int nan_guard, unsigned long long* nonfinite, hipStream_t s);
hipError_t launch_standardize(const float* x, float* y, const float* affine, int per_sample, int64_t total, hipStream_t s);
hipError_t launch_finalize(const float* y, void* out, int out_dtype, const float* affine, int per_sample, int64_t total,
                           int nan_guard, unsigned long long* nonfinite, hipStream_t s);

}  // namespace srcfd
