// Training step, last four layers of decoder_400 (ConvT 64->32 -> ConvT 32->16 -> ConvT 16->8, all 2x2 stride 2 swish,
// -> Conv 3x3 SAME 8->1 linear; sr-ae-conv.ipynb:c283-286) as TWO launches instead of ~25 (train_tail.hip):
//   forward : tail32<TRAIN> (kernels_tail32.hip) streams the 50x50x64 activation to the loss gradient dpred and the
//             squared-error sum; nothing between them touches HBM;
//   backward: tail_bwd32 recomputes the three transposed convolutions per 16-pixel tile of the 50x50 level in registers
//             and runs every data and weight gradient of the four layers on them: per sample it reads 2 x 0.64 MB of
//             activations + 0.64 MB of dpred and writes 0.64 MB (dZ of ConvT#1) -- the layer-by-layer path moved ~60 MB.
// Internal to the library (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <vector>

#include "kernels.h"
#include "model.h"

namespace srcfd {

// Flat-parameter layout of the four layers (Keras trainable_weights order: kernel, bias per layer), relative to the first
// parameter of ConvT#2.  One gradient slab of tail_bwd32 has exactly this layout.
constexpr int TT_O_W1 = 0, TT_O_B1 = 8192, TT_O_W2 = 8224, TT_O_B2 = 10272, TT_O_W3 = 10288, TT_O_B3 = 10800, TT_O_WC = 10808, TT_O_BC = 10880;
constexpr int TT_PARAMS = 10881;

// Operand packs, all gathered from the flat parameters by train.hip's gather kernel (index map + per-slot scale):
//   t32_* : the operands of tail32 (engine.hip, plan_tail32: log2(e) folded into w1 / biases, 1 / log2(e) into wc)
//   wf    : unscaled forward A fragments  w1f[4][2][16][64] | w2f[4][8][64] | w3f[2][4][64]      (same lane maps as tail32)
//   wb    : data-gradient A fragments     a1b[4][4][2][4][64] | a2b[4][2][4][64] | a3b[2][4][64] (train_tail.hip)
//   wt    : the output conv as a banded 16 x 12 matrix, wt[3][64]
//   bias  : b1[32] | b2[16] | b3[8] | wc[72] | bc[1]  (unscaled)
constexpr int TT_WF = 4 * 2 * 16 * 64 + 4 * 8 * 64 + 2 * 4 * 64;   // 10752 floats
constexpr int TT_WB = 4 * 4 * 8 * 64 + 4 * 2 * 4 * 64 + 2 * 4 * 64; // 10752
constexpr int TT_WT = 3 * 64;                                       // Toeplitz fragments of the output conv (train_tail.hip)
constexpr int TT_BIAS = 32 + 16 + 8 + 72 + 1;                       // 129 (padded to 192 in the pack)

struct TrainTailPlan {
  bool ok = false;
  int first_layer = 0;        // compute-layer ordinal of ConvT#2 (the tail is layers first_layer .. first_layer + 3)
  int H = 0, W = 0;           // spatial size of its input (50 x 50)
  size_t param_off = 0;       // flat index of ConvT#2's first kernel element
  // pack (floats, relative to the start of the tail's pack region): maps hold flat index + 1 (0: padding)
  std::vector<int> map;
  std::vector<float> scale;
  size_t t32_w1 = 0, t32_b1 = 0, t32_w2 = 0, t32_b2 = 0, t32_w3 = 0, t32_b3 = 0, t32_wc = 0, wf = 0, wb = 0, wt = 0, bias = 0;
};
// kernel_off / bias_off: flat offsets of the four layers' kernels and biases (train.hip, LayerInfo)
void train_tail_plan(const ModelDesc& desc, const int* desc_index, const size_t* kernel_off, const size_t* bias_off, int n_compute_layers,
                     TrainTailPlan& plan);

struct TailBwdParams {
  const float* y1;      // (n, H, W, 64): swish output of ConvT#1
  const float* z1;      // its pre-activation
  const float* dpred;   // (n, 8H, 8W): loss gradient
  float* dz1;           // out: (n, H, W, 64) gradient w.r.t. ConvT#1's pre-activation
  float* slabs;         // out: [tail_bwd32_blocks()][TT_PARAMS] weight-gradient partial sums, one slab per workgroup
  const float* wf; const float* wb; const float* wt; const float* bias;
  int n, H, W;
  unsigned magic_hw = 0, magic_w = 0;   // filled by launch_tail_bwd32: division by H W and by W as a multiply
};
int tail_bwd32_blocks(int n, int H, int W, int num_cus);
hipError_t launch_tail_bwd32(const TailBwdParams& p, int num_cus, hipStream_t s);

}  // namespace srcfd
