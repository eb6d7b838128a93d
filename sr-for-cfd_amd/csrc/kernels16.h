// Declarations shared by kernels_bf16.hip (device) and fused_bf16.hip (host).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <utility>
#include <vector>

#include "kernels.h"

namespace srcfd {

// Constant blob the tail kernel copies into LDS (byte offsets).  Fragment
// arrays hold one 16-byte MFMA operand per lane: [fragment][64 lanes].
constexpr int TC_OFF_WC = 0;                      // output-conv Toeplitz B operands, 10 k-steps (16x16x32)
constexpr int TC_OFF_W3 = TC_OFF_WC + 10 * 1024;  // ConvT#3 A operands [m-tile 2][k-step 2]
constexpr int TC_OFF_W4 = TC_OFF_W3 + 4 * 1024;   // ConvT#4 A operand (k permuted to accumulator order)
constexpr int TC_OFF_B2 = TC_OFF_W4 + 1024;       // bias as 32x32 accumulator init, [lane half 2][16] f32
constexpr int TC_OFF_B3 = TC_OFF_B2 + 128;
constexpr int TC_OFF_B4 = TC_OFF_B3 + 128;
constexpr int TC_OFF_BC = TC_OFF_B4 + 128;        // output-conv bias (f32) + padding
constexpr int TAIL_CONST_BYTES = TC_OFF_BC + 16;

// Raises a kernel's dynamic-LDS limit once per (kernel, device): the attribute is per device, and one process may
// drive several GPUs through separate handles.
inline hipError_t lds_attr_once(const void* fn, int bytes) {
  static thread_local std::vector<std::pair<const void*, int>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  for (auto& d : done) if (d.first == fn && d.second == dev) return hipSuccess;
  e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.emplace_back(fn, dev);
  return e;
}

struct TailParams {
  const uint16_t* in;      // (n,50,50,64) activations of ConvT#1, scaled by log2e
  void* out;               // (n,400,400) of out_dtype
  int n;
  const void* consts;      // TAIL_CONST_BYTES, device
  const void* w2frags;     // ConvT#2 A operands [m-tile 4][k-step 4][64 lanes] x 16 B, device
  const float* aff_out;    // (n,2) mean,std or null
  int nan_guard;
  unsigned long long* nonfinite;
  int out_dtype;
  int ablate;              // diagnostic (SRCFD_TAIL_ABLATE): 1 no swish, 2 no D, 4 no A, 8 no BC, 16 BC before D, 32 D before BC, 64 skip the D items that hold only tiles 48-49 of a row pair, 128 no priority raise
  int seg;                 // segments per sample (1, 2, 5, 10 or 25): small batches spread one sample over several workgroups
  unsigned long long* prof;  // diagnostic (SRCFD_TAIL_PROF): per-wave cycle totals [block 0][16 waves][D, BC, A, barrier, total], else null
};

// ConvT#0 -> ConvT#1 fused kernel (kernels_mid16.hip)
struct MidParams {
  const uint16_t* in;      // (n,12,12,256) dense_1 output, scaled by log2e
  uint16_t* out;           // (n,50,50,64) ConvT#1 output, scaled by log2e
  int n;
  const uint16_t* w0[4];   // per output phase (py*2+px): ConvT#0 weights [128 ch][kpad], k = tap*256 + ci
  int kpad[4];
  const uint16_t* w0t[4];  // the same weights as the LDS images of the kernel's stages: [stage c * taps + t][1024 pieces of 16 B] (fused_bf16.hip)
  const float* b0f;        // ConvT#0 bias as accumulator init [m-tile 4][lane half 2][16]
  const void* w1f;         // ConvT#1 A operands [m-tile 8][k-step 8][64 lanes] x 16 B, k in accumulator order
  const float* b1f;        // ConvT#1 bias as accumulator init [channel half 2][lane half 2][16]
  int nblk[4];             // workgroups per output phase (filled by launch_mid16)
  int order;               // 0: phase by phase; 1: phases 0 / 3 and 1 / 2 alternating (SRCFD_MID_ORDER, A/B)
  int ablate;              // diagnostic (SRCFD_MID_ABLATE): 1 no main-loop MFMA, 2 no ConvT#1 stage, 4 no operand loads, 8 no global stores, 16 no ConvT#1 swish
  unsigned long long* prof;  // diagnostic (SRCFD_MID_PROF, DIAG builds): per phase [4][6]: workgroups, cycles entry -> tables, -> first stage ready, main loop, ConvT#0 swish, ConvT#1 stage; else null
};
hipError_t launch_mid16(bool f16, const MidParams& p, int waves /*4, 8 or 16 per workgroup*/, hipStream_t s);

// The whole encoder in one launch (kernels_enc16.hip): conv2d -> conv2d_1 -> dense -> latent_vector
constexpr int ENC_G = 5;   // samples per workgroup
struct EncParams {
  const float* x;          // (n,10,10,1) f32
  const float* affine;     // (n,2) mean,std or null
  int n;
  const float* w1;         // conv2d weights [9][64] f32, scaled
  const float* b1;         // conv2d bias [64] f32, scaled
  const void* w2f;         // conv2d_1 A operands [channel tile 4][k-step 36][64 lanes] x 16 B (32x32x16; k = tap*64 + ci)
  const float* b2f;        // conv2d_1 bias as accumulator init [channel tile 4][lane half 2][16]
  const void* wdf;         // dense A operands [feature tile 8][k-step 100][64 lanes] x 16 B (16x16x32)
  const float* bd;         // dense bias [128], scaled
  const void* wlf;         // latent_vector A operands [feature tile 4][k-step 4][64 lanes] x 16 B
  const float* bl;         // latent_vector bias [64] (50 + zero padding)
  uint16_t* z;             // (n,64) latent vectors, 16-bit
  int act_dense, act_latent;
  unsigned long long* prof;  // diagnostic (-DSRCFD_DIAG, SRCFD_ENC_PROF): cycle stamps of workgroup 7 [8 waves][8], else null
};
hipError_t launch_enc16(bool f16, const EncParams& p, hipStream_t s);
// Dense with one 64-deep k tile and N a multiple of 144 (dense_1: 64 -> 36 864): one workgroup per 144 features, all samples
bool dense1_16_qualifies(const GemmDesc& d, int Kpad);
hipError_t launch_dense1_16(bool f16, const GemmDesc& d, const uint16_t* X, const uint16_t* Wt, const float* bias, uint16_t* Y, hipStream_t s);

hipError_t launch_enc_conv1_16(bool f16, const float* x, const float* affine, const float* w, const float* b, uint16_t* y, int n, hipStream_t s);
// part/splits: optional split-K (dense layers with few rows): one f32 slab (M x Npad) per K slice in `part`,
// summed in slice order by a finish kernel (deterministic)
hipError_t launch_gemm16(bool f16, const GemmDesc& d, const uint16_t* X, const uint16_t* Wt, int Kpad, const float* bias, uint16_t* Y,
                         float* part, int splits, hipStream_t s);
hipError_t launch_tail16(bool f16, const TailParams& p, int blocks, hipStream_t s);    // shipped kernel (kernels_bf16.hip): 16 waves, stage by stage
hipError_t launch_tail16s(bool f16, const TailParams& p, int blocks, hipStream_t s);   // second implementation (kernels_tail16s.hip, SRCFD_TAIL=s): 8 waves, MFMAs inside the swish stream
int tail_lds_bytes();

}  // namespace srcfd
