// tail_bwd32: backward pass of decoder_400's last four layers in ONE kernel (gfx950, f32) -- see train_tail.h.
//   ConvT#2 64->32, ConvT#3 32->16, ConvT#4 16->8 (2x2 stride 2, swish), output conv 3x3 SAME 8->1 (linear)
//   reference: sr-ae-conv.ipynb:c283-286 (layers), c306-320 (train_step: gradients of every trainable weight)
//
// kernel == stride for the three transposed convolutions, so a pixel of the 50x50 level owns its 8x8 block of the image
// and everything above it: the forward values are RECOMPUTED per tile from the stored 50x50x64 activation instead of
// being written by the forward pass and read back (5.12 + 2.56 + 1.28 MB per sample and tensor, three tensors each).
// Only the output conv looks across pixels, and only through dpred: a tile stages the 10x10 window of dpred around
// each of its pixels' 8x8 blocks.
//
// Work split: a tile = 16 consecutive pixels of the 50x50 level (flattened over the batch); four waves take the four
// taps (ty1, tx1) of ConvT#2 for the same tile, so each wave owns a quarter of every pixel's block; a workgroup is
// eight waves = two tile slots (the two waves of a SIMD work on different tiles).  All matrix work is
// v_mfma_f32_16x16x4_f32 (exact f32 products):
//   T form (rows = channels, columns = the 16 pixels; lane (n, g) register i = element (channel 4g + i, pixel n)):
//     the forward chain and the data gradients, as in tail32 -- an accumulator is the B operand of the next product;
//   P form (rows = pixels; lane (n, g) register i = element (pixel 4g + i, channel n)): both operands of a weight
//     gradient dW[co][ci] = sum_px dZ[px][co] X[px][ci] (the pixel index is the MFMA's k).  T -> P is one 16-byte LDS
//     store + four 4-byte loads per tile through a wave-private 16 x 20 float image (conflict-free both ways).
// The output conv's two gradients are banded (Toeplitz) MFMA products over the 3 x 4 dpred window that serves both
// columns tx3 of a row pair (on the vector unit they cost 576 FMAs per item and 72 more registers per lane).
// Weight gradients accumulate in registers over all tiles of a wave (ConvT#2: the wave's own tap, 32 registers;
// ConvT#3 / #4 / output conv / biases: partial sums per wave), are summed over lanes and waves in a fixed order at the
// end and leave as ONE slab per workgroup in flat-parameter order; train.hip's wgrad_finish_all adds the slabs.  No
// float atomics: gradients are bit-identical run to run.
// The data gradient of ConvT#2's input sums over the four taps = the four waves of a slot: through LDS, then
// x swish'(Z1) and out.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <stdexcept>

#include "kernels16.h"  // lds_attr_once
#include "train_tail.h"

namespace srcfd {

typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int TB_PITCH = 101;                       // dpred window image: [16 pixels][10 x 10], odd pitch
constexpr int TB_YP = 68;                           // activation tile image: [16 pixels][64 channels + 4]
constexpr int L_WF = 0;
constexpr int L_WB = L_WF + TT_WF;                  // 10752
constexpr int L_WT = L_WB + TT_WB;                  // 21504: Toeplitz fragments of the output conv, 3 k-steps
constexpr int L_DP = L_WT + TT_WT;                  // 21696: two tile slots x 1664
constexpr int L_Y1 = L_DP + 2 * 1664;               // two slots x 16 x 68
constexpr int L_TR = L_Y1 + 2 * 16 * TB_YP;         // wave-private transposition images: 8 waves x 2 x 320 floats
constexpr int L_RED = L_TR + 8 * 640;               // dY1 partial sums of the four taps: [slot][wave][tile][lane] float4
constexpr int L_END = L_RED + 2 * 4 * 4 * 64 * 4;   // 40512 floats
constexpr int TB_LDS = L_END * 4;                   // 162048 B: one workgroup per CU
static_assert(TB_LDS <= 160 * 1024, "tail_bwd32 LDS budget");
// end-of-kernel reduction images (the weight images are dead by then)
constexpr int E_RW = 44;                            // registers per wave: dW3 (32) | dW4 (8) | Toeplitz gradient of the output conv (4)
constexpr int E_R40 = 0;                            // [wave][44][lane]
constexpr int E_R2 = 8 * E_RW * 64;                 // [wave][64]: db4 8 | db3 16 | db2 32 | dbc 1
constexpr int E_R2S = 64;
static_assert(E_R2 + 8 * E_R2S <= L_TR && L_TR + 4 * 32 * 64 <= L_END, "reduction images fit below the parking area of ConvT#2's partial sums");

__device__ __forceinline__ f4 mf(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// y = z sigmoid(z), gd = swish'(z) = s + y (1 - s); hardware exp2 / rcp (~1 ulp each; the gradient test's bar is 2e-4).
// z -> -inf: exp2 = inf, s = 0, y = -0, gd = 0; a NaN stays a NaN.
__device__ __forceinline__ void act(const f4 z, f4& y, f4& gd) {
  f4 e, s;
#pragma unroll
  for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(z[i] * -1.4426950408889634f);
  e = e + 1.0f;
#pragma unroll
  for (int i = 0; i < 4; ++i) s[i] = __builtin_amdgcn_rcpf(e[i]);
  y = z * s;
  gd = __builtin_elementwise_fma(y, 1.0f - s, s);
}

// T form -> P form of one 16 x 16 tile through the wave's LDS image (LDS instructions of one wave execute in order:
// no barrier; the compiler keeps the order because the accesses may alias)
__device__ __forceinline__ f4 t2p(float* img, const int w_off, const int r_off, const f4 v) {
  *reinterpret_cast<f4*>(img + w_off) = v;
  f4 r;
#pragma unroll
  for (int i = 0; i < 4; ++i) r[i] = img[r_off + 20 * i];
  return r;
}

__device__ __forceinline__ float red16(float v) {   // over the 16 pixels of a lane group
#pragma unroll
  for (int o = 1; o < 16; o <<= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// Eight waves: two tile slots x the four taps of ConvT#2; the two waves of a SIMD work on different tiles (one wave's LDS
// round trips and vector work beside the other's matrix work).  256 registers per wave.
__global__ void __launch_bounds__(512) tail_bwd32(TailBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, g = lane >> 4, h = g & 1;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int slot = wave >> 2, tap1 = wave & 3, ty1 = tap1 >> 1, tx1 = tap1 & 1, stid = tid & 255;
  const int H = p.H, W = p.W, HW = H * W, OW = 8 * W, OH = 8 * H;
  const int NP = p.n * HW, ntiles = (NP + 15) >> 4;

  for (int i = tid; i < (TT_WF + TT_WB + TT_WT) / 4; i += 512) {
    f4 v;
    if (i < TT_WF / 4) v = reinterpret_cast<const f4*>(p.wf)[i];
    else if (i < (TT_WF + TT_WB) / 4) v = reinterpret_cast<const f4*>(p.wb)[i - TT_WF / 4];
    else v = reinterpret_cast<const f4*>(p.wt)[i - (TT_WF + TT_WB) / 4];
    reinterpret_cast<f4*>(sm)[i] = v;
  }
  f4 bz2[2], bz3, bz4;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    bz2[0][i] = p.bias[4 * g + i]; bz2[1][i] = p.bias[16 + 4 * g + i];
    bz3[i] = p.bias[32 + 4 * g + i];
    bz4[i] = p.bias[48 + 4 * h + i];
  }
  const float* w1f = sm + L_WF + tap1 * (2 * 16 * 64) + lane;     // [(t*16 + s) * 64]
  const float* w2f = sm + L_WF + 8192 + lane;                     // [(tap2*8 + ks) * 64]
  const float* w3f = sm + L_WF + 8192 + 2048 + lane;              // [(u*4 + i) * 64]
  const float* a1b = sm + L_WB + tap1 * (4 * 8 * 64) + lane;      // [((t*2 + c)*4 + i) * 64]
  const float* a2b = sm + L_WB + 8192 + lane;                     // [((tap2*2 + t)*4 + i) * 64]
  const float* a3b = sm + L_WB + 8192 + 2048 + lane;              // [(u*4 + i) * 64]
  const float* wtp = sm + L_WT + lane;                            // [s * 64]
  float* tr0 = sm + L_TR + wave * 640;
  float* tr1 = tr0 + 320;
  const int tr_w = n * 20 + 4 * g, tr_r = 80 * g + n;
  float* dpi = sm + L_DP + slot * 1664;
  float* y1i = sm + L_Y1 + slot * (16 * TB_YP);
  // dpred window of this lane's pixel, T form (B operand of the Toeplitz product): row (2 ty2 + a), column 2 tx2 + b' with b' = g
  const float* dpT = dpi + n * TB_PITCH + 40 * ty1 + 4 * tx1 + g;
  // P form (B operand of the output conv's weight gradient): pixel 4g + i, window element (a, b') = (n >> 2, n & 3)
  const float* dpP = dpi + 4 * g * TB_PITCH + 40 * ty1 + 4 * tx1 + 10 * (n >> 2) + (n & 3);
  f4* red = reinterpret_cast<f4*>(sm + L_RED) + slot * (4 * 4 * 64);

  // weight-gradient accumulators, over all tiles of this wave
  f4 dW2[2][4], dW3[4][2], dW4[2], gT, db2[2], db3, db4;
  float dbc = 0.f;
  const f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    db2[c] = zero; dW4[c] = zero;
#pragma unroll
    for (int t = 0; t < 4; ++t) { dW2[c][t] = zero; dW3[t][c] = zero; }
  }
  gT = zero; db3 = zero; db4 = zero;

  for (int tp = (int)blockIdx.x; 2 * tp < ntiles; tp += (int)gridDim.x) {
    const int tile = 2 * tp + slot;
    const bool tile_on = tile < ntiles;   // wave-uniform
    const int P = 16 * tile + n;
    const bool px_ok = P < NP;
    const int Pc = px_ok ? P : NP - 1;
    // ---- stage the dpred windows of the tile's pixels (zero outside the image = the conv's SAME padding, and for the
    // pixels past the end of the batch: everything they would add to a gradient is then zero) and the activation tile ----
    for (int e = stid; e < 1600; e += 256) {
      const int nn = e / 100, rc = e - nn * 100, r = rc / 10, c = rc - r * 10;
      const int PP = 16 * tile + nn;
      float v = 0.f;
      if (PP < NP) {
        const int smp = (int)__umulhi((unsigned)PP, p.magic_hw), rem = PP - smp * HW, y = (int)__umulhi((unsigned)rem, p.magic_w), x = rem - y * W;
        const int Y = 8 * y - 1 + r, X = 8 * x - 1 + c;
        if (Y >= 0 && Y < OH && X >= 0 && X < OW) v = p.dpred[((size_t)smp * OH + Y) * OW + X];
      }
      dpi[nn * TB_PITCH + rc] = v;
      if (r >= 1 && r <= 8 && c >= 1 && c <= 8) dbc += v;   // the interior of a window is the pixel's own block: every image pixel once
    }
    {
      const int nn = stid >> 4, c4 = (stid & 15) * 4;
      const f4 v = *reinterpret_cast<const f4*>(p.y1 + (size_t)min(16 * tile + nn, NP - 1) * 64 + c4);
      *reinterpret_cast<f4*>(y1i + nn * TB_YP + c4) = v;
    }
    __syncthreads();   // A: images staged (and the previous tile's reduction image read)

    f4 dy1[4] = {zero, zero, zero, zero};
    if (tile_on) {
      // ================= this wave's tap of ConvT#2 =================
      f4 z2[2] = {bz2[0], bz2[1]};
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f4 xq = *reinterpret_cast<const f4*>(y1i + n * TB_YP + 16 * g + 4 * q);   // channel 16g + s, s = 4q + j
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          z2[0] = mf(w1f[(4 * q + j) * 64], xq[j], z2[0]);
          z2[1] = mf(w1f[(16 + 4 * q + j) * 64], xq[j], z2[1]);
        }
      }
      f4 y2[2], g2[2], y2p[2], dy2[2] = {zero, zero};
      act(z2[0], y2[0], g2[0]); act(z2[1], y2[1], g2[1]);
      y2p[0] = t2p(tr0, tr_w, tr_r, y2[0]);
      y2p[1] = t2p(tr1, tr_w, tr_r, y2[1]);
#pragma unroll
      for (int tap2 = 0; tap2 < 4; ++tap2) {
        const int ty2 = tap2 >> 1, tx2 = tap2 & 1;
        // ---- forward: ConvT#3, ConvT#4 of this sub-tree ----
        f4 z3 = bz3;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) z3 = mf(w2f[(tap2 * 8 + ks) * 64], ks < 4 ? y2[0][ks & 3] : y2[1][ks & 3], z3);
        f4 y3, g3;
        act(z3, y3, g3);
        f4 z4[2] = {bz4, bz4}, y4[2], g4[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int i = 0; i < 4; ++i) z4[u] = mf(w3f[(u * 4 + i) * 64], y3[i], z4[u]);
          act(z4[u], y4[u], g4[u]);
        }
        // ---- output conv, banded (Toeplitz) over the 3 x 4 dpred window that serves both columns tx3 of a row pair:
        //   dY4[(tx3, co)][px] = sum_{a, b'} Wt[(tx3, co)][(a, b')] dpred[row + a - 1][col0 + b' - 1],  Wt = Wc[2-a][2-(b'-tx3)][co] or 0
        //   gT[(tx3, co)][(a, b')] += sum_px Y4[px][(tx3, co)] dpred[...]      (unfolded into dWc at the end) ----
        float dpr[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) dpr[a] = dpT[(2 * ty2 + a) * 10 + 2 * tx2];
        f4 dz4[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          f4 dy4 = zero;
#pragma unroll
          for (int s = 0; s < 3; ++s) dy4 = mf(wtp[s * 64], dpr[u + s], dy4);
          dz4[u] = dy4 * g4[u];
          db4 = db4 + dz4[u];
          const f4 y4p = t2p(u ? tr1 : tr0, tr_w, tr_r, y4[u]);
#pragma unroll
          for (int i = 0; i < 4; ++i) gT = mf(y4p[i], dpP[i * TB_PITCH + (2 * ty2 + u) * 10 + 2 * tx2], gT);
        }
        // ---- ConvT#4: data gradient, weight gradient ----
        f4 dy3 = zero;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
          for (int i = 0; i < 4; ++i) dy3 = mf(a3b[(u * 4 + i) * 64], dz4[u][i], dy3);
        const f4 dz3 = dy3 * g3;
        db3 = db3 + dz3;
        const f4 y3p = t2p(tr0, tr_w, tr_r, y3);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          const f4 dzp = t2p(tr1, tr_w, tr_r, dz4[u]);
#pragma unroll
          for (int i = 0; i < 4; ++i) dW4[u] = mf(dzp[i], y3p[i], dW4[u]);
        }
        // ---- ConvT#3: weight gradient, data gradient ----
        const f4 dz3p = t2p(tr0, tr_w, tr_r, dz3);
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dW3[tap2][t] = mf(dz3p[i], y2p[t][i], dW3[tap2][t]);
            dy2[t] = mf(a2b[((tap2 * 2 + t) * 4 + i) * 64], dz3[i], dy2[t]);
          }
      }
      // ---- ConvT#2 (this wave's tap): weight gradient, data gradient ----
      f4 dz2[2], dz2p[2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        dz2[c] = dy2[c] * g2[c];
        db2[c] = db2[c] + dz2[c];
      }
      dz2p[0] = t2p(tr0, tr_w, tr_r, dz2[0]);
      dz2p[1] = t2p(tr1, tr_w, tr_r, dz2[1]);
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        f4 y1p;                                   // P form of channel tile t: pixel 4g + i, channel 16t + n
#pragma unroll
        for (int i = 0; i < 4; ++i) y1p[i] = y1i[(4 * g + i) * TB_YP + 16 * t + n];
#pragma unroll
        for (int c = 0; c < 2; ++c)
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            dW2[c][t] = mf(dz2p[c][i], y1p[i], dW2[c][t]);
            dy1[t] = mf(a1b[((t * 2 + c) * 4 + i) * 64], dz2[c][i], dy1[t]);
          }
      }
    }
    // ---- dY1 = sum over the four taps (waves, in wave order); wave w finishes channel tile w: x swish'(Z1), store ----
#pragma unroll
    for (int t = 0; t < 4; ++t) red[(tap1 * 4 + t) * 64 + lane] = dy1[t];
    const f4 z1 = *reinterpret_cast<const f4*>(p.z1 + (size_t)Pc * 64 + 16 * tap1 + 4 * g);
    __syncthreads();   // B
    {
      f4 s = red[(0 * 4 + tap1) * 64 + lane];
#pragma unroll
      for (int w = 1; w < 4; ++w) s = s + red[(w * 4 + tap1) * 64 + lane];
      f4 yy, gd;
      act(z1, yy, gd);
      if (px_ok) *reinterpret_cast<f4*>(p.dz1 + (size_t)P * 64 + 16 * tap1 + 4 * g) = s * gd;
    }
  }

  // ================= end: one slab of TT_PARAMS floats per workgroup, flat-parameter order =================
  float* slab = p.slabs + (size_t)blockIdx.x * TT_PARAMS;
  __syncthreads();   // the weight images are dead
  {
    float* r40 = sm + E_R40 + wave * E_RW * 64 + lane;
#pragma unroll
    for (int tap2 = 0; tap2 < 4; ++tap2)
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) r40[((tap2 * 2 + t) * 4 + j) * 64] = dW3[tap2][t][j];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int j = 0; j < 4; ++j) r40[(32 + u * 4 + j) * 64] = dW4[u][j];
#pragma unroll
    for (int j = 0; j < 4; ++j) r40[(40 + j) * 64] = gT[j];
    float* r2 = sm + E_R2 + wave * E_R2S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float v = red16(db4[i]);
      v += __shfl_xor(v, 32, 64);              // the two tx3 halves
      if (n == 0 && g < 2) r2[4 * h + i] = v;
      v = red16(db3[i]);
      if (n == 0) r2[8 + 4 * g + i] = v;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        v = red16(db2[c][i]);
        if (n == 0) r2[24 + 16 * c + 4 * g + i] = v;
      }
    }
    float v = red16(dbc);
    v += __shfl_xor(v, 16, 64);
    v += __shfl_xor(v, 32, 64);
    if (lane == 0) r2[56] = v;
  }
  // ConvT#2's kernel: the two waves of tap w (slots 0 / 1) hold partial sums over different tiles: slot 1 parks its registers
  // in the dead dY1 / transposition images, slot 0 adds and stores.  D[co 16c + 4g + j][ci 16t + n] of tap `tap1`.
  float* park = sm + L_TR + tap1 * (32 * 64) + lane;   // 4 taps x 32 registers x 64 lanes = 8192 floats <= L_END - L_TR
  if (slot == 1) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) park[((c * 4 + t) * 4 + j) * 64] = dW2[c][t][j];
  }
  __syncthreads();
  if (slot == 0) {
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j)
          slab[TT_O_W1 + (tap1 * 32 + 16 * c + 4 * g + j) * 64 + 16 * t + n] = dW2[c][t][j] + park[((c * 4 + t) * 4 + j) * 64];
  }
  const float* R40 = sm + E_R40;
  const float* R2 = sm + E_R2;
  auto sum8 = [&](const float* base, int idx, int stride) {   // over the eight waves, fixed order
    float v = base[idx];
#pragma unroll
    for (int w = 1; w < 8; ++w) v += base[w * stride + idx];
    return v;
  };
  for (int o = tid; o < 2048; o += 512) {        // ConvT#3 kernel (2, 2, 16, 32): flat (tap2*16 + co)*32 + ci
    const int tap2 = o >> 9, co = (o >> 5) & 15, ci = o & 31;
    const int idx = (((tap2 * 2 + (ci >> 4)) * 4 + (co & 3)) * 64) + 16 * (co >> 2) + (ci & 15);
    slab[TT_O_W2 + o] = sum8(R40, idx, E_RW * 64);
  }
  {                                              // ConvT#4 kernel (2, 2, 8, 16): flat (tap3*8 + co)*16 + ci; D row = 8 tx3 + co of tile u = ty3
    const int o = tid;
    const int tap3 = o >> 7, co = (o >> 4) & 7, ci = o & 15;
    const int row = 8 * (tap3 & 1) + co;
    const int idx = ((32 + (tap3 >> 1) * 4 + (row & 3)) * 64) + 16 * (row >> 2) + ci;
    slab[TT_O_W3 + o] = sum8(R40, idx, E_RW * 64);
  }
  if (tid < 72) {                                // output conv (3, 3, 8): dWc[dy][dx][co] = sum_tx3 gT[(tx3, co)][(a, b') = (2 - dy, 2 - dx + tx3)]
    const int dy = tid / 24, dx = (tid >> 3) % 3, co = tid & 7;
    float v = 0.f;
#pragma unroll
    for (int tx3 = 0; tx3 < 2; ++tx3) {
      const int row = 8 * tx3 + co, col = 4 * (2 - dy) + (2 - dx + tx3);
      v += sum8(R40, (40 + (row & 3)) * 64 + 16 * (row >> 2) + col, E_RW * 64);
    }
    slab[TT_O_WC + tid] = v;
  } else if (tid >= 128 && tid < 128 + 57) {
    const int o = tid - 128;
    const float v = sum8(R2, o, E_R2S);
    if (o < 8) slab[TT_O_B3 + o] = v;
    else if (o < 24) slab[TT_O_B2 + o - 8] = v;
    else if (o < 56) slab[TT_O_B1 + o - 24] = v;
    else slab[TT_O_BC] = v;
  }
}

int tail_bwd32_blocks(int n, int H, int W, int num_cus) {
  const int64_t pairs = (((int64_t)n * H * W + 15) / 16 + 1) / 2;   // a workgroup takes two tiles at a time
  return (int)std::max<int64_t>(1, std::min<int64_t>(pairs, num_cus));
}

hipError_t launch_tail_bwd32(const TailBwdParams& p, int num_cus, hipStream_t s) {
  if (p.n <= 0) return hipSuccess;
  // x / d as umulhi(x, ceil(2^32 / d)): exact while x (ceil(2^32 / d) d - 2^32) < 2^32, i.e. for every x < 2^32 / d
  const uint64_t hw = (uint64_t)p.H * p.W;
  if (p.W < 1 || p.H < 1 || (uint64_t)p.n * hw * hw >= (1ull << 32)) return hipErrorInvalidValue;
  TailBwdParams q = p;
  q.magic_hw = (unsigned)(((1ull << 32) + hw - 1) / hw);
  q.magic_w = (unsigned)(((1ull << 32) + p.W - 1) / p.W);
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(tail_bwd32), TB_LDS);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(tail_bwd32, dim3(tail_bwd32_blocks(p.n, p.H, p.W, num_cus)), dim3(512), TB_LDS, s, q);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// host: which flat parameter (and which factor) lands in which pack slot
// ---------------------------------------------------------------------------
void train_tail_plan(const ModelDesc& desc, const int* desc_index, const size_t* kernel_off, const size_t* bias_off, int n_compute_layers,
                     TrainTailPlan& plan) {
  plan = TrainTailPlan();
  if (n_compute_layers < 5) return;
  const int f = n_compute_layers - 4;
  const Layer &L1 = desc.layers[desc_index[f]], &L2 = desc.layers[desc_index[f + 1]], &L3 = desc.layers[desc_index[f + 2]],
              &LO = desc.layers[desc_index[f + 3]];
  auto convt = [](const Layer& L, int cin, int cout) {
    return L.kind == SRCFD_LAYER_CONV2D_TRANSPOSE && L.kh == 2 && L.kw == 2 && L.stride == 2 && !L.same && L.cin == cin && L.cout == cout &&
           L.act == SRCFD_ACT_SWISH;
  };
  if (!convt(L1, 64, 32) || !convt(L2, 32, 16) || !convt(L3, 16, 8)) return;
  if (LO.kind != SRCFD_LAYER_CONV2D || LO.kh != 3 || LO.kw != 3 || LO.stride != 1 || !LO.same || LO.cin != 8 || LO.cout != 1 ||
      LO.act != SRCFD_ACT_LINEAR)
    return;
  if (desc_index[f + 1] != desc_index[f] + 1 || desc_index[f + 2] != desc_index[f] + 2 || desc_index[f + 3] != desc_index[f] + 3) return;
  if (L1.in_shape[1] > 50) return;   // tail32's LDS ring holds rows of up to 400 pixels
  // the four layers' parameters must be one contiguous run in flat order (they are: kernel, bias per layer)
  const size_t o = kernel_off[f];
  if (bias_off[f] != o + TT_O_B1 || kernel_off[f + 1] != o + TT_O_W2 || bias_off[f + 1] != o + TT_O_B2 || kernel_off[f + 2] != o + TT_O_W3 ||
      bias_off[f + 2] != o + TT_O_B3 || kernel_off[f + 3] != o + TT_O_WC || bias_off[f + 3] != o + TT_O_BC)
    return;
  plan.first_layer = f;
  plan.H = L1.in_shape[0]; plan.W = L1.in_shape[1];
  plan.param_off = o;
  const double LOG2E = 1.4426950408889634;
  auto& mp = plan.map;
  auto& sc = plan.scale;
  auto align = [&]() { while (mp.size() % 64) { mp.push_back(0); sc.push_back(0.f); } };
  auto put = [&](size_t flat, double scale) { mp.push_back((int)(flat + 1)); sc.push_back((float)scale); };
  const size_t W1 = o + TT_O_W1, B1 = o + TT_O_B1, W2 = o + TT_O_W2, B2 = o + TT_O_B2, W3 = o + TT_O_W3, B3 = o + TT_O_B3, WC = o + TT_O_WC,
               BC = o + TT_O_BC;
  // Conv2DTranspose kernels are (kh, kw, Cout, Cin): flat (tap*Cout + co)*Cin + ci
  auto w1 = [&](int tap, int co, int ci) { return W1 + ((size_t)tap * 32 + co) * 64 + ci; };
  auto w2 = [&](int tap, int co, int ci) { return W2 + ((size_t)tap * 16 + co) * 32 + ci; };
  auto w3 = [&](int tap, int co, int ci) { return W3 + ((size_t)tap * 8 + co) * 16 + ci; };
  // fragment lane maps of engine.hip plan_tail32 (lane = (m = lane & 15, kg = lane >> 4)), with factor `s1` on the first layer
  auto fwd_frags = [&](double s1) {
    for (int tap = 0; tap < 4; ++tap)
      for (int t = 0; t < 2; ++t)
        for (int s = 0; s < 16; ++s)
          for (int lane = 0; lane < 64; ++lane) put(w1(tap, 16 * t + (lane & 15), 16 * (lane >> 4) + s), s1);
  };
  auto fwd_frags2 = [&]() {
    for (int tap = 0; tap < 4; ++tap)
      for (int t = 0; t < 2; ++t)
        for (int i = 0; i < 4; ++i)
          for (int lane = 0; lane < 64; ++lane) put(w2(tap, lane & 15, 16 * t + 4 * (lane >> 4) + i), 1.0);
  };
  auto fwd_frags3 = [&]() {
    for (int u = 0; u < 2; ++u)
      for (int i = 0; i < 4; ++i)
        for (int lane = 0; lane < 64; ++lane) put(w3(2 * u + ((lane & 15) >> 3), lane & 7, 4 * (lane >> 4) + i), 1.0);
  };
  // ---- tail32's operands (swish layers produce log2(e) x: kernels_tail32.hip, swish_l2e) ----
  align(); plan.t32_w1 = mp.size(); fwd_frags(LOG2E);
  align(); plan.t32_b1 = mp.size(); for (int c = 0; c < 32; ++c) put(B1 + c, LOG2E);
  align(); plan.t32_w2 = mp.size(); fwd_frags2();
  align(); plan.t32_b2 = mp.size(); for (int c = 0; c < 16; ++c) put(B2 + c, LOG2E);
  align(); plan.t32_w3 = mp.size(); fwd_frags3();
  align(); plan.t32_b3 = mp.size(); for (int c = 0; c < 8; ++c) put(B3 + c, LOG2E);
  align(); plan.t32_wc = mp.size(); for (int k = 0; k < 72; ++k) put(WC + k, 1.0 / LOG2E); put(BC, 1.0);
  // ---- tail_bwd32: unscaled forward fragments, data-gradient fragments, biases ----
  align(); plan.wf = mp.size(); fwd_frags(1.0); fwd_frags2(); fwd_frags3();
  align(); plan.wb = mp.size();
  for (int tap = 0; tap < 4; ++tap)      // a1b[tap][t][c][i][lane] = W1[tap][co 16c + 4kg + i][ci 16t + m]
    for (int t = 0; t < 4; ++t)
      for (int c = 0; c < 2; ++c)
        for (int i = 0; i < 4; ++i)
          for (int lane = 0; lane < 64; ++lane) put(w1(tap, 16 * c + 4 * (lane >> 4) + i, 16 * t + (lane & 15)), 1.0);
  for (int tap = 0; tap < 4; ++tap)      // a2b[tap][t][i][lane] = W2[tap][co 4kg + i][ci 16t + m]
    for (int t = 0; t < 2; ++t)
      for (int i = 0; i < 4; ++i)
        for (int lane = 0; lane < 64; ++lane) put(w2(tap, 4 * (lane >> 4) + i, 16 * t + (lane & 15)), 1.0);
  for (int u = 0; u < 2; ++u)            // a3b[u][i][lane] = W3[tap3 = 2u + (r >> 3)][co r & 7][ci m], r = 4kg + i
    for (int i = 0; i < 4; ++i)
      for (int lane = 0; lane < 64; ++lane) {
        const int r = 4 * (lane >> 4) + i;
        put(w3(2 * u + (r >> 3), r & 7, lane & 15), 1.0);
      }
  align(); plan.wt = mp.size();
  for (int s3 = 0; s3 < 3; ++s3)         // wt[s][lane] = Wc[2 - s][2 - (b' - tx3)][co] for 0 <= b' - tx3 <= 2, else 0; m = 8 tx3 + co, b' = kg
    for (int lane = 0; lane < 64; ++lane) {
      const int m = lane & 15, tx3 = m >> 3, co = m & 7, d = (lane >> 4) - tx3;
      if (d >= 0 && d <= 2) put(WC + (size_t)((2 - s3) * 3 + (2 - d)) * 8 + co, 1.0);
      else { mp.push_back(0); sc.push_back(0.f); }
    }
  align(); plan.bias = mp.size();
  for (int c = 0; c < 32; ++c) put(B1 + c, 1.0);
  for (int c = 0; c < 16; ++c) put(B2 + c, 1.0);
  for (int c = 0; c < 8; ++c) put(B3 + c, 1.0);
  for (int k = 0; k < 72; ++k) put(WC + k, 1.0);
  put(BC, 1.0);
  align();
  if (mp.size() - plan.wf < (size_t)TT_WF || plan.wt - plan.wb != (size_t)TT_WB || plan.bias - plan.wt != (size_t)TT_WT) throw std::runtime_error("train_tail_plan: pack sizes");
  plan.ok = true;
}

}  // namespace srcfd
