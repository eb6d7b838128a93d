// gemm_x3: f32-grade implicit GEMM on the bf16 matrix cores (gfx950 only) -- precision SRCFD_PREC_FP32X3.
//
// Why: v_mfma_f32_*_f32 runs at 1/16 of the bf16 MFMA rate on this chip and does not overlap vector work, and the two wide
// decoder layers (ConvT#0 3x3 s2 256->128 and ConvT#1 2x2 s2 128->64; SURVEY.md 8a rows a13, a14: 45 % of the network's MACs)
// are pure matrix time: 1.08 ms of the f32 parity path's 3.04 ms per 768 samples.  A float32 value is EXACTLY the sum of three
// bfloat16 values (8 + 8 + 8 significant bits: hi = top 16 bits, mid = top 16 bits of x - hi, lo = x - hi - mid), so
//     x w = (xh + xm + xl)(wh + wm + wl) = xh wh + xh wm + xm wh + xh wl + xl wh + xm wm  + terms below 2^-24 |x w|
// six bf16 MFMAs with f32 accumulation instead of one f32 MFMA: 6/16 of the matrix time, and products that are exact in
// f32 before they are summed.  Measured on the CPU emulation (tests/split_precision_study.py, profiles/r04/n_...): 3e-7
// relative L2 on the whole network against float64 -- tighter than plain f32 accumulation (5e-7) -- where two terms
// (3 MFMAs) give 1.3e-5 and miss the 1e-5 bar.
//
// Structure: activations stay float32 in HBM (the neighbours -- dense_skinny32 in front, tail32 behind -- are f32 kernels) and are
// split in registers (truncation split: two ANDs and two exact subtractions per element; the packs take the high halves);
// weights are split on the host ([plane][n][Kpad] bf16) and are the only operand that goes through LDS.  See the kernel.
// Epilogue: bias + swish in f32, rows leave as whole 128-byte lines through a wave-private LDS image.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstring>

#include "dev16.h"
#include "kernels.h"
#include "kernels16.h"   // lds_attr_once

namespace srcfd {

constexpr int X3_BP = 128, X3_BN = 128, X3_BK = 32, X3_PITCH = 40;     // weight rows in LDS: 32 k + 8 pad = 80 B (5 x 16 B: conflict-free b128 reads)
constexpr int X3_PLANE = X3_BN * X3_PITCH;                              // elements per weight plane
constexpr int X3_BUF = 3 * X3_PLANE;                                    // one stage: three planes
constexpr int X3_WLDS = 2 * X3_BUF * 2;                                 // two stages: 61 440 B
constexpr int X3_EP = 36;                                               // wave image pitch (floats): 32 + 4 (144 B = 9 x 16 B: conflict-free)
constexpr int X3_LDS = X3_WLDS + 4 * 32 * X3_EP * 4;                    // + one image per wave: 79 872 B -> two workgroups per CU

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));   // native vectors: arrays of HIP's uint4 (a struct) end up in scratch
__device__ __forceinline__ f32x16 mfma_bf(const u32x4& a, const u32x4& b, const f32x16& c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
}

__device__ __forceinline__ float swish_f32(float z) { return z * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f)); }

// x = hi + mid + lo exactly: hi / mid are x / the first remainder with their low 16 bits cleared, lo is the second remainder (at
// most 8 significant bits are left in it).  Eight consecutive k values of one pixel -> one MFMA B fragment per plane; the packs
// take the HIGH halves (v_perm_b32), which for lo is the same truncation.
__device__ __forceinline__ void split8(const f32x4& v0, const f32x4& v1, u32x4& fh, u32x4& fm, u32x4& fl) {
  uint32_t hm[8], mm[8], lm[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const float x = i < 4 ? v0[i & 3] : v1[i & 3];
    hm[i] = __builtin_bit_cast(uint32_t, x) & 0xffff0000u;
    const float r1 = x - __builtin_bit_cast(float, hm[i]);
    mm[i] = __builtin_bit_cast(uint32_t, r1) & 0xffff0000u;
    lm[i] = __builtin_bit_cast(uint32_t, r1 - __builtin_bit_cast(float, mm[i]));
  }
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    fh[q] = __builtin_amdgcn_perm(hm[2 * q + 1], hm[2 * q], 0x07060302u);
    fm[q] = __builtin_amdgcn_perm(mm[2 * q + 1], mm[2 * q], 0x07060302u);
    fl[q] = __builtin_amdgcn_perm(lm[2 * q + 1], lm[2 * q], 0x07060302u);
  }
}

// Epilogue of one wave's 32 pixels x 128 channels: the two accumulator sets meet, bias + activation in f32.  A lane owns one pixel
// and 4 consecutive channels per register quad; each 32 x 32 sub-tile goes through the wave's LDS image and leaves as 32 rows of
// 128 contiguous bytes.  orow[it]: output element offset of pixel (lane >> 3) + 8 it without the phase / channel part, -1 = no pixel.
__device__ __forceinline__ void x3_epilogue(const GemmDesc& d, const f32x16 (&accH)[4], const f32x16 (&accR)[4], const float* __restrict__ bias,
                                            float* __restrict__ Y, float* image, const int n0, const int lane, const int (&orow)[4]) {
  const int h = lane >> 5, l31 = lane & 31;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const int nb = n0 + a * 32;   // wave-uniform
    if (nb >= d.N) continue;
    const float* bp = bias + nb + 4 * h;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * q);
      f32x4 v = {accH[a][4 * q] + accR[a][4 * q] + bv.x, accH[a][4 * q + 1] + accR[a][4 * q + 1] + bv.y, accH[a][4 * q + 2] + accR[a][4 * q + 2] + bv.z,
                 accH[a][4 * q + 3] + accR[a][4 * q + 3] + bv.w};
      if (d.act == SRCFD_ACT_SWISH) {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = swish_f32(v[i]);
      }
      *reinterpret_cast<f32x4*>(image + l31 * X3_EP + 8 * q + 4 * h) = v;
    }
    const int ph = nb / d.CO, co = nb - ph * d.CO, py = ph / d.nphx, px = ph - py * d.nphx;
    const int poff = (py * d.OW + px) * d.OC + co + 4 * (lane & 7);
#pragma unroll
    for (int it = 0; it < 4; ++it) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(image + ((lane >> 3) + 8 * it) * X3_EP + 4 * (lane & 7));
      if (orow[it] >= 0) *reinterpret_cast<f32x4*>(Y + (orow[it] + poff)) = v;
    }
  }
}

// A wave's 32 pixels x 32 k of one k-tile come from HBM as whole 128-byte row segments (lane -> pixel (lane >> 3) + 8 j, 16-byte chunk
// lane & 7: 8 rows per instruction), are parked in the wave's LDS image (the epilogue's: 32 rows of 36 floats) and read back in
// MFMA fragment shape (lane -> pixel lane & 31, 8 consecutive k).  Loading the fragment shape straight from HBM (32 rows x two
// 16-byte pieces per instruction) thrashed L1 -- eight waves x 16 KB of row pieces per CU -- and ran 10 % slower.
__device__ __forceinline__ void x3_park_and_split(float* image, const int lane, const f32x4 (&xr)[4], u32x4 (&bh)[2], u32x4 (&bm)[2], u32x4 (&bl)[2]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(image + ((lane >> 3) + 8 * j) * X3_EP + 4 * (lane & 7)) = xr[j];
  const float* src = image + (lane & 31) * X3_EP + 8 * (lane >> 5);
#pragma unroll
  for (int kk = 0; kk < 2; ++kk) {
    const f32x4 v0 = *reinterpret_cast<const f32x4*>(src + 16 * kk), v1 = *reinterpret_cast<const f32x4*>(src + 16 * kk + 4);
    split8(v0, v1, bh[kk], bm[kk], bl[kk]);
  }
}

// Workgroup tile 128 pixels x 128 channels, four waves of 32 pixels x 128 channels.  A wave stages and splits its own pixels (loaded
// one k-tile ahead, parked in a wave-private LDS image, split in registers: every pixel is split exactly once, no barrier involved);
// only the weights are shared by the four waves: three planes, two LDS stages, ONE barrier per k-tile.
// Two accumulator sets: products 2^-8 and 2^-16 below the hi x hi ones lose their low bits when they are added to a running
// hi x hi sum -- each of an MFMA's 16 products is cut at the accumulator's last bit on the way into the adder, not the sum once
// (tools/microbench11.hip mode 3, profiles/r04/o_...: 2.5e-4 of the small plane's sum against 6e-7 in an accumulator of its own).
// With all six products in one accumulator a layer came out at 3e-5 where the arithmetic is good to 1.5e-7.  So the hi x hi
// products accumulate alone and the five small products in a set of their own (2^-8 terms last); the sets meet in one f32 add.
__device__ __forceinline__ void gemm_x3_tile(const GemmDesc& d, const float* __restrict__ X, const uint16_t* __restrict__ Wt, const int Kpad, const int64_t wplane,
                                             const float* __restrict__ bias, float* __restrict__ Y, const int bx, const int by) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  uint16_t* Ws = reinterpret_cast<uint16_t*>(gsm);                 // [stage 2][plane 3][X3_BN][X3_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int m0 = bx * X3_BP, n0 = by * X3_BN;

  // lane l31 of either half decodes pixel l31 of the wave's tile; the lanes that load / store a row fetch its numbers by shuffle
  const int m = m0 + wave * 32 + l31;
  int img = -1, my = 0, mx = 0;
  if (m < d.M) {
    const int per = d.MH * d.MW;
    img = m / per;
    const int r = m - img * per;
    my = r / d.MW;
    mx = r - my * d.MW;
  }
  const bool check = d.TY * d.TX > 1 || d.cy != 0 || d.cx != 0;   // taps that can leave the image
  const int by0 = my * d.ay + d.cy, bx0 = mx * d.ax + d.cx;
  const int xy_own = img >= 0 ? by0 : -(1 << 28);                  // rows past M never pass the bounds test
  int xoff_own = ((img * d.IH + by0) * d.IW + bx0) * d.CI;
  if (!check && img < 0) xoff_own = 0;                             // k == s layers: read row 0, nothing is stored
  const int o_own = img >= 0 ? ((img * d.OH + my * d.os + d.oy0) * d.OW + mx * d.os + d.ox0) * d.OC : -1;   // 32-bit: gemm_x3_qualifies bounds the batch
  int xo[4], xy[4], xx[4], orow[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int r = (lane >> 3) + 8 * j;
    xo[j] = __shfl(xoff_own, r, 64) + 4 * (lane & 7);
    xy[j] = __shfl(xy_own, r, 64);
    xx[j] = __shfl(bx0, r, 64);
    orow[j] = __shfl(o_own, r, 64);
  }
  // weight staging: rows (tid >> 2) and + 64, 16-byte chunk tid & 3 of the k-tile, per plane
  const int wrow = tid >> 2, c8 = tid & 3;
  const int woff0 = (n0 + wrow) * Kpad + c8 * 8, woff1 = (n0 + wrow + 64) * Kpad + c8 * 8;
  const int wlds = wrow * X3_PITCH + c8 * 8;
  float* image = reinterpret_cast<float*>(gsm + X3_WLDS) + wave * (32 * X3_EP);

  f32x4 xr[4];
  u32x4 wr[3][2];
  auto g2r = [&](int k0) {
    const int tap = k0 / d.CI, ci0 = k0 - tap * d.CI;
    const int ty = tap / d.TX, tx = tap - ty * d.TX;
    const int dy = ty * d.by, dx = tx * d.bx;
    const int toff = (dy * d.IW + dx) * d.CI + ci0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (!check || ((unsigned)(xy[j] + dy) < (unsigned)d.IH && (unsigned)(xx[j] + dx) < (unsigned)d.IW)) v = *reinterpret_cast<const f32x4*>(X + (xo[j] + toff));
      xr[j] = v;
    }
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      wr[p][0] = *reinterpret_cast<const u32x4*>(Wt + (p * wplane + woff0 + k0));
      wr[p][1] = *reinterpret_cast<const u32x4*>(Wt + (p * wplane + woff1 + k0));
    }
  };
  auto w2l = [&](int stage) {
    uint16_t* dst = Ws + stage * X3_BUF + wlds;
#pragma unroll
    for (int p = 0; p < 3; ++p) {
      *reinterpret_cast<u32x4*>(dst + p * X3_PLANE) = wr[p][0];
      *reinterpret_cast<u32x4*>(dst + p * X3_PLANE + 64 * X3_PITCH) = wr[p][1];
    }
  };

  f32x16 accH[4], accR[4];
#pragma unroll
  for (int a = 0; a < 4; ++a)
#pragma unroll
    for (int r = 0; r < 16; ++r) { accH[a][r] = 0.f; accR[a][r] = 0.f; }

  g2r(0);
  w2l(0);
  u32x4 bh[2], bm[2], bl[2];
  x3_park_and_split(image, lane, xr, bh, bm, bl);
  __syncthreads();
  const uint16_t* wsr = Ws + l31 * X3_PITCH + h * 8;
  int stage = 0;
  for (int k0 = 0; k0 < d.K; k0 += X3_BK) {
    const bool more = k0 + X3_BK < d.K;
    if (more) g2r(k0 + X3_BK);
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const uint16_t* wp = wsr + stage * X3_BUF + a * 32 * X3_PITCH + kk * 16;
        const u32x4 ah = *reinterpret_cast<const u32x4*>(wp), am = *reinterpret_cast<const u32x4*>(wp + X3_PLANE),
                    al = *reinterpret_cast<const u32x4*>(wp + 2 * X3_PLANE);
        f32x16 c = accR[a];
        c = mfma_bf(am, bm[kk], c);
        c = mfma_bf(ah, bl[kk], c);
        c = mfma_bf(al, bh[kk], c);
        c = mfma_bf(ah, bm[kk], c);
        c = mfma_bf(am, bh[kk], c);
        accR[a] = c;
        accH[a] = mfma_bf(ah, bh[kk], accH[a]);
      }
    }
    if (more) {
      w2l(stage ^ 1);    // the other stage was last read before the previous barrier
      x3_park_and_split(image, lane, xr, bh, bm, bl);
    }
    __syncthreads();
    stage ^= 1;
  }
  x3_epilogue(d, accH, accR, bias, Y, image, n0, lane, orow);
}

__global__ void __launch_bounds__(256, 2) gemm_x3(GemmDesc d, const float* __restrict__ X, const uint16_t* __restrict__ Wt, int Kpad, int64_t wplane,
                                                   const float* __restrict__ bias, float* __restrict__ Y) {
  gemm_x3_tile(d, X, Wt, Kpad, wplane, bias, Y, (int)blockIdx.x, (int)blockIdx.y);
}

// The GEMMs of one layer (the four output phases of ConvT#0: K = 4, 2, 2, 1 taps x 256) as ONE launch, longest first: as four
// launches each phase ends on a partly filled last round of workgroups (1014 / 936 / 936 / 864 workgroups on 512 slots) -- 13 % of
// the layer's time by the workgroups' own durations.  The table travels as a kernel argument.
struct X3Group {
  GemmDesc d[4];
  const uint16_t* Wt[4];
  const float* bias[4];
  int first[5];     // first linear workgroup id of op i; first[count] = total
  int count;
};
__global__ void __launch_bounds__(256, 2) gemm_x3_group(const X3Group g, const float* __restrict__ X, float* __restrict__ Y) {
  int op = 0;
  while (op + 1 < g.count && (int)blockIdx.x >= g.first[op + 1]) ++op;     // block-uniform: scalar compares on the kernel arguments
  const GemmDesc d = g.d[op];
  const int lb = (int)blockIdx.x - g.first[op], nby = d.N / X3_BN;
  const int Kpad = (d.K + X3_BK - 1) / X3_BK * X3_BK;
  gemm_x3_tile(d, X, g.Wt[op], Kpad, (int64_t)d.N * Kpad, g.bias[op], Y, lb / nby, lb - (lb / nby) * nby);
}

// Short-K layers without taps (ConvT#1: K = 128, kernel == stride, 492 MB of f32 output per 768 samples): the k loop is four tiles
// long, so in gemm_x3 a workgroup is mostly prologue (first loads at full latency) and epilogue (64 KB of stores), with two
// workgroups per CU to hide them behind.  Here the three weight planes of a 128-channel block stay in LDS for the whole kernel
// (3 x 128 x K bf16 = 96 KB), loaded once; eight waves per CU each walk their own 32-pixel tiles with NO barrier after that, the
// next tile's first operands in flight under the current tile's epilogue, so the waves drift apart and one wave's stores and
// loads sit under the others' MFMAs.
constexpr int X3R_K = 128, X3R_PITCH = X3R_K + 8;                       // weight rows: 272 B = 17 x 16 B (conflict-free b128 reads)
constexpr int X3R_PLANE = X3_BN * X3R_PITCH;
constexpr int X3R_LDS = 3 * X3R_PLANE * 2 + 8 * 32 * X3_EP * 4;         // 104 448 + 36 864 B

__global__ void __launch_bounds__(512, 2) gemm_x3_res(GemmDesc d, const float* __restrict__ X, const uint16_t* __restrict__ Wt, int Kpad, int64_t wplane,
                                                       const float* __restrict__ bias, float* __restrict__ Y, int ntiles) {
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  uint16_t* Ws = reinterpret_cast<uint16_t*>(gsm);                 // [plane 3][X3_BN][X3R_PITCH]
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int n0 = blockIdx.y * X3_BN;
  float* image = reinterpret_cast<float*>(gsm + 3 * X3R_PLANE * 2) + wave * (32 * X3_EP);
  // the block's weights: 3 planes x 128 rows x K / 8 chunks of 16 bytes
  const int cpr = d.K >> 3;
  for (int e = tid; e < 3 * X3_BN * cpr; e += 512) {
    const int p = e / (X3_BN * cpr), r = (e - p * X3_BN * cpr) / cpr, c = e - (p * X3_BN + r) * cpr;
    *reinterpret_cast<u32x4*>(Ws + p * X3R_PLANE + r * X3R_PITCH + c * 8) = *reinterpret_cast<const u32x4*>(Wt + (p * wplane + (int64_t)(n0 + r) * Kpad + c * 8));
  }
  __syncthreads();
  const uint16_t* wsr = Ws + l31 * X3R_PITCH + h * 8;
  const int per = d.MH * d.MW, nkt = d.K / X3_BK;

  // lane l31 decodes pixel l31 of the tile; the lanes that load / store row (lane >> 3) + 8 j fetch its offsets by shuffle
  auto decode = [&](int tile, int (&xo)[4], int (&orow)[4]) {
    const int m = tile * 32 + l31;
    int xoff = 0, o = -1;
    if (m < d.M) {
      const int img = m / per, r = m - img * per, my = r / d.MW, mx = r - my * d.MW;
      xoff = ((img * d.IH + my * d.ay) * d.IW + mx * d.ax) * d.CI;   // rows past M read row 0, nothing is stored
      o = ((img * d.OH + my * d.os + d.oy0) * d.OW + mx * d.os + d.ox0) * d.OC;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      xo[j] = __shfl(xoff, (lane >> 3) + 8 * j, 64) + 4 * (lane & 7);
      orow[j] = __shfl(o, (lane >> 3) + 8 * j, 64);
    }
  };
  f32x4 xr[4];
  auto g2r = [&](const int (&xo)[4], int k0) {
#pragma unroll
    for (int j = 0; j < 4; ++j) xr[j] = *reinterpret_cast<const f32x4*>(X + (xo[j] + k0));
  };

  int tile = (int)blockIdx.x * 8 + wave;
  const int tstep = (int)gridDim.x * 8;
  int xo[4], nxo[4] = {0, 0, 0, 0}, orow[4], norow[4] = {-1, -1, -1, -1};
  if (tile < ntiles) { decode(tile, xo, orow); g2r(xo, 0); }
  for (; tile < ntiles; tile += tstep) {
    f32x16 accH[4], accR[4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
      for (int r = 0; r < 16; ++r) { accH[a][r] = 0.f; accR[a][r] = 0.f; }
    const bool more_tiles = tile + tstep < ntiles;
    if (more_tiles) decode(tile + tstep, nxo, norow);
    for (int kt = 0; kt < nkt; ++kt) {
      u32x4 bh[2], bm[2], bl[2];
      x3_park_and_split(image, lane, xr, bh, bm, bl);
      if (kt + 1 < nkt) g2r(xo, (kt + 1) * X3_BK);
      else if (more_tiles) g2r(nxo, 0);            // the next tile's first operands ride under this tile's last MFMAs and its epilogue
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) {
#pragma unroll
        for (int a = 0; a < 4; ++a) {
          const uint16_t* wp = wsr + a * 32 * X3R_PITCH + kt * X3_BK + kk * 16;
          const u32x4 ah = *reinterpret_cast<const u32x4*>(wp), am = *reinterpret_cast<const u32x4*>(wp + X3R_PLANE),
                      al = *reinterpret_cast<const u32x4*>(wp + 2 * X3R_PLANE);
          f32x16 c = accR[a];
          c = mfma_bf(am, bm[kk], c);
          c = mfma_bf(ah, bl[kk], c);
          c = mfma_bf(al, bh[kk], c);
          c = mfma_bf(ah, bm[kk], c);
          c = mfma_bf(am, bh[kk], c);
          accR[a] = c;
          accH[a] = mfma_bf(ah, bh[kk], accH[a]);
        }
      }
    }
    x3_epilogue(d, accH, accR, bias, Y, image, n0, lane, orow);
#pragma unroll
    for (int j = 0; j < 4; ++j) { xo[j] = nxo[j]; orow[j] = norow[j]; }
  }
}

bool gemm_x3_res_qualifies(const GemmDesc& d) {
  return d.K <= X3R_K && d.K % X3_BK == 0 && d.TY * d.TX == 1 && d.cy == 0 && d.cx == 0;
}

// Qualifies: a GEMM the generic f32 kernel would run, with whole 32-deep k-tiles inside a tap and whole 128-channel blocks,
// channel groups of 32 inside one output phase, 16-byte aligned rows on both sides, swish or linear.
bool gemm_x3_qualifies(const GemmDesc& d) {
  return d.K >= 128 && d.CI % X3_BK == 0 && d.N % X3_BN == 0 && d.CO % 32 == 0 && d.OC % 4 == 0 && (d.act == SRCFD_ACT_SWISH || d.act == SRCFD_ACT_LINEAR) &&
         (d.M <= 0 || ((int64_t)(d.M / (d.MH * d.MW) + 1) * d.IH * d.IW * d.CI < (1ll << 31) &&     // 32-bit element offsets into X ...
                       (int64_t)(d.M / (d.MH * d.MW) + 1) * d.OH * d.OW * d.OC < (1ll << 31)));     // ... and into Y
}

int gemm_x3_kpad(const GemmDesc& d) { return (d.K + X3_BK - 1) / X3_BK * X3_BK; }

// B[K][Npad] f32 (the f32 engine's operand) -> Wt[plane 3][N][Kpad] bf16, exact three-way split by truncation
void gemm_x3_split_weights(const GemmDesc& d, const float* B, uint16_t* out) {
  const int Kpad = gemm_x3_kpad(d);
  const size_t plane = (size_t)d.N * Kpad;
  std::memset(out, 0, 3 * plane * sizeof(uint16_t));
  for (int k = 0; k < d.K; ++k)
    for (int n = 0; n < d.N; ++n) {
      const float w = B[(size_t)k * d.Npad + n];
      uint32_t b0, b1, b2;
      std::memcpy(&b0, &w, 4);
      b0 &= 0xffff0000u;
      float hi; std::memcpy(&hi, &b0, 4);
      const float r1 = w - hi;
      std::memcpy(&b1, &r1, 4);
      b1 &= 0xffff0000u;
      float mid; std::memcpy(&mid, &b1, 4);
      const float r2 = r1 - mid;
      std::memcpy(&b2, &r2, 4);
      const size_t o = (size_t)n * Kpad + k;
      out[o] = (uint16_t)(b0 >> 16); out[plane + o] = (uint16_t)(b1 >> 16); out[2 * plane + o] = (uint16_t)(b2 >> 16);
    }
}

hipError_t launch_gemm_x3(const GemmDesc& d, const float* X, const uint16_t* Wt, const float* bias, float* Y, hipStream_t s, int num_cus) {
  if (d.M <= 0) return hipSuccess;
  const int Kpad = gemm_x3_kpad(d);
  if (gemm_x3_res_qualifies(d)) {   // short K, no taps: weights resident in LDS, one workgroup per CU and channel block walks the pixel tiles
    hipError_t e = lds_attr_once(reinterpret_cast<const void*>(gemm_x3_res), X3R_LDS);
    if (e != hipSuccess) return e;
    const int ntiles = (d.M + 31) / 32, nblk = d.N / X3_BN;
    const int gx = std::max(1, std::min((ntiles + 7) / 8, std::max(1, num_cus / nblk)));
    hipLaunchKernelGGL(gemm_x3_res, dim3(gx, nblk), dim3(512), X3R_LDS, s, d, X, Wt, Kpad, (int64_t)d.N * Kpad, bias, Y, ntiles);
    return hipGetLastError();
  }
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(gemm_x3), X3_LDS);
  if (e != hipSuccess) return e;
  dim3 grid((unsigned)((d.M + X3_BP - 1) / X3_BP), (unsigned)(d.N / X3_BN));
  hipLaunchKernelGGL(gemm_x3, grid, dim3(256), X3_LDS, s, d, X, Wt, Kpad, (int64_t)d.N * Kpad, bias, Y);
  return hipGetLastError();
}

// ops of one layer: one launch when every op takes the tiled kernel (gemm_x3), else one launch per op
hipError_t launch_gemm_x3_group(const GemmDesc* ds, int count, const float* X, const uint16_t* const* Wts, const float* const* biases, float* Y,
                                hipStream_t s, int num_cus) {
  bool group = count >= 2 && count <= 4;
  for (int i = 0; group && i < count; ++i) group = ds[i].M > 0 && !gemm_x3_res_qualifies(ds[i]);
  if (!group) {
    for (int i = 0; i < count; ++i) {
      hipError_t e = launch_gemm_x3(ds[i], X, Wts[i], biases[i], Y, s, num_cus);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(gemm_x3_group), X3_LDS);
  if (e != hipSuccess) return e;
  X3Group g;
  g.count = count;
  int total = 0;
  for (int i = 0; i < count; ++i) {
    g.d[i] = ds[i]; g.Wt[i] = Wts[i]; g.bias[i] = biases[i];
    g.first[i] = total;
    total += ((ds[i].M + X3_BP - 1) / X3_BP) * (ds[i].N / X3_BN);
  }
  for (int i = count; i <= 4; ++i) g.first[i] = total;
  hipLaunchKernelGGL(gemm_x3_group, dim3((unsigned)total), dim3(256), X3_LDS, s, g, X, Y);
  return hipGetLastError();
}

}  // namespace srcfd
