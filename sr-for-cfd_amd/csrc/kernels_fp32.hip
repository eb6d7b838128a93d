// fp32 kernels of the SR engine (gfx950 only).
//
// One implicit-GEMM formulation covers every layer of encoder_10 / decoder_400
// (sr-ae-conv.ipynb:c162-169, c277-287):
//   out[row m, col n] = act( bias[n] + sum_k A[m,k] * B[k,n] )
//   m -> (image, my, mx) on a per-image row grid; k -> (ty, tx, ci)
//   A[m,k] = X[image, my*ay + ty*by + cy, mx*ax + tx*bx + cx, ci]   (0 outside)
//   n -> (phase, co); stored at (my*os + oy0 + py(phase), mx*os + ox0 + px(phase), co)
// Conv2D: ay=stride, by=1, cy=-pad_top.  Conv2DTranspose is split into its
// stride^2 output phases (ay=1, by=-1): no zero-inserted taps are multiplied.
// When kernel==stride the phases share one tap and are merged into a single
// GEMM with N = stride^2*Cout and a pixel-shuffle store.  Dense: 1x1 grid.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "act_device.h"
#include "kernels.h"

namespace srcfd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float act_apply(float v, int act) {
  switch (act) {
    case SRCFD_ACT_SWISH: {
      // x * sigmoid(x); v_exp_f32 + v_rcp_f32 (<= 2 ulp each)
      float e = __builtin_amdgcn_exp2f(v * -1.4426950408889634f);
      return v * __builtin_amdgcn_rcpf(1.0f + e);
    }
    case SRCFD_ACT_RELU: return fmaxf(v, 0.f);
    case SRCFD_ACT_SIGMOID: return 1.0f / (1.0f + __expf(-v));
    case SRCFD_ACT_TANH: return tanhf(v);
    default: return v;
  }
}

// The parity path's swish: x * rcp(1 + exp2(-log2(e) x)) on the hardware exp2 and reciprocal (<= 1-2 ulp each): 5 vector
// instructions per activation.  It moves the result by a few 1e-7 relative against libm expf + IEEE division (which cost
// ~35 instructions and were the largest single item of this path's time), an order of magnitude below the f32
// accumulation-order differences against the float64 oracle (tests: 1e-5 bar; measured 6e-7 on the full model).  A Newton
// step on the reciprocal (3 more instructions; the training kernels keep it, act_device.h) bought nothing measurable here,
// and on this chip vector instructions are not hidden behind the f32 MFMAs (tools/microbench8.hip: their times add).
// For x -> -inf: exp2 = inf, rcp = 0, x * 0 = -0 (no NaN); x -> +inf: exp2 = 0, x * 1.
__device__ __forceinline__ float act_apply_precise(float v, int act) {
  if (act == SRCFD_ACT_SWISH) return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f));
  return act_apply(v, act);
}

__device__ __forceinline__ void row_decode(const GemmDesc& d, int m, int& img, int& my, int& mx) {
  int per = d.MH * d.MW;
  img = m / per;
  int r = m - img * per;
  my = r / d.MW;
  mx = r - my * d.MW;
}

__device__ __forceinline__ int64_t out_offset(const GemmDesc& d, int img, int my, int mx, int n) {
  int ph = n / d.CO, co = n - ph * d.CO;
  int py = ph / d.nphx, px = ph - py * d.nphx;
  int oy = my * d.os + d.oy0 + py, ox = mx * d.os + d.ox0 + px;
  return (((int64_t)img * d.OH + oy) * d.OW + ox) * d.OC + co;
}

// ---------------------------------------------------------------------------
// bring-up kernel: one thread per output element, sequential f32 FMA chain in
// (ty,tx,ci) order -- the order oracle/sr_oracle.c uses.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) gemm_naive_f32(GemmDesc d, const float* __restrict__ X,
                                                       const float* __restrict__ B,
                                                       const float* __restrict__ bias,
                                                       float* __restrict__ Y) {
  int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  int64_t total = (int64_t)d.M * d.N;
  if (idx >= total) return;
  int m = (int)(idx / d.N), n = (int)(idx - (int64_t)m * d.N);
  int img, my, mx;
  row_decode(d, m, img, my, mx);
  float acc = bias[n];
  for (int ty = 0; ty < d.TY; ++ty) {
    int iy = my * d.ay + ty * d.by + d.cy;
    if (iy < 0 || iy >= d.IH) continue;
    for (int tx = 0; tx < d.TX; ++tx) {
      int ix = mx * d.ax + tx * d.bx + d.cx;
      if (ix < 0 || ix >= d.IW) continue;
      const float* xp = X + (((int64_t)img * d.IH + iy) * d.IW + ix) * d.CI;
      const float* bp = B + (int64_t)((ty * d.TX + tx) * d.CI) * d.Npad + n;
      for (int ci = 0; ci < d.CI; ++ci) acc = fmaf(xp[ci], bp[(int64_t)ci * d.Npad], acc);
    }
  }
  Y[out_offset(d, img, my, mx, n)] = act_apply_precise(acc, d.act);
}

// ---------------------------------------------------------------------------
// f32 MFMA implicit GEMM.  Block = 256 threads = 4 waves; block tile 128 x
// (32*NB); each wave owns 32 rows x NB tiles of 32 columns
// (v_mfma_f32_32x32x2_f32: exact f32 products, k-ordered fmaf chain).
// ---------------------------------------------------------------------------
constexpr int BM = 128;

// VEC 0: scalar gather (any CI); 1: CI % 4 == 0 (each 16-byte group of k sits inside one tap).
// BKT: depth of a K slab, 16 or 32.  One slab = one round trip to memory, so launches that cannot hide it behind other
// workgroups (a single field, a training micro-batch) take the deeper slab; full-batch launches run ~1 % faster with 16.
// The k order of the MFMA chain does not depend on it.
// The next slab's global loads are issued into registers before the current slab's MFMAs and
// parked in LDS after them, so HBM/L2 latency overlaps the matrix work (one LDS stage).
template <int NB, int VEC, int BKT>
__device__ __forceinline__ void gemm_mfma_tile(const GemmDesc& d, const float* __restrict__ X, const float* __restrict__ B,
                                               const float* __restrict__ bias, float* __restrict__ Y, float* __restrict__ ws,
                                               int kchunk, int bx, int by, int bz, const EpiAux& aux) {
  constexpr int BN = 32 * NB;
  constexpr int BK = BKT, LDA = BK + 1;
  __shared__ float As[BM * LDA];
  __shared__ float Bs[BK * BN];
  __shared__ int row_img[BM], row_my[BM], row_mx[BM];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int m0 = bx * BM, n0 = by * BN;

  if (tid < BM) {
    int m = m0 + tid, img = -1, my = 0, mx = 0;
    if (m < d.M) row_decode(d, m, img, my, mx);
    row_img[tid] = img; row_my[tid] = my; row_mx[tid] = mx;
  }
  __syncthreads();

  f32x16 acc[NB];
#pragma unroll
  for (int i = 0; i < NB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;

  // split-K (ws != nullptr): this block reduces k in [z*kchunk, (z+1)*kchunk) and leaves raw partial
  // sums in ws[z][m][n]; splitk_finish_f32 adds the slabs in z order (reproducible), bias and activation.
  const int K = ws ? min(d.K, (bz + 1) * kchunk) : d.K;
  const int kbeg = ws ? bz * kchunk : 0;

  constexpr int KG = BK / 4;            // 16-byte k groups per row (VEC)
  constexpr int RJ = BM * KG / 256;     // rows per thread (VEC)
  constexpr int AJ = BM * BK / 256;     // A elements per thread
  constexpr int BJ = BK * BN / 256;     // B elements per thread
  float ra[AJ], rb[BJ];
  auto fetch = [&](int k0) {
    if (VEC) {
      const int cg = tid % KG, k = k0 + 4 * cg;
      const int tap = k / d.CI, ci0 = k - tap * d.CI, ty = tap / d.TX, tx = tap - ty * d.TX;
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        const int row = tid / KG + (256 / KG) * j;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        const int img = row_img[row];
        if (img >= 0 && k < K) {
          int iy = row_my[row] * d.ay + ty * d.by + d.cy, ix = row_mx[row] * d.ax + tx * d.bx + d.cx;
          if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW)
            v = *reinterpret_cast<const float4*>(X + (((int64_t)img * d.IH + iy) * d.IW + ix) * d.CI + ci0);
        }
        ra[4 * j] = v.x; ra[4 * j + 1] = v.y; ra[4 * j + 2] = v.z; ra[4 * j + 3] = v.w;
      }
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) {
        int e = tid + 256 * j, row = e / BK, kk = e % BK, k = k0 + kk;
        float v = 0.f;
        int img = row_img[row];
        if (img >= 0 && k < K) {
          int tap = k / d.CI, ci = k - tap * d.CI;
          int ty = tap / d.TX, tx = tap - ty * d.TX;
          int iy = row_my[row] * d.ay + ty * d.by + d.cy, ix = row_mx[row] * d.ax + tx * d.bx + d.cx;
          if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW)
            v = X[(((int64_t)img * d.IH + iy) * d.IW + ix) * d.CI + ci];
        }
        ra[j] = v;
      }
    }
    // B tile [BK][BN] (B is zero-padded to Npad columns)
#pragma unroll
    for (int j = 0; j < BJ; ++j) {
      int e = tid + 256 * j, kk = e / BN, c = e - kk * BN, k = k0 + kk;
      rb[j] = (k < K) ? B[(int64_t)k * d.Npad + n0 + c] : 0.f;
    }
  };
  auto park = [&]() {
    if (VEC) {
#pragma unroll
      for (int j = 0; j < RJ; ++j) {
        float* dst = As + (tid / KG + (256 / KG) * j) * LDA + (tid % KG) * 4;
        dst[0] = ra[4 * j]; dst[1] = ra[4 * j + 1]; dst[2] = ra[4 * j + 2]; dst[3] = ra[4 * j + 3];
      }
    } else {
#pragma unroll
      for (int j = 0; j < AJ; ++j) { int e = tid + 256 * j; As[(e / BK) * LDA + (e % BK)] = ra[j]; }
    }
#pragma unroll
    for (int j = 0; j < BJ; ++j) Bs[tid + 256 * j] = rb[j];
  };

  if (kbeg < K) fetch(kbeg);
  for (int k0 = kbeg; k0 < K; k0 += BK) {
    park();
    __syncthreads();
    if (k0 + BK < K) fetch(k0 + BK);
    const float* ap = As + (wave * 32 + (lane & 31)) * LDA + (lane >> 5);
    const float* bp = Bs + (lane >> 5) * BN + (lane & 31);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float a = ap[kk];
#pragma unroll
      for (int i = 0; i < NB; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[kk * BN + 32 * i], acc[i], 0, 0, 0);
    }
    __syncthreads();
  }

  if (ws) {
    float* slab = ws + (int64_t)bz * d.M * d.Npad;
#pragma unroll
    for (int i = 0; i < NB; ++i) {
      int n = n0 + 32 * i + (lane & 31);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (m < d.M) slab[(int64_t)m * d.Npad + n] = acc[i][r];
      }
    }
    return;
  }
  // ---- epilogue: bias + activation + (pixel-shuffle) store ----
#pragma unroll
  for (int i = 0; i < NB; ++i) {
    int n = n0 + 32 * i + (lane & 31);
    if (n >= d.N) continue;
    float bv = bias[n];
    int ph = n / d.CO, co = n - ph * d.CO;
    int py = ph / d.nphx, px = ph - py * d.nphx;
    // the activation switch is taken once per tile, not once per element: with the switch inlined 16 x NB times the
    // epilogue was 70 KB of branchy code (NB = 4) that every launch walked through instruction-cache misses
    float v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = acc[i][r] + bv;
    if (aux.mode) {  // training: the element-wise pass that would follow this launch (kernels.h, EpiAux)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        int img = row_img[row];
        if (img < 0) continue;
        const float t = v[r];
        int oy = row_my[row] * d.os + d.oy0 + py, ox = row_mx[row] * d.os + d.ox0 + px;
        const int64_t off = (((int64_t)img * d.OH + oy) * d.OW + ox) * d.OC + co;
        if (aux.mode == 1) { Y[off] = t; aux.y2[off] = swish_train(t); }
        else Y[off] = t * swish_grad_train(aux.zaux[off]);
      }
      continue;
    }
    if (d.act == SRCFD_ACT_SWISH) {
#pragma unroll
      for (int r = 0; r < 16; ++r) v[r] = act_apply_precise(v[r], SRCFD_ACT_SWISH);
    } else if (d.act != SRCFD_ACT_LINEAR) {
#pragma unroll 1
      for (int r = 0; r < 16; ++r) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) t = (q == r) ? v[q] : t;
        t = act_apply(t, d.act);
#pragma unroll
        for (int q = 0; q < 16; ++q) v[q] = (q == r) ? t : v[q];
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      int row = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
      int img = row_img[row];
      if (img < 0) continue;
      int oy = row_my[row] * d.os + d.oy0 + py, ox = row_mx[row] * d.os + d.ox0 + px;
      Y[(((int64_t)img * d.OH + oy) * d.OW + ox) * d.OC + co] = v[r];
    }
  }
}

template <int NB, int VEC, int BKT>
__global__ void __launch_bounds__(256) gemm_mfma_f32(GemmDesc d, const float* __restrict__ X, const float* __restrict__ B,
                                                      const float* __restrict__ bias, float* __restrict__ Y, float* __restrict__ ws,
                                                      int kchunk, EpiAux aux) {
  gemm_mfma_tile<NB, VEC, BKT>(d, X, B, bias, Y, ws, kchunk, blockIdx.x, blockIdx.y, blockIdx.z, aux);
}

// The output phases of one transposed convolution (kernel != stride: up to stride^2 GEMMs that read the same input and write
// disjoint pixels of the same output) as ONE launch: blockIdx.z runs over (phase, K split).  A single field is a handful
// of row tiles per phase, so four launches + their split-K finishes were four round trips of latency (~85 us of the
// ~205 us one-field forward); results are those of the separate launches, bit for bit.
struct GemmGroup {
  GemmDesc d[4];
  const float* B[4];
  const float* bias[4];
  float* ws[4];      // split-K slabs of the phase, or nullptr
  int kchunk[4];
  int zend[4];       // exclusive end of the phase's blockIdx.z range
  int count;
};

template <int NB, int VEC, int BKT>
__global__ void __launch_bounds__(256) gemm_mfma_group_f32(GemmGroup g, const float* __restrict__ X, float* __restrict__ Y, EpiAux aux) {
  int p = 0;
  while (p + 1 < g.count && (int)blockIdx.z >= g.zend[p]) ++p;
  const int zbeg = p ? g.zend[p - 1] : 0;
  const GemmDesc d = g.d[p];
  if ((int)blockIdx.x * BM >= d.M) return;
  gemm_mfma_tile<NB, VEC, BKT>(d, X, g.B[p], g.bias[p], Y, g.ws[p], g.kchunk[p], blockIdx.x, blockIdx.y, (int)blockIdx.z - zbeg, aux);
}

__device__ __forceinline__ void splitk_finish_tile(const GemmDesc& d, const float* __restrict__ ws, int splits,
                                                   const float* __restrict__ bias, float* __restrict__ Y, int groups, int bx,
                                                   const EpiAux& aux) {
  // (256 / groups) outputs x `groups` slab groups per block: group g adds slabs g, g+groups, ... and the
  // groups are added in order, so the result does not depend on scheduling.
  __shared__ float red[256];
  const int epb = 256 / groups, e = threadIdx.x % epb, g = threadIdx.x / epb;
  const int64_t idx = (int64_t)bx * epb + e, total = (int64_t)d.M * d.N;
  int m = 0, n = 0;
  float acc = 0.f;
  if (idx < total) {
    m = (int)(idx / d.N); n = (int)(idx - (int64_t)m * d.N);
    for (int z = g; z < splits; z += groups) acc += ws[((int64_t)z * d.M + m) * d.Npad + n];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (g != 0 || idx >= total) return;
  for (int q = 1; q < groups; ++q) acc += red[q * epb + e];
  int img, my, mx;
  row_decode(d, m, img, my, mx);
  const int64_t off = out_offset(d, img, my, mx, n);
  const float v = acc + bias[n];
  if (aux.mode == 1) { Y[off] = v; aux.y2[off] = swish_train(v); }
  else if (aux.mode == 2) Y[off] = v * swish_grad_train(aux.zaux[off]);
  else Y[off] = act_apply_precise(v, d.act);
}

__global__ void __launch_bounds__(256) splitk_finish_f32(GemmDesc d, const float* __restrict__ ws, int splits,
                                                          const float* __restrict__ bias, float* __restrict__ Y, int groups, EpiAux aux) {
  splitk_finish_tile(d, ws, splits, bias, Y, groups, blockIdx.x, aux);
}

__global__ void __launch_bounds__(256) splitk_finish_group_f32(GemmGroup g, float* __restrict__ Y, EpiAux aux) {
  const int p = blockIdx.y;
  if (!g.ws[p]) return;  // the phase was not split: its GEMM epilogue already wrote Y
  const GemmDesc d = g.d[p];
  if ((int64_t)blockIdx.x * 256 >= (int64_t)d.M * d.N) return;
  splitk_finish_tile(d, g.ws[p], g.zend[p] - (p ? g.zend[p - 1] : 0), g.bias[p], Y, 1, blockIdx.x, aux);
}

// ---------------------------------------------------------------------------
// single-output-channel conv (decoder's 3x3 8->1 `output_image_400`): the
// GEMM tile would be 97 % padding at N=1, so one thread owns one output pixel and walks the
// taps with 16-byte loads; same (ty,tx,ci) f32 FMA order as the oracle.
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) conv_n1_f32(GemmDesc d, const float* __restrict__ X, const float* __restrict__ B,
                                                    const float* __restrict__ bias, float* __restrict__ Y) {
  __shared__ float w[512];
  for (int i = threadIdx.x; i < d.K; i += 256) w[i] = B[(int64_t)i * d.Npad];
  __syncthreads();
  int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= d.M) return;
  int img, my, mx;
  row_decode(d, (int)m, img, my, mx);
  float acc = bias[0];
  for (int ty = 0; ty < d.TY; ++ty) {
    int iy = my * d.ay + ty * d.by + d.cy;
    if (iy < 0 || iy >= d.IH) continue;
    for (int tx = 0; tx < d.TX; ++tx) {
      int ix = mx * d.ax + tx * d.bx + d.cx;
      if (ix < 0 || ix >= d.IW) continue;
      const float* xp = X + (((int64_t)img * d.IH + iy) * d.IW + ix) * d.CI;
      const float* wp = w + (ty * d.TX + tx) * d.CI;
      if ((d.CI & 3) == 0) {
        for (int ci = 0; ci < d.CI; ci += 4) {
          float4 v = *reinterpret_cast<const float4*>(xp + ci);
          acc = fmaf(v.x, wp[ci], acc); acc = fmaf(v.y, wp[ci + 1], acc);
          acc = fmaf(v.z, wp[ci + 2], acc); acc = fmaf(v.w, wp[ci + 3], acc);
        }
      } else {
        for (int ci = 0; ci < d.CI; ++ci) acc = fmaf(xp[ci], wp[ci], acc);
      }
    }
  }
  Y[out_offset(d, img, my, mx, 0)] = act_apply_precise(acc, d.act);
}

// Same layer, four horizontally adjacent outputs per thread: a 3-wide tap window over 4 pixels touches
// 6 input pixels per kernel row instead of 12, halving the L1 traffic that bounds conv_n1_f32.
// Needs unit stride, a 3x3 kernel, 8 input channels and rows that are a multiple of 4 wide.
__global__ void __launch_bounds__(256) conv_n1x4_f32(GemmDesc d, const float* __restrict__ X, const float* __restrict__ B,
                                                      const float* __restrict__ bias, float* __restrict__ Y) {
  __shared__ float w[512];
  for (int i = threadIdx.x; i < d.K; i += 256) w[i] = B[(int64_t)i * d.Npad];
  __syncthreads();
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx * 4 >= d.M) return;
  int img, my, mx;
  row_decode(d, (int)(idx * 4), img, my, mx);
  const float b0 = bias[0];
  float acc[4] = {b0, b0, b0, b0};
  // all 36 loads are issued before the first FMA: out-of-range taps read a clamped (valid) address and
  // are zeroed afterwards, so there is no branch between the loads and they overlap in flight
  float4 v[3][6][2];
#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
    const int iy = my + ty * d.by + d.cy;
    const bool yok = iy >= 0 && iy < d.IH;
    const float* row = X + ((int64_t)img * d.IH + min(max(iy, 0), d.IH - 1)) * d.IW * 8;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int ix = mx + d.cx + j;
      const bool ok = yok && ix >= 0 && ix < d.IW;
      const float* px = row + (int64_t)min(max(ix, 0), d.IW - 1) * 8;
      float4 a = *reinterpret_cast<const float4*>(px), c = *reinterpret_cast<const float4*>(px + 4);
      v[ty][j][0] = ok ? a : make_float4(0.f, 0.f, 0.f, 0.f);
      v[ty][j][1] = ok ? c : make_float4(0.f, 0.f, 0.f, 0.f);
    }
  }
#pragma unroll
  for (int ty = 0; ty < 3; ++ty) {
#pragma unroll
    for (int tx = 0; tx < 3; ++tx) {
      const float* wp = w + (ty * 3 + tx) * 8;
#pragma unroll
      for (int p = 0; p < 4; ++p) {
        const float4 a = v[ty][p + tx][0], c = v[ty][p + tx][1];
        float t = acc[p];
        t = fmaf(a.x, wp[0], t); t = fmaf(a.y, wp[1], t); t = fmaf(a.z, wp[2], t); t = fmaf(a.w, wp[3], t);
        t = fmaf(c.x, wp[4], t); t = fmaf(c.y, wp[5], t); t = fmaf(c.z, wp[6], t); t = fmaf(c.w, wp[7], t);
        acc[p] = t;
      }
    }
  }
  *reinterpret_cast<float4*>(Y + out_offset(d, img, my, mx, 0)) =
      make_float4(act_apply_precise(acc[0], d.act), act_apply_precise(acc[1], d.act), act_apply_precise(acc[2], d.act),
                  act_apply_precise(acc[3], d.act));
}

// Same layer again, LDS-tiled: the per-pixel gathers above touch one 128-byte line per lane and load
// instruction (the L1 tag rate, not HBM, bounds them at ~1.4 TB/s).  Here a workgroup stages an
// (16+2) x (64+2)-pixel input tile with fully coalesced 16-byte loads, then one thread per output pixel
// reads its 3x3x8 window from LDS (32-byte pixel pitch: 8 consecutive lanes cover all 64 banks).
// FIN != 0: the network's last layer -- de-standardise, NaN/Inf guard and the output cast (finalize_out below,
// same roundings) happen here instead of in a second pass over the image (FIN - 1 = output type: 0 f32, 1 bf16, 2 f16).
struct FinalEpilogue {
  void* out;
  const float* affine;  // per-sample (mean, std) or nullptr
  int nan_guard;
  unsigned long long* nonfinite;
};

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f);

constexpr int CT_H = 16, CT_W = 64, CT_PW = CT_W + 2;
template <int FIN>
__global__ void __launch_bounds__(256) conv_n1_tile_f32(GemmDesc d, const float* __restrict__ X, const float* __restrict__ B,
                                                         const float* __restrict__ bias, float* __restrict__ Y, FinalEpilogue fe) {
  __shared__ float4 tile[(CT_H + 2) * CT_PW * 2];
  __shared__ float w[72];
  const int tid = threadIdx.x;
  const int x0 = blockIdx.x * CT_W, y0 = blockIdx.y * CT_H, img = blockIdx.z;
  if (tid < 72) w[tid] = B[(int64_t)tid * d.Npad];
  // stage rows y0-1 .. y0+CT_H, pixels x0-1 .. x0+CT_W (zero outside the image = SAME padding)
  const float4* src = reinterpret_cast<const float4*>(X + (int64_t)img * d.IH * d.IW * 8);
  for (int e = tid; e < (CT_H + 2) * CT_PW * 2; e += 256) {
    const int r = e / (CT_PW * 2), c = e - r * (CT_PW * 2), px = c >> 1, hf = c & 1;
    const int iy = y0 - 1 + r, ix = x0 - 1 + px;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW) v = src[((int64_t)iy * d.IW + ix) * 2 + hf];
    tile[e] = v;
  }
  __syncthreads();
  const int lx = tid & (CT_W - 1), ly0 = tid / CT_W;  // 4 rows per pass
  const float b0 = bias[0];
  const bool in_x = x0 + lx < d.OW;
  float f_mean = 0.f, f_std = 1.f;
  if (FIN && fe.affine) { f_mean = fe.affine[2 * img]; f_std = fe.affine[2 * img + 1]; }
  unsigned nbad = 0;
#pragma unroll
  for (int pass = 0; pass < CT_H / 4; ++pass) {
    const int ly = ly0 + 4 * pass;
    if (!in_x || y0 + ly >= d.OH) continue;
    float acc = b0;
#pragma unroll
    for (int ty = 0; ty < 3; ++ty)
#pragma unroll
      for (int tx = 0; tx < 3; ++tx) {
        const float4 a = tile[((ly + ty) * CT_PW + lx + tx) * 2], c = tile[((ly + ty) * CT_PW + lx + tx) * 2 + 1];
        const float* wp = w + (ty * 3 + tx) * 8;
        acc = fmaf(a.x, wp[0], acc); acc = fmaf(a.y, wp[1], acc); acc = fmaf(a.z, wp[2], acc); acc = fmaf(a.w, wp[3], acc);
        acc = fmaf(c.x, wp[4], acc); acc = fmaf(c.y, wp[5], acc); acc = fmaf(c.z, wp[6], acc); acc = fmaf(c.w, wp[7], acc);
      }
    float v = act_apply_precise(acc, d.act);
    const int64_t o = ((int64_t)img * d.OH + y0 + ly) * d.OW + x0 + lx;
    if (!FIN) { Y[o] = v; continue; }
    if (fe.affine) v = __fadd_rn(__fmul_rn(v, f_std), f_mean);
    if (fe.nan_guard && !(fabsf(v) <= 3.402823466e38f)) { ++nbad; v = 0.f; }
    if (FIN == 1) reinterpret_cast<float*>(fe.out)[o] = v;
    else if (FIN == 2) reinterpret_cast<unsigned short*>(fe.out)[o] = f32_to_bf16_bits(v);
    else reinterpret_cast<_Float16*>(fe.out)[o] = (_Float16)v;
  }
  if (FIN && fe.nan_guard && fe.nonfinite) {  // every thread of the block reaches this point
    for (int o = 32; o > 0; o >>= 1) nbad += __shfl_xor(nbad, o, 64);
    if ((tid & 63) == 0 && nbad) atomicAdd(fe.nonfinite, (unsigned long long)nbad);
  }
}

// single-input-channel conv with <= 8 output channels (the data gradient of `output_image_400`:
// a 3x3 1->8 flipped-tap conv over 400x400): one thread per pixel, 8 accumulators, two 16-byte stores.
__global__ void __launch_bounds__(256) conv_ci1_f32(GemmDesc d, const float* __restrict__ X, const float* __restrict__ B,
                                                     const float* __restrict__ bias, float* __restrict__ Y, EpiAux aux) {
  __shared__ float w[64 * 8];
  for (int i = threadIdx.x; i < d.K * 8; i += 256) { int k = i >> 3, c = i & 7; w[i] = c < d.N ? B[(int64_t)k * d.Npad + c] : 0.f; }
  __syncthreads();
  int64_t m = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (m >= d.M) return;
  int img, my, mx;
  row_decode(d, (int)m, img, my, mx);
  float acc[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) acc[c] = c < d.N ? bias[c] : 0.f;
  for (int ty = 0; ty < d.TY; ++ty) {
    const int iy = my * d.ay + ty * d.by + d.cy;
    const bool yok = iy >= 0 && iy < d.IH;
    const float* row = X + ((int64_t)img * d.IH + min(max(iy, 0), d.IH - 1)) * d.IW;
    for (int tx = 0; tx < d.TX; ++tx) {
      const int ix = mx * d.ax + tx * d.bx + d.cx;
      float v = row[min(max(ix, 0), d.IW - 1)];       // clamped address, zeroed below: no branch around the load
      v = (yok && ix >= 0 && ix < d.IW) ? v : 0.f;
      const float* wp = w + (ty * d.TX + tx) * 8;
#pragma unroll
      for (int c = 0; c < 8; ++c) acc[c] = fmaf(v, wp[c], acc[c]);
    }
  }
  const int64_t off0 = out_offset(d, img, my, mx, 0);
  float* yp = Y + off0;
  if (aux.mode && d.N == 8 && (d.OC & 3) == 0) {  // training epilogues (kernels.h, EpiAux); the 8 channels of a pixel are 32 contiguous bytes: two 16-byte accesses per array instead of eight 4-byte ones 32 B apart across lanes
    float r[8];
    if (aux.mode == 1) {
#pragma unroll
      for (int c = 0; c < 8; ++c) r[c] = swish_train(acc[c]);
      reinterpret_cast<float4*>(yp)[0] = make_float4(acc[0], acc[1], acc[2], acc[3]);
      reinterpret_cast<float4*>(yp)[1] = make_float4(acc[4], acc[5], acc[6], acc[7]);
      reinterpret_cast<float4*>(aux.y2 + off0)[0] = make_float4(r[0], r[1], r[2], r[3]);
      reinterpret_cast<float4*>(aux.y2 + off0)[1] = make_float4(r[4], r[5], r[6], r[7]);
    } else {
      const float4 z0 = reinterpret_cast<const float4*>(aux.zaux + off0)[0], z1 = reinterpret_cast<const float4*>(aux.zaux + off0)[1];
      const float z[8] = {z0.x, z0.y, z0.z, z0.w, z1.x, z1.y, z1.z, z1.w};
#pragma unroll
      for (int c = 0; c < 8; ++c) r[c] = acc[c] * swish_grad_train(z[c]);
      reinterpret_cast<float4*>(yp)[0] = make_float4(r[0], r[1], r[2], r[3]);
      reinterpret_cast<float4*>(yp)[1] = make_float4(r[4], r[5], r[6], r[7]);
    }
    return;
  }
  if (aux.mode) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      if (c >= d.N) continue;
      if (aux.mode == 1) { yp[c] = acc[c]; aux.y2[off0 + c] = swish_train(acc[c]); }
      else yp[c] = acc[c] * swish_grad_train(aux.zaux[off0 + c]);
    }
    return;
  }
  if (d.N == 8 && (d.OC & 3) == 0) {
    reinterpret_cast<float4*>(yp)[0] = make_float4(act_apply_precise(acc[0], d.act), act_apply_precise(acc[1], d.act),
                                                    act_apply_precise(acc[2], d.act), act_apply_precise(acc[3], d.act));
    reinterpret_cast<float4*>(yp)[1] = make_float4(act_apply_precise(acc[4], d.act), act_apply_precise(acc[5], d.act),
                                                    act_apply_precise(acc[6], d.act), act_apply_precise(acc[7], d.act));
  } else {
#pragma unroll
    for (int c = 0; c < 8; ++c) if (c < d.N) yp[c] = act_apply_precise(acc[c], d.act);
  }
}

// ---------------------------------------------------------------------------
// element-wise pre / post
// ---------------------------------------------------------------------------
// x := (x - mean) / std in float32, exactly numpy's float32 arithmetic
// (standardize_with_stats, PyCFD_ML_accelerated.py:665-668).
__global__ void __launch_bounds__(256) standardize_f32(const float* __restrict__ x, float* __restrict__ y,
                                                        const float* __restrict__ affine, int per_sample, int64_t total) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  float v = x[i];
  if (affine) {
    int s = (int)(i / per_sample);
    float mean = affine[2 * s], sd = affine[2 * s + 1];
    if (sd == 0.f) sd = 1e-8f;
    v = __fdiv_rn(__fsub_rn(v, mean), sd);
  }
  y[i] = v;
}

__device__ __forceinline__ unsigned short f32_to_bf16_bits(float f) {
  __bf16 h = (__bf16)f;  // v_cvt_pk_bf16_f32: RNE, NaN stays NaN
  return __builtin_bit_cast(unsigned short, h);
}

// y := y * std + mean (two roundings, inverse_standardize PyCFD...:671-673), then
// the NaN/Inf zero-fill of PyCFD...:869-876, then the output cast.
template <int OUT>  // 0 f32, 1 bf16, 2 f16
__global__ void __launch_bounds__(256) finalize_out(const float* __restrict__ y, void* __restrict__ out,
                                                     const float* __restrict__ affine, int per_sample, int64_t total,
                                                     int nan_guard, unsigned long long* __restrict__ nonfinite) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  bool bad = false;
  if (i < total) {
    float v = y[i];
    if (affine) {
      int s = (int)(i / per_sample);
      v = __fadd_rn(__fmul_rn(v, affine[2 * s + 1]), affine[2 * s]);
    }
    if (nan_guard && !(fabsf(v) <= 3.402823466e38f)) { bad = true; v = 0.f; }
    if (OUT == 0) ((float*)out)[i] = v;
    else if (OUT == 1) ((unsigned short*)out)[i] = f32_to_bf16_bits(v);
    else ((_Float16*)out)[i] = (_Float16)v;
  }
  if (nan_guard && nonfinite) {
    unsigned long long mask = __ballot(bad);
    if (mask && (threadIdx.x & 63) == 0) atomicAdd(nonfinite, (unsigned long long)__popcll(mask));
  }
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_gemm_naive(const GemmDesc& d, const float* X, const float* B, const float* bias, float* Y, hipStream_t s) {
  int64_t total = (int64_t)d.M * d.N;
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(gemm_naive_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, d, X, B, bias, Y);
  return hipGetLastError();
}

// Skinny problems (few output tiles, long K: the Dense layers, and every layer at training batch
// sizes) are cut along K so that the weight matrix is streamed by >= 256 blocks instead of 1-6.
int gemm_splitk_splits(const GemmDesc& d, int* kchunk_out, bool batch_invariant) {
  if (kchunk_out) *kchunk_out = d.K;
  if (batch_invariant) {
    // inference: the summation order of a sample must not depend on the batch it sits in, so the cut depends on the
    // layer only: Dense layers (one row per sample) with a long K are always cut into 256-deep slabs, small conv maps into 144-deep ones
    if (d.M == 0) return 1;
    if (d.MH * d.MW == 1 && d.K >= 1024) {
      if (kchunk_out) *kchunk_out = 256;
      return (d.K + 255) / 256;
    }
    if (d.MH * d.MW <= 64 && d.K >= 512) {  // small feature maps with a long K (encoder conv2d_1: 25 pixels, K = 576): too few row tiles
      if (kchunk_out) *kchunk_out = 144;
      return (d.K + 143) / 144;
    }
    if (d.MH * d.MW <= 256 && d.K >= 512) {  // ConvT#0's phases (144-169 pixels per sample, K up to 1024): a single field is 4 row tiles
      if (kchunk_out) *kchunk_out = 256;     // -- costs ~3 % at batch 256 (slab traffic), cuts the one-field call by a third
      return (d.K + 255) / 256;
    }
    return 1;
  }
  int nb = (d.Npad % 128 == 0) ? 4 : ((d.Npad % 64 == 0) ? 2 : 1);
  int64_t tiles = (int64_t)((d.M + BM - 1) / BM) * (d.Npad / (32 * nb));
  if (tiles >= 128 || d.K < 256 || tiles == 0) return 1;
  int want = (int)std::min<int64_t>(std::min<int64_t>((512 + tiles - 1) / tiles, d.K / 64), 256);
  if (want <= 1) return 1;
  int kchunk = ((d.K + want - 1) / want + 31) / 32 * 32;
  if (kchunk_out) *kchunk_out = kchunk;
  return (d.K + kchunk - 1) / kchunk;
}

size_t gemm_splitk_ws_floats(const GemmDesc& d, bool batch_invariant) {
  int splits = gemm_splitk_splits(d, nullptr, batch_invariant);
  return splits > 1 ? (size_t)splits * d.M * d.Npad : 0;
}

static bool is_tiled_n1_conv(const GemmDesc& d) {
  return d.N == 1 && d.nphx == 1 && d.TX == 3 && d.TY == 3 && d.CI == 8 && d.ax == 1 && d.bx == 1 && d.ay == 1 && d.by == 1 && d.cx == -1 &&
         d.cy == -1 && d.os == 1 && d.ox0 == 0 && d.oy0 == 0 && d.OC == 1 && d.IH == d.OH && d.IW == d.OW && d.MH == d.OH && d.MW == d.OW &&
         d.M > 0 && d.M % (d.MH * d.MW) == 0;
}

bool gemm_fuses_finalize(const GemmDesc& d) { return is_tiled_n1_conv(d); }

hipError_t launch_gemm_finalize(const GemmDesc& d, const float* X, const float* B, const float* bias, void* out, int out_dtype,
                                const float* affine, int nan_guard, unsigned long long* nonfinite, hipStream_t s) {
  if (!is_tiled_n1_conv(d)) return hipErrorInvalidValue;
  dim3 grid((d.OW + CT_W - 1) / CT_W, (d.OH + CT_H - 1) / CT_H, d.M / (d.MH * d.MW));
  FinalEpilogue fe{out, affine, nan_guard, nonfinite};
  if (out_dtype == SRCFD_F32) hipLaunchKernelGGL(conv_n1_tile_f32<1>, grid, dim3(256), 0, s, d, X, B, bias, nullptr, fe);
  else if (out_dtype == SRCFD_BF16) hipLaunchKernelGGL(conv_n1_tile_f32<2>, grid, dim3(256), 0, s, d, X, B, bias, nullptr, fe);
  else if (out_dtype == SRCFD_F16) hipLaunchKernelGGL(conv_n1_tile_f32<3>, grid, dim3(256), 0, s, d, X, B, bias, nullptr, fe);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

static bool is_ci1_conv(const GemmDesc& d) { return d.CI == 1 && d.N <= 8 && d.K <= 64 && d.nphx == 1 && d.CO == d.N; }

hipError_t launch_gemm_mfma(const GemmDesc& d, const float* X, const float* B, const float* bias, float* Y, hipStream_t s, float* ws,
                            size_t ws_floats, bool batch_invariant, EpiAux aux) {
  if (d.M == 0 || d.N == 0) return hipSuccess;
  if (aux.mode && !gemm_supports_epi_aux(d)) return hipErrorInvalidValue;
  if (is_tiled_n1_conv(d)) {
    dim3 grid((d.OW + CT_W - 1) / CT_W, (d.OH + CT_H - 1) / CT_H, d.M / (d.MH * d.MW));
    hipLaunchKernelGGL(conv_n1_tile_f32<0>, grid, dim3(256), 0, s, d, X, B, bias, Y, FinalEpilogue{});
    return hipGetLastError();
  }
  if (d.N == 1 && d.nphx == 1 && d.TX == 3 && d.TY == 3 && d.CI == 8 && d.ax == 1 && d.bx == 1 && d.ay == 1 && d.os == 1 && d.ox0 == 0 && d.OC == 1 &&
      d.MW % 4 == 0 && d.OW % 4 == 0 && d.K <= 512) {
    hipLaunchKernelGGL(conv_n1x4_f32, dim3((unsigned)((d.M / 4 + 255) / 256)), dim3(256), 0, s, d, X, B, bias, Y);
    return hipGetLastError();
  }
  if (d.N == 1 && d.K <= 512 && d.nphx == 1) {
    hipLaunchKernelGGL(conv_n1_f32, dim3((unsigned)((d.M + 255) / 256)), dim3(256), 0, s, d, X, B, bias, Y);
    return hipGetLastError();
  }
  if (d.CI == 1 && d.N <= 8 && d.K <= 64 && d.nphx == 1 && d.CO == d.N) {
    hipLaunchKernelGGL(conv_ci1_f32, dim3((unsigned)((d.M + 255) / 256)), dim3(256), 0, s, d, X, B, bias, Y, aux);
    return hipGetLastError();
  }
  const int vec = (d.K > 0 && d.CI % 4 == 0) ? 1 : 0;
  int nb = (d.Npad % 128 == 0) ? 4 : ((d.Npad % 64 == 0) ? 2 : 1);
  int kchunk = d.K;
  int splits = ws ? gemm_splitk_splits(d, &kchunk, batch_invariant) : 1;
  if (splits > 1 && (size_t)splits * d.M * d.Npad > ws_floats) splits = 1;
  float* wsp = splits > 1 ? ws : nullptr;
  // few rows (a solver-side call is 3 samples): narrower column blocks per wave until the launch covers the chip, and
  // deeper K slabs.  The k order of every output element is the same for any nb / slab depth, so results stay
  // bit-identical across batch sizes.
  const int64_t row_tiles = (d.M + BM - 1) / BM;
  while (nb > 1 && row_tiles * (d.Npad / (32 * nb)) * splits < 512) nb >>= 1;
  const bool deep = row_tiles * (d.Npad / (32 * nb)) * splits < 1024;
  dim3 grid((unsigned)row_tiles, d.Npad / (32 * nb), splits);
#define GO2(NBV, VECV) do { if (deep) hipLaunchKernelGGL((gemm_mfma_f32<NBV, VECV, 32>), grid, dim3(256), 0, s, d, X, B, bias, Y, wsp, kchunk, aux); \
                            else hipLaunchKernelGGL((gemm_mfma_f32<NBV, VECV, 16>), grid, dim3(256), 0, s, d, X, B, bias, Y, wsp, kchunk, aux); } while (0)
#define GO(NBV) do { if (vec) GO2(NBV, 1); else GO2(NBV, 0); } while (0)
  if (nb == 4) GO(4); else if (nb == 2) GO(2); else GO(1);
#undef GO2
#undef GO
  if (splits > 1) {
    int64_t total = (int64_t)d.M * d.N;
    const int groups = (splits >= 32 && total < 65536) ? 8 : 1, epb = 256 / groups;
    hipLaunchKernelGGL(splitk_finish_f32, dim3((unsigned)((total + epb - 1) / epb)), dim3(256), 0, s, d, wsp, splits, bias, Y, groups, aux);
  }
  return hipGetLastError();
}

static bool uses_generic_gemm(const GemmDesc& d) {
  if (d.M <= 0 || d.N <= 0 || d.K <= 0) return false;
  if (is_tiled_n1_conv(d)) return false;
  if (d.N == 1 && d.K <= 512 && d.nphx == 1) return false;
  if (d.CI == 1 && d.N <= 8 && d.K <= 64 && d.nphx == 1 && d.CO == d.N) return false;
  return true;
}

size_t gemm_group_ws_floats(const GemmDesc* ds, int count, bool batch_invariant) {
  size_t total = 0;
  for (int i = 0; i < count; ++i) total += gemm_splitk_ws_floats(ds[i], batch_invariant);
  return total;
}

bool gemm_supports_epi_aux(const GemmDesc& d) { return uses_generic_gemm(d) || (is_ci1_conv(d) && d.M > 0 && d.K > 0); }

hipError_t launch_gemm_mfma_group(const GemmDesc* ds, int count, const float* X, const float* const* Bs, const float* const* biases, float* Y,
                                  hipStream_t s, float* ws, size_t ws_floats, bool batch_invariant, EpiAux aux) {
  bool group = count >= 2 && count <= 4;
  for (int i = 0; group && i < count; ++i)
    group = uses_generic_gemm(ds[i]) && ds[i].Npad == ds[0].Npad && (ds[i].CI % 4 == 0) == (ds[0].CI % 4 == 0);
  GemmGroup g{};
  int64_t tiles = 0, max_rows = 0, max_finish = 0;
  size_t ws_used = 0;
  bool any_split = false;
  for (int i = 0; group && i < count; ++i) {
    const GemmDesc& d = ds[i];
    int kchunk = d.K;
    int splits = ws ? gemm_splitk_splits(d, &kchunk, batch_invariant) : 1;
    const size_t need = splits > 1 ? (size_t)splits * d.M * d.Npad : 0;
    if (ws_used + need > ws_floats) { group = false; break; }
    if (splits >= 32 && (int64_t)d.M * d.N < 65536) { group = false; break; }  // the single launch would finish in 8 groups
    g.d[i] = d; g.B[i] = Bs[i]; g.bias[i] = biases[i];
    g.ws[i] = splits > 1 ? ws + ws_used : nullptr;
    g.kchunk[i] = kchunk;
    g.zend[i] = (i ? g.zend[i - 1] : 0) + splits;
    ws_used += need;
    const int64_t rows = (d.M + BM - 1) / BM;
    tiles += rows * splits;
    max_rows = std::max(max_rows, rows);
    if (splits > 1) { any_split = true; max_finish = std::max<int64_t>(max_finish, ((int64_t)d.M * d.N + 255) / 256); }
  }
  if (!group) {
    for (int i = 0; i < count; ++i) {
      hipError_t e = launch_gemm_mfma(ds[i], X, Bs[i], biases[i], Y, s, ws, ws_floats, batch_invariant, aux);
      if (e != hipSuccess) return e;
    }
    return hipSuccess;
  }
  g.count = count;
  const int vec = ds[0].CI % 4 == 0 ? 1 : 0, Npad = ds[0].Npad;
  int nb = (Npad % 128 == 0) ? 4 : ((Npad % 64 == 0) ? 2 : 1);
  while (nb > 1 && tiles * (Npad / (32 * nb)) < 512) nb >>= 1;
  const bool deep = tiles * (Npad / (32 * nb)) < 1024;
  dim3 grid((unsigned)max_rows, Npad / (32 * nb), g.zend[count - 1]);
#define GO2(NBV, VECV) do { if (deep) hipLaunchKernelGGL((gemm_mfma_group_f32<NBV, VECV, 32>), grid, dim3(256), 0, s, g, X, Y, aux); \
                            else hipLaunchKernelGGL((gemm_mfma_group_f32<NBV, VECV, 16>), grid, dim3(256), 0, s, g, X, Y, aux); } while (0)
#define GO(NBV) do { if (vec) GO2(NBV, 1); else GO2(NBV, 0); } while (0)
  if (nb == 4) GO(4); else if (nb == 2) GO(2); else GO(1);
#undef GO2
#undef GO
  if (any_split) hipLaunchKernelGGL(splitk_finish_group_f32, dim3((unsigned)max_finish, count), dim3(256), 0, s, g, Y, aux);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// ConvT (2x2, s2, 32 -> 16) chained into ConvT (2x2, s2, 16 -> 8), f32 (kernels.h, PairDesc).
// Transposed formulation: D = W^T (rows = tap x channel) * X^T (columns = 32 pixels of the wave), so the swished
// accumulators of the first GEMM -- lane (pixel, h) holds rows (r&3) + 8(r>>2) + 4h -- ARE the B operands of the second
// one (k-step u contracts channels (u&3) + 8(u>>2) + 4h; the weights are packed in that order on the host).  All 40
// A-operand registers (both layers' weights) stay resident while the wave walks its pixel groups.
// ---------------------------------------------------------------------------
constexpr int PAIR_PITCH = 144;  // bytes per staged pixel row (128 + 16: conflict-free 16-byte accesses)

__global__ void __launch_bounds__(256) convt_pair_f32(PairDesc d, const float* __restrict__ X, const float* __restrict__ wa,
                                                       const float* __restrict__ ba, const float* __restrict__ wb,
                                                       const float* __restrict__ bb, float* __restrict__ Y) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  float wA[2][16], wB[8], bA[8], bB[4];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int s = 0; s < 16; ++s) wA[T][s] = wa[(T * 16 + s) * 64 + lane];
#pragma unroll
  for (int u = 0; u < 8; ++u) wB[u] = wb[u * 64 + lane];
#pragma unroll
  for (int r = 0; r < 8; ++r) bA[r] = ba[(r & 3) + 4 * h + 8 * (r >> 2)];
#pragma unroll
  for (int r = 0; r < 4; ++r) bB[r] = bb[r + 4 * h];
  const int64_t M = (int64_t)d.n * d.H * d.W, groups = (M + 31) / 32;
  const int OH = 4 * d.H, OW = 4 * d.W;
  __shared__ __attribute__((aligned(16))) char pair_stage[4][2 * 32 * PAIR_PITCH];
  char* stage = pair_stage[wave];
  for (int64_t g = (int64_t)blockIdx.x * 4 + wave; g < groups; g += (int64_t)gridDim.x * 4) {
    const int64_t m = g * 32 + l31;
    const bool ok = m < M;
    const int64_t mm = ok ? m : M - 1;
    const float4* xp = reinterpret_cast<const float4*>(X + mm * 32 + 16 * h);
    const float4 x0 = xp[0], x1 = xp[1], x2 = xp[2], x3 = xp[3];
    const float xs[16] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w, x2.x, x2.y, x2.z, x2.w, x3.x, x3.y, x3.z, x3.w};
    const int img = (int)(mm / ((int64_t)d.H * d.W));
    const int rem = (int)(mm - (int64_t)img * d.H * d.W), y = rem / d.W, x = rem - y * d.W;
    const int64_t obase = ok ? (((int64_t)img * OH + 4 * y) * OW + 4 * x) * 8 : -1;
    f32x16 accA[2];
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[T][r] = bA[(r & 3) + 4 * ((r >> 2) & 1)];
#pragma unroll
    for (int s = 0; s < 16; ++s)
#pragma unroll
      for (int T = 0; T < 2; ++T) accA[T] = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[T][s], xs[s], accA[T], 0, 0, 0);
#pragma unroll
    for (int T = 0; T < 2; ++T)
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[T][r] = act_apply_precise(accA[T][r], d.act_a);
    // Output rows leave through a wave-private LDS tile: for one (ty_a, ty_b) output row, an input pixel owns 4 output pixels
    // x 32 B = 128 contiguous bytes, produced by this lane pair as 8 chunks of 16 B spread over two taps and four
    // registers.  Written as [pixel][chunk] (144-byte pitch), read back with 8 lanes per pixel, the 32 pixels of the wave
    // store 4 KB of whole 128-byte lines instead of 32-byte pieces 128 B apart.
#pragma unroll
    for (int tya = 0; tya < 2; ++tya) {
#pragma unroll
      for (int txa = 0; txa < 2; ++txa) {
        const int ta = 2 * tya + txa, T = ta >> 1, half = ta & 1;
        f32x16 accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = bB[r & 3];
#pragma unroll
        for (int u = 0; u < 8; ++u) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wB[u], accA[T][8 * half + u], accB, 0, 0, 0);
        // rows (r&3) + 8(r>>2) + 4h: second-layer tap tb = r>>2 = (ty_b, tx_b), channels 4h .. 4h+3
#pragma unroll
        for (int tb = 0; tb < 4; ++tb) {
          const float4 v = make_float4(act_apply_precise(accB[4 * tb], d.act_b), act_apply_precise(accB[4 * tb + 1], d.act_b),
                                       act_apply_precise(accB[4 * tb + 2], d.act_b), act_apply_precise(accB[4 * tb + 3], d.act_b));
          *reinterpret_cast<float4*>(stage + (tb >> 1) * (32 * PAIR_PITCH) + l31 * PAIR_PITCH + (4 * txa + 2 * (tb & 1) + h) * 16) = v;
        }
      }
      // both taps of this ty_a are staged: rows ty_b = 0, 1.  Lane -> (pixel = lane/8 + 8q, chunk = lane%8)
#pragma unroll
      for (int tyb = 0; tyb < 2; ++tyb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int pix = (lane >> 3) + 8 * q;
          const float4 v = *reinterpret_cast<const float4*>(stage + tyb * (32 * PAIR_PITCH) + pix * PAIR_PITCH + (lane & 7) * 16);
          const int64_t base = __shfl(obase, pix, 64);   // element offset of output pixel (4y, 4x) of that lane's input pixel
          if (base >= 0) *reinterpret_cast<float4*>(Y + base + ((int64_t)(2 * tya + tyb) * OW) * 8 + (lane & 7) * 4) = v;
        }
    }
  }
}

hipError_t launch_convt_pair_f32(const PairDesc& d, const float* X, const float* wa, const float* ba, const float* wb, const float* bb,
                                 float* Y, hipStream_t s) {
  const int64_t M = (int64_t)d.n * d.H * d.W;
  if (M == 0) return hipSuccess;
  const int64_t groups = (M + 31) / 32;
  const int blocks = (int)std::min<int64_t>((groups + 3) / 4, 256 * 4);  // <= 4 resident workgroups per CU, each wave loops
  hipLaunchKernelGGL(convt_pair_f32, dim3(blocks), dim3(256), 0, s, d, X, wa, ba, wb, bb, Y);
  return hipGetLastError();
}

// ---------------------------------------------------------------------------
// ConvT (64 -> 32) -> ConvT (32 -> 16) -> ConvT (16 -> 8), all 2x2 stride 2, f32: convt_pair_f32 with one more level.
// A tile of the first GEMM is exactly one tap (32 channels = 32 rows), so accA[tap1] feeds the second GEMM whole
// (k-step u contracts channels (u&3) + 8(u>>2) + 4h); the first GEMM is done one tap and the second one 32-row tile (= ty2)
// at a time, so that 16 accumulators of each are live (all 64 first-layer accumulators at once spilled to scratch and
// left one wave per SIMD: 0.73 ms per 256 samples).  A workgroup takes one 32-pixel group and its four waves one first-layer
// tap each (the taps are independent all the way down), so a wave holds one tile of the first layer's weights (32 registers)
// next to the other two layers' 40, and a single field (235 groups) still spreads over 940 waves.  Each input pixel
// becomes an 8 x 8 output block; rows leave through the same wave-private staging tile as in the pair kernel (128-byte segments: 4 output pixels x 32 B).
// ---------------------------------------------------------------------------
__global__ void __launch_bounds__(256) convt_triple_f32(TripleDesc d, const float* __restrict__ X, const float* __restrict__ w1,
                                                         const float* __restrict__ b1, const float* __restrict__ w2,
                                                         const float* __restrict__ b2, const float* __restrict__ w3,
                                                         const float* __restrict__ b3, float* __restrict__ Y) {
  __shared__ __attribute__((aligned(16))) char tri_stage[4][2 * 32 * PAIR_PITCH];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, l31 = lane & 31;
  // work item = (pixel group, first-layer tap) with tap == wave index (below), so a wave needs one tile of w1 only: 32 registers
  float wA[32], wB[2][16], wC[8], bA[16], bB[8], bC[4];
#pragma unroll
  for (int s = 0; s < 32; ++s) wA[s] = w1[((size_t)wave * 32 + s) * 64 + lane];
#pragma unroll
  for (int T = 0; T < 2; ++T)
#pragma unroll
    for (int u = 0; u < 16; ++u) wB[T][u] = w2[(T * 16 + u) * 64 + lane];
#pragma unroll
  for (int v = 0; v < 8; ++v) wC[v] = w3[v * 64 + lane];
#pragma unroll
  for (int r = 0; r < 16; ++r) bA[r] = b1[(r & 3) + 8 * (r >> 2) + 4 * h];
#pragma unroll
  for (int r = 0; r < 8; ++r) bB[r] = b2[(r & 3) + 4 * h + 8 * (r >> 2)];
#pragma unroll
  for (int r = 0; r < 4; ++r) bC[r] = b3[r + 4 * h];
  char* stage = tri_stage[wave];
  const int64_t M = (int64_t)d.n * d.H * d.W, groups = (M + 31) / 32;
  const int OH = 8 * d.H, OW = 8 * d.W;
  // one work item = (32-pixel group, first-layer tap): the four taps of a group are independent all the way down, so they go
  // to different waves -- a quarter of the serial chain per item (a single field is only 235 groups), input re-read from L2
  const int tap1 = wave;
  for (int64_t g = blockIdx.x; g < groups; g += gridDim.x) {
    const int64_t m = g * 32 + l31;
    const bool ok = m < M;
    const int64_t mm = ok ? m : M - 1;
    const float4* xp = reinterpret_cast<const float4*>(X + mm * 64 + 32 * h);
    float xs[32];
#pragma unroll
    for (int q = 0; q < 8; ++q) { const float4 t = xp[q]; xs[4 * q] = t.x; xs[4 * q + 1] = t.y; xs[4 * q + 2] = t.z; xs[4 * q + 3] = t.w; }
    const int img = (int)(mm / ((int64_t)d.H * d.W));
    const int rem = (int)(mm - (int64_t)img * d.H * d.W), y = rem / d.W, x = rem - y * d.W;
    const int64_t obase = ok ? (((int64_t)img * OH + 8 * y) * OW + 8 * x) * 8 : -1;
    {
      const int ty1 = tap1 >> 1, tx1 = tap1 & 1;
      // first layer, one tap (= one 32-row tile) at a time: 16 live accumulators instead of 64
      f32x16 accA;
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[r] = bA[r];
#pragma unroll
      for (int s = 0; s < 32; ++s) accA = __builtin_amdgcn_mfma_f32_32x32x2f32(wA[s], xs[s], accA, 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 16; ++r) accA[r] = act_apply_precise(accA[r], d.act1);
#pragma unroll
      for (int ty2 = 0; ty2 < 2; ++ty2) {
        f32x16 accB;
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = bB[(r & 3) + 4 * ((r >> 2) & 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) accB = __builtin_amdgcn_mfma_f32_32x32x2f32(wB[ty2][u], accA[u], accB, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 16; ++r) accB[r] = act_apply_precise(accB[r], d.act2);
#pragma unroll
        for (int tx2 = 0; tx2 < 2; ++tx2) {
          f32x16 accC;
#pragma unroll
          for (int r = 0; r < 16; ++r) accC[r] = bC[r & 3];
#pragma unroll
          for (int v = 0; v < 8; ++v) accC = __builtin_amdgcn_mfma_f32_32x32x2f32(wC[v], accB[8 * tx2 + v], accC, 0, 0, 0);
#pragma unroll
          for (int tb = 0; tb < 4; ++tb) {  // third-layer tap tb = (ty3, tx3), channels 4h .. 4h+3
            const float4 o = make_float4(act_apply_precise(accC[4 * tb], d.act3), act_apply_precise(accC[4 * tb + 1], d.act3),
                                         act_apply_precise(accC[4 * tb + 2], d.act3), act_apply_precise(accC[4 * tb + 3], d.act3));
            *reinterpret_cast<float4*>(stage + (tb >> 1) * (32 * PAIR_PITCH) + l31 * PAIR_PITCH + (4 * tx2 + 2 * (tb & 1) + h) * 16) = o;
          }
        }
        // rows ty3 = 0, 1 of (ty1, ty2), the 4-pixel segment tx1: lane -> (pixel = lane/8 + 8q, chunk = lane%8)
#pragma unroll
        for (int ty3 = 0; ty3 < 2; ++ty3)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int pix = (lane >> 3) + 8 * q;
            const float4 o = *reinterpret_cast<const float4*>(stage + ty3 * (32 * PAIR_PITCH) + pix * PAIR_PITCH + (lane & 7) * 16);
            const int64_t base = __shfl(obase, pix, 64);
            if (base >= 0) *reinterpret_cast<float4*>(Y + base + ((int64_t)(4 * ty1 + 2 * ty2 + ty3) * OW + 4 * tx1) * 8 + (lane & 7) * 4) = o;
          }
      }
    }
  }
}

hipError_t launch_convt_triple_f32(const TripleDesc& d, const float* X, const float* w1, const float* b1, const float* w2, const float* b2,
                                   const float* w3, const float* b3, float* Y, hipStream_t s) {
  const int64_t M = (int64_t)d.n * d.H * d.W;
  if (M == 0) return hipSuccess;
  const int64_t groups = (M + 31) / 32;
  const int blocks = (int)std::min<int64_t>(groups, 256 * 2);  // one workgroup per pixel group at a time (wave = tap); two resident per CU
  hipLaunchKernelGGL(convt_triple_f32, dim3(blocks), dim3(256), 0, s, d, X, w1, b1, w2, b2, w3, b3, Y);
  return hipGetLastError();
}

hipError_t launch_standardize(const float* x, float* y, const float* affine, int per_sample, int64_t total, hipStream_t s) {
  if (total == 0) return hipSuccess;
  hipLaunchKernelGGL(standardize_f32, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, y, affine, per_sample, total);
  return hipGetLastError();
}

hipError_t launch_finalize(const float* y, void* out, int out_dtype, const float* affine, int per_sample, int64_t total,
                           int nan_guard, unsigned long long* nonfinite, hipStream_t s) {
  if (total == 0) return hipSuccess;
  dim3 grid((unsigned)((total + 255) / 256));
  if (out_dtype == SRCFD_F32) hipLaunchKernelGGL(finalize_out<0>, grid, dim3(256), 0, s, y, out, affine, per_sample, total, nan_guard, nonfinite);
  else if (out_dtype == SRCFD_BF16) hipLaunchKernelGGL(finalize_out<1>, grid, dim3(256), 0, s, y, out, affine, per_sample, total, nan_guard, nonfinite);
  else if (out_dtype == SRCFD_F16) hipLaunchKernelGGL(finalize_out<2>, grid, dim3(256), 0, s, y, out, affine, per_sample, total, nan_guard, nonfinite);
  else return hipErrorInvalidValue;
  return hipGetLastError();
}

}  // namespace srcfd
