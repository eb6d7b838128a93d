// gemm32_big: the f32 implicit GEMM for LARGE launches of the decoder's wide layers (gfx950 only).
//
// Same formulation and the same arithmetic as gemm_mfma_f32 (kernels_fp32.hip: rows = output pixels with an affine
// gather, k = (ty, tx, ci), v_mfma_f32_32x32x2_f32 k-ordered chains; results are bit-identical to it), for the launches
// that fill the chip: ConvT#0's four output phases and ConvT#1 at batch sizes of a hundred samples and more
// (sr-ae-conv.ipynb:c281-282).  The generic kernel spent as many cycles on vector
// instructions as on the MFMAs there (rocprofv3: SQ_ACTIVE_INST_VALU ~ SQ_VALU_MFMA_BUSY_CYCLES, 35 % of the kernel
// each), and on this chip the two do not overlap (tools/microbench8.hip).  What changes:
//   * the gather's per-row work (image, pixel, bounds) is hoisted: a K slab of 16 sits inside one tap (CI % 16 == 0),
//     so a thread recomputes its two source addresses only when the tap changes, not per 16-byte load;
//   * the weight tile is fetched as 16-byte pieces;
//   * a K split that the generic path does across workgroups (f32 slabs + a finish kernel, so that a sample's sums do not
//     depend on its batch: gemm_splitk_splits) is done inside the workgroup: one MFMA chain per `kchunk`-deep slab,
//     added in slab order in registers, then the bias -- the same additions in the same order, without the slabs' round
//     trip through HBM (ConvT#0: 0.5 GB written + read per 256 samples) or the finish launch;
//   * the epilogue computes a row's output address once (not once per column tile), adds the lane's column offset, and
//     applies bias + activation on the accumulators.
// The output phases of a transposed convolution go out as ONE launch (blockIdx.z = phase), like gemm_mfma_group_f32.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "kernels.h"

namespace srcfd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int GB_BM = 128, GB_BN = 128, GB_BK = 16, GB_LDA = GB_BK + 1;

// x * rcp(1 + exp2(-log2(e) x)): kernels_fp32.hip, act_apply_precise (same expression, same bits)
__device__ __forceinline__ float gb_swish(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)); }

template <bool SPLIT>   // SPLIT: some phase sums more than one K slab (needs a second accumulator set)
__global__ void __launch_bounds__(256) gemm32_big(Gemm32Group g, const float* __restrict__ X, float* __restrict__ Y) {
  __shared__ float As[GB_BM * GB_LDA];
  __shared__ __attribute__((aligned(16))) float Bs[GB_BK * GB_BN];
  __shared__ int64_t row_out[GB_BM];   // element offset of every tile row's output pixel (-1: past M): the two integer divisions per row are done once per workgroup, not once per accumulator register
  const int ph = blockIdx.z;
  const GemmDesc& d = g.d[ph];
  const int m0 = blockIdx.x * GB_BM, n0 = blockIdx.y * GB_BN;
  if (m0 >= d.M) return;
  const float* __restrict__ B = g.B[ph];
  const float* __restrict__ bias = g.bias[ph];
  const int kchunk = g.kchunk[ph];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;

  // ---- A gather: thread -> rows (tid / 4) + 64 j, 16-byte k group tid % 4 of every slab ----
  const int cg = tid & 3;
  const int per = d.MH * d.MW;
  int64_t a_img[2];
  int a_y[2], a_x[2];
  bool a_ok[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int m = m0 + (tid >> 2) + 64 * j;
    a_ok[j] = m < d.M;
    const int mm = a_ok[j] ? m : 0;
    const int img = mm / per, r = mm - img * per, my = r / d.MW, mx = r - my * d.MW;
    a_img[j] = (int64_t)img * d.IH * d.IW * d.CI;
    a_y[j] = my * d.ay + d.cy;
    a_x[j] = mx * d.ax + d.cx;
  }
  if (tid < GB_BM) {
    const int m = m0 + tid;
    int64_t o = -1;
    if (m < d.M) {
      const int img = m / per, r = m - img * per, my = r / d.MW, mx = r - my * d.MW;
      o = (((int64_t)img * d.OH + my * d.os + d.oy0) * d.OW + mx * d.os + d.ox0) * d.OC;
    }
    row_out[tid] = o;
  }
  // ---- B tile: thread -> k row tid / 32 (+ 8), columns 4 (tid % 32) ----
  const float* bsrc = B + (int64_t)(tid >> 5) * d.Npad + n0 + 4 * (tid & 31);

  f32x4 ra[2], rb[2];
  int cur_tap = -1;
  const float* asrc[2] = {X, X};
  bool aval[2] = {false, false};
  auto fetch = [&](const int k0) {
    const int tap = k0 / d.CI, ci0 = k0 - tap * d.CI + 4 * cg;
    if (tap != cur_tap) {          // wave-uniform: a slab of 16 never straddles two taps (CI % 16 == 0)
      cur_tap = tap;
      const int ty = tap / d.TX, tx = tap - ty * d.TX;
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const int iy = a_y[j] + ty * d.by, ix = a_x[j] + tx * d.bx;
        aval[j] = a_ok[j] && iy >= 0 && iy < d.IH && ix >= 0 && ix < d.IW;
        asrc[j] = X + a_img[j] + ((int64_t)iy * d.IW + ix) * d.CI;
      }
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) ra[j] = aval[j] ? *reinterpret_cast<const f32x4*>(asrc[j] + ci0) : f32x4{0.f, 0.f, 0.f, 0.f};
    const float* bp = bsrc + (int64_t)k0 * d.Npad;
    rb[0] = *reinterpret_cast<const f32x4*>(bp);
    rb[1] = *reinterpret_cast<const f32x4*>(bp + (int64_t)8 * d.Npad);
  };
  auto park = [&]() {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float* dst = As + ((tid >> 2) + 64 * j) * GB_LDA + 4 * cg;
      dst[0] = ra[j][0]; dst[1] = ra[j][1]; dst[2] = ra[j][2]; dst[3] = ra[j][3];
    }
    *reinterpret_cast<f32x4*>(Bs + (tid >> 5) * GB_BN + 4 * (tid & 31)) = rb[0];
    *reinterpret_cast<f32x4*>(Bs + ((tid >> 5) + 8) * GB_BN + 4 * (tid & 31)) = rb[1];
  };

  f32x16 tot[4], acc[4];
  const float* ap = As + (wave * 32 + (lane & 31)) * GB_LDA + (lane >> 5);
  const float* bp = Bs + (lane >> 5) * GB_BN + (lane & 31);
  fetch(0);
  for (int kb = 0; kb < d.K; kb += kchunk) {       // one MFMA chain per K slab of the batch-invariant split
    const int kend = min(d.K, kb + kchunk);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int k0 = kb; k0 < kend; k0 += GB_BK) {
      park();
      __syncthreads();
      if (k0 + GB_BK < d.K) fetch(k0 + GB_BK);
#pragma unroll
      for (int kk = 0; kk < GB_BK; kk += 2) {
        const float a = ap[kk];
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bp[kk * GB_BN + 32 * i], acc[i], 0, 0, 0);
      }
      __syncthreads();
    }
    if (SPLIT) {
      if (kb == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) tot[i] = acc[i];
      } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) tot[i] += acc[i];   // slab order, as splitk_finish adds them (0 + p0 = p0 exactly)
      }
    }
  }
  if (!SPLIT) {
#pragma unroll
    for (int i = 0; i < 4; ++i) tot[i] = acc[i];
  }

  // ---- epilogue ----
  // column side: n -> (phase, channel) -> offset inside the output pixel block; row side: one address per accumulator row
  int64_t col_off[4];
  float bv[4];
  bool col_ok[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int n = n0 + 32 * i + (lane & 31);
    col_ok[i] = n < d.N;
    const int nn = col_ok[i] ? n : 0;
    const int pz = nn / d.CO, co = nn - pz * d.CO, py = pz / d.nphx, px = pz - py * d.nphx;
    col_off[i] = ((int64_t)py * d.OW + px) * d.OC + co;
    bv[i] = bias[nn];
  }
  const bool sw = d.act == SRCFD_ACT_SWISH;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int64_t ro = row_out[wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];   // written before the K loop's barriers
    if (ro < 0) continue;
    float* yrow = Y + ro;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!col_ok[i]) continue;
      float v = tot[i][r] + bv[i];
      if (sw) v = gb_swish(v);
      yrow[col_off[i]] = v;
    }
  }
}

// A layer (1 GEMM) or the output phases of a transposed convolution (2-4 GEMMs) qualify when every GEMM is one of the
// wide decoder layers this kernel is written for and the launch is large enough that the generic kernel would take its
// widest tile anyway (so the choice does not depend on anything but the layer and "is the batch large").
bool gemm32_big_qualifies(const GemmDesc* ds, int count) {
  if (count < 1 || count > 4) return false;
  int64_t tiles = 0;
  for (int i = 0; i < count; ++i) {
    const GemmDesc& d = ds[i];
    if (d.M <= 0 || d.K <= 0 || d.CI % 16 != 0 || d.K % 16 != 0 || d.Npad % 128 != 0 || d.Npad != ds[0].Npad) return false;
    if (d.act != SRCFD_ACT_SWISH && d.act != SRCFD_ACT_LINEAR) return false;
    int kchunk = d.K;
    (void)gemm_splitk_splits(d, &kchunk, true);
    if (kchunk % GB_BK != 0) return false;   // a K slab of the batch-invariant split must be whole 16-deep tiles
    tiles += (int64_t)((d.M + GB_BM - 1) / GB_BM) * (d.Npad / GB_BN);
  }
  return tiles >= 1024;
}

hipError_t launch_gemm32_big(const GemmDesc* ds, int count, const float* X, const float* const* Bs, const float* const* biases, float* Y,
                             hipStream_t s) {
  Gemm32Group g{};
  int max_rows = 0;
  for (int i = 0; i < count; ++i) {
    g.d[i] = ds[i]; g.B[i] = Bs[i]; g.bias[i] = biases[i];
    int kchunk = ds[i].K;
    (void)gemm_splitk_splits(ds[i], &kchunk, true);   // the batch-invariant cut of the generic path: the same slabs, summed in-kernel
    g.kchunk[i] = kchunk;
    max_rows = std::max(max_rows, (ds[i].M + GB_BM - 1) / GB_BM);
  }
  g.count = count;
  bool split = false;
  for (int i = 0; i < count; ++i) split = split || g.kchunk[i] < ds[i].K;
  if (split) hipLaunchKernelGGL(gemm32_big<true>, dim3(max_rows, ds[0].Npad / GB_BN, count), dim3(256), 0, s, g, X, Y);
  else hipLaunchKernelGGL(gemm32_big<false>, dim3(max_rows, ds[0].Npad / GB_BN, count), dim3(256), 0, s, g, X, Y);
  return hipGetLastError();
}

}  // namespace srcfd

namespace srcfd {

// ---------------------------------------------------------------------------
// dense_skinny32: a Dense layer with a short K and a very wide N (decoder dense_1: 50 -> 36 864, swish; SURVEY.md 8a
// row a11), f32.  On the generic implicit GEMM this layer took 0.098 ms per 768 samples for 2.8 GFLOP: K is a single
// shallow slab, so every workgroup was one global-load -> LDS -> MFMA -> store latency chain.  Here a workgroup owns 144
// output features (36 864 = 256 x 144: one workgroup per CU) for ALL samples: its weights (<= 52 x 144 f32) stay in
// registers as v_mfma_f32_16x16x4_f32 A operands for the whole kernel, each wave walks 16-sample tiles (B = the input
// rows, straight from memory / L2), and the activated tile leaves through a wave-private LDS transpose as 576-byte runs
// per sample.  No workgroup barrier.  Every output column (sample) accumulates over k in the same fixed order whatever the
// batch: results do not depend on the batch size or on the sample's position.
// ---------------------------------------------------------------------------
constexpr int DS_FT = 9, DS_NF = DS_FT * 16, DS_KS = 13, DS_WAVES = 8;   // K <= 4 * DS_KS = 52
constexpr int DS_PITCH = DS_NF * 4 + 16;                                 // LDS row pitch of the transpose tile, bytes
constexpr int DS_LDS = DS_WAVES * 16 * DS_PITCH;

__global__ void __launch_bounds__(64 * DS_WAVES, 1) dense_skinny32(GemmDesc d, const float* __restrict__ X, const float* __restrict__ B,
                                                                    const float* __restrict__ bias, float* __restrict__ Y) {
  extern __shared__ __attribute__((aligned(16))) char dssm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.x * DS_NF, K = d.K, M = d.M, N = d.OC;
  float wa[DS_FT][DS_KS];
  f32x4 bs[DS_FT];
#pragma unroll
  for (int ft = 0; ft < DS_FT; ++ft) {
#pragma unroll
    for (int ks = 0; ks < DS_KS; ++ks) {
      const int k = ks * 4 + kg;
      wa[ft][ks] = k < K ? B[(size_t)k * d.Npad + n0 + ft * 16 + col] : 0.f;
    }
    bs[ft] = *reinterpret_cast<const f32x4*>(bias + n0 + ft * 16 + kg * 4);
  }
  char* st = dssm + wave * 16 * DS_PITCH;
  const int tiles = (M + 15) / 16;
  const bool swish = d.act == SRCFD_ACT_SWISH;
  for (int t = wave; t < tiles; t += DS_WAVES) {
    const int sv = min(t * 16 + col, M - 1);
    float xb[DS_KS];
#pragma unroll
    for (int ks = 0; ks < DS_KS; ++ks) {
      const int k = ks * 4 + kg;
      xb[ks] = k < K ? X[(size_t)sv * K + k] : 0.f;
    }
#pragma unroll
    for (int ft = 0; ft < DS_FT; ++ft) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int ks = 0; ks < DS_KS; ++ks) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[ft][ks], xb[ks], acc, 0, 0, 0);
      acc += bs[ft];
      if (swish) {
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] = gb_swish(acc[r]);
      }
      *reinterpret_cast<f32x4*>(st + col * DS_PITCH + ft * 64 + kg * 16) = acc;
    }
    // the tile is wave-private: lanes read chunks other lanes of the SAME wave wrote.  The hardware executes one wave's LDS
    // operations in order; the wave-scope fence + barrier keep the compiler from moving these loads above the stores (and, below,
    // the next tile's stores above these loads) -- no instruction is emitted for either
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // write-out: 16 samples x 36 chunks of 16 B, consecutive lanes on consecutive chunks of one sample's 576-byte run
#pragma unroll
    for (int r = 0; r < 16 * DS_NF / 4 / 64; ++r) {
      const int c = lane + 64 * r, row = c / (DS_NF / 4), ch = c - row * (DS_NF / 4), s = t * 16 + row;
      if (s < M) *reinterpret_cast<f32x4*>(Y + (size_t)s * N + n0 + ch * 4) = *reinterpret_cast<const f32x4*>(st + row * DS_PITCH + ch * 16);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
}

bool dense_skinny32_qualifies(const GemmDesc& d) {
  return d.MH == 1 && d.MW == 1 && d.TY == 1 && d.TX == 1 && d.K <= 4 * DS_KS && d.CI == d.K && d.N == d.Npad && d.N % DS_NF == 0 && d.OC == d.N &&
         d.N >= 64 * DS_NF && (d.act == SRCFD_ACT_SWISH || d.act == SRCFD_ACT_LINEAR);
}

hipError_t launch_dense_skinny32(const GemmDesc& d, const float* X, const float* B, const float* bias, float* Y, hipStream_t s) {
  if (d.M == 0) return hipSuccess;
  static thread_local int attr_dev = -1;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (attr_dev != dev) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(dense_skinny32), hipFuncAttributeMaxDynamicSharedMemorySize, DS_LDS);
    if (e != hipSuccess) return e;
    attr_dev = dev;
  }
  hipLaunchKernelGGL(dense_skinny32, dim3(d.N / DS_NF), dim3(64 * DS_WAVES), DS_LDS, s, d, X, B, bias, Y);
  return hipGetLastError();
}

}  // namespace srcfd
