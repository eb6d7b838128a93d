// mid16: ConvT#0 (3x3, stride 2, 256->128, 12x12 -> 25x25) chained into ConvT#1 (2x2,
// stride 2, 128->64, -> 50x50) in one kernel (SURVEY.md 8a rows a13 + a14), 16-bit operands.
//
// Why not the generic implicit GEMM: measured on MI355X it spent ~0.45 ms/batch on these two
// layers, bound by re-fetching the same input pixels from beyond L2 once per kernel tap, and by
// writing / re-reading the 25x25x128 activation.  Here
//   * ConvT#0 is split into its four output phases (blockIdx.y); a workgroup owns 128 output
//     pixels of one phase and stages the contiguous slab of input pixels they touch ("patch",
//     <= 176 pixels) in LDS once per 64-channel chunk; every tap then gathers its B operand from
//     LDS with a per-lane row index, so the input is read from memory once, not once per tap;
//   * a wave owns 32 pixels x all 128 channels (4 accumulator tiles), so after bias + swish the
//     accumulators, packed to 16 bits, ARE the B operands of ConvT#1 (k order permuted on the
//     host to the accumulator's register order): 64 more MFMAs per wave produce the four 2x2 taps
//     x 64 channels, and only the 50x50x64 result goes to HBM.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev16.h"
#include "kernels16.h"

namespace srcfd {

constexpr int M_PITCH = 72;          // LDS row pitch in elements (144 B)
constexpr int M_PATCH = 176;         // patch rows; row M_PATCH is all zero (out-of-image taps)
constexpr int M_OFF_W = (M_PATCH + 1) * M_PITCH * 2;           // bytes
constexpr int M_OFF_META = M_OFF_W + 128 * M_PITCH * 2;
constexpr int M_LDS = M_OFF_META + 3 * 128 * 4;
static_assert(M_OFF_META >= 32768, "ConvT#1 operand half (32 KB) is staged over the patch + weight tiles");

template <bool F16>
__global__ void __launch_bounds__(256, 3) mid16(MidParams p) {
  extern __shared__ __attribute__((aligned(16))) char msm[];
  uint16_t* Ps = reinterpret_cast<uint16_t*>(msm);
  uint16_t* Ws = reinterpret_cast<uint16_t*>(msm + M_OFF_W);
  int* row_img = reinterpret_cast<int*>(msm + M_OFF_META);
  int* row_my = row_img + 128;
  int* row_mx = row_my + 128;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int phase = blockIdx.y, py = phase >> 1, px = phase & 1;
  const int TY = py ? 1 : 2, TX = px ? 1 : 2, NT = TY * TX;
  const int MH = py ? 12 : 13, MW = px ? 12 : 13, per = MH * MW;
  const int M = p.n * per, m0 = blockIdx.x * 128;
  if (m0 >= M) return;
  const uint16_t* Wt = p.w0[phase];
  const int Kp = p.kpad[phase];

  if (tid < 128) {
    int m = m0 + tid, img = -1, my = 0, mx = 0;
    if (m < M) { img = m / per; int r = m - img * per; my = r / MW; mx = r - my * MW; }
    row_img[tid] = img; row_my[tid] = my; row_mx[tid] = mx;
  }
  if (tid < M_PITCH / 2) reinterpret_cast<uint32_t*>(Ps + M_PATCH * M_PITCH)[tid] = 0;
  __syncthreads();

  // contiguous slab of input pixels (tensor index img*144 + iy*12 + ix) this workgroup touches
  const int mlast = min(m0 + 127, M - 1) - m0;
  const int lo = row_img[0] * 144 + max(row_my[0] - 1, 0) * 12;
  const int hi = row_img[mlast] * 144 + min(row_my[mlast], 11) * 12 + 11;
  const int NP = hi - lo + 1;  // <= M_PATCH by construction (<= 13 input rows of 12 + one row of slack)

  // this lane's pixel and the patch row each tap reads
  const int prow = wave * 32 + l31;
  const int img = row_img[prow], my = row_my[prow], mx = row_mx[prow];
  int trow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ty = t / TX, tx = t - ty * TX;  // only t < NT is used
    const int iy = my - ty, ix = mx - tx;
    const bool ok = img >= 0 && t < NT && (unsigned)iy < 12u && (unsigned)ix < 12u;
    trow[t] = (ok ? img * 144 + iy * 12 + ix - lo : M_PATCH) * M_PITCH + h * 8;
  }

  // staging roles: 16-byte column c8 of rows xrow + 32j
  const int xrow = tid >> 3, c8 = tid & 7;
  uint4 wr[4], pr[6];
  auto g2r_w = [&](int t, int c) {
#pragma unroll
    for (int j = 0; j < 4; ++j) wr[j] = *reinterpret_cast<const uint4*>(Wt + (xrow + 32 * j) * Kp + t * 256 + c * 64 + c8 * 8);
  };
  auto g2r_p = [&](int c) {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int r = xrow + 32 * j;
      pr[j] = r < NP ? *reinterpret_cast<const uint4*>(p.in + (size_t)(lo + r) * 256 + c * 64 + c8 * 8) : make_uint4(0, 0, 0, 0);
    }
  };
  auto r2l_w = [&]() {
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(Ws + (xrow + 32 * j) * M_PITCH + c8 * 8) = wr[j];
  };
  auto r2l_p = [&]() {
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int r = xrow + 32 * j;
      if (r < M_PATCH) *reinterpret_cast<uint4*>(Ps + r * M_PITCH + c8 * 8) = pr[j];
    }
  };

  // accumulators start at the bias (ConvT#0 bias as an MFMA C operand)
  f32x16 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = load_bias16(reinterpret_cast<const char*>(p.b0f) + (mt * 2 + h) * 64);

  const int NS = 4 * NT;  // stages: 64-channel chunk c = s / NT (outer), tap t = s % NT (inner)
  g2r_w(0, 0);
  g2r_p(0);
  const uint16_t* wsl = Ws + l31 * M_PITCH + h * 8;
  for (int s = 0; s < NS; ++s) {
    const int c = s / NT, t = s - c * NT;
    r2l_w();
    if (t == 0) r2l_p();
    __syncthreads();
    if (s + 1 < NS) {
      const int c1 = (s + 1) / NT, t1 = (s + 1) - c1 * NT;
      g2r_w(t1, c1);
      if (t1 == 0) g2r_p(c1);
    }
    const uint16_t* bsrc = Ps + trow[t];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint4 bf = *reinterpret_cast<const uint4*>(bsrc + kk * 16);
      uint4 af[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const uint4*>(wsl + mt * 32 * M_PITCH + kk * 16);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma32<F16>(af[mt], bf, acc[mt]);
    }
    __syncthreads();
  }

  // ---- ConvT#0 epilogue: swish, pack; the packed accumulators are ConvT#1's B operands ----
  uint32_t fb[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) swish_pack16<F16>(acc[mt], fb[mt]);

  // ---- ConvT#1: 8 tiles of 32 rows (tap = j8 >> 1, channels 32*(j8&1)..+31), K = 128 = 8 k-steps.
  // Its 64 KB of A operands go through the (now free) LDS in two halves, shared by the four waves.
  const uint4* w1 = reinterpret_cast<const uint4*>(p.w1f);
  uint4* w1s = reinterpret_cast<uint4*>(msm);
  const int Y = 2 * my + py, X = 2 * mx + px;  // 25x25-level pixel
#pragma unroll 1
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int j = 0; j < 8; ++j) w1s[tid + 256 * j] = w1[half * 2048 + tid + 256 * j];
    __syncthreads();
#pragma unroll 1
    for (int jj = 0; jj < 4; ++jj) {
      const int j8 = half * 4 + jj;
      f32x16 a1 = load_bias16(reinterpret_cast<const char*>(p.b1f) + ((j8 & 1) * 2 + h) * 64);
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const uint4 wf = w1s[(jj * 8 + s) * 64 + lane];
        const uint4 bf = make_uint4(fb[s >> 1][4 * (s & 1)], fb[s >> 1][4 * (s & 1) + 1], fb[s >> 1][4 * (s & 1) + 2], fb[s >> 1][4 * (s & 1) + 3]);
        a1 = mfma32<F16>(wf, bf, a1);
      }
      uint32_t o[8];
      swish_pack16<F16>(a1, o);
      if (img >= 0) {
        const int tap = j8 >> 1, Yo = 2 * Y + (tap >> 1), Xo = 2 * X + (tap & 1);
        uint16_t* dst = p.out + ((size_t)(img * 50 + Yo) * 50 + Xo) * 64 + 32 * (j8 & 1) + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(dst + 8 * q) = make_uint2(o[2 * q], o[2 * q + 1]);
      }
    }
    __syncthreads();
  }
}

int mid16_lds_bytes() { return M_LDS; }

hipError_t launch_mid16(bool f16, const MidParams& p, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  static bool attr_done[2] = {};
  void (*fn)(MidParams) = f16 ? mid16<true> : mid16<false>;
  if (!attr_done[f16 ? 1 : 0]) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, M_LDS);
    if (e != hipSuccess) return e;
    attr_done[f16 ? 1 : 0] = true;
  }
  const int blocks = (p.n * 169 + 127) / 128;  // the largest phase (13x13 pixels per sample)
  hipLaunchKernelGGL(fn, dim3(blocks, 4), dim3(256), M_LDS, s, p);
  return hipGetLastError();
}

}  // namespace srcfd
