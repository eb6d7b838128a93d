// Training step, forward pass of encoder_10 (conv2d 3x3 s2 1->64 -> conv2d_1 3x3 s1 64->128 on 5x5 -> flatten + dense 3200->128
// -> latent_vector 128->nl; sr-ae-conv.ipynb:c162-169) as TWO launches instead of six (train_enc.h).
//
// Why: at micro-batch sizes these four layers are ~0.04 GFLOP -- six generic launches (three GEMMs + their split-K sums) of 5-12 us
// each on the step's dependent chain, 45 us of a 0.45 ms step (now 17 + 5 us).  The inference path's one-launch encoder (enc32) keeps a sample's
// whole chain in ONE workgroup: 49 us whatever the batch, because one CU streams the dense layer's 1.6 MB of weights alone.
// Here the dense layer's K dimension is what the grid is cut along:
//   launch A, workgroup (pixel p of the 5x5 level, channel half hf, group of 16 samples): conv2d for the 3x3 neighbourhood of p
//     (vector unit, recomputed by every workgroup that needs it: 9 x 64 x 9 MACs per sample), conv2d_1 for (p, 64 channels) on
//     v_mfma_f32_16x16x4_f32 (rows = channels, columns = samples, K = 576), then THIS slice's contribution to the dense layer
//     (K = the 64 flattened inputs (p, channel): 32 KB of the layer's weights, read by no other workgroup) -> partial[slice][sample][128];
//   launch B, one workgroup per sample: the 50 slices summed in slice order + bias, swish, latent_vector.
// Every layer's pre-activation and activation are stored for the backward pass, which stays layer by layer.
#include "train_enc.h"

#include <hip/hip_runtime.h>

#include "act_device.h"

namespace srcfd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int TE_NS = 16;                       // samples per workgroup of launch A (the MFMA's columns)
constexpr int TE_X = 0;                         // LDS: x [16][100]
constexpr int TE_Y0 = TE_X + TE_NS * 100;       //      y0 of the 9 neighbours [tap 9][ci 64][sample 16]
constexpr int TE_Y1 = TE_Y0 + 9 * 64 * TE_NS;   //      y1 of (p, 64 channels) [c 64][sample 16]
constexpr int TE_LDS_FLOATS = TE_Y1 + 64 * TE_NS;

__device__ __forceinline__ float te_act(float z, int swish) { return swish ? swish_train(z) : z; }

__global__ void __launch_bounds__(256) train_enc_a(TrainEncParams q) {
  __shared__ __attribute__((aligned(16))) float sm[TE_LDS_FLOATS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
  const int slice = blockIdx.x, p = slice >> 1, hf = slice & 1, py = p / 5, px = p - py * 5;
  const int s0 = blockIdx.y * TE_NS, ns = min(TE_NS, q.n - s0);

  // The matrix stages' A operands (this wave's 16 channels of conv2d_1: one float per k-step and lane; its two feature tiles of the
  // dense slice) depend on nothing computed here: all 176 loads are issued first and land while conv2d runs -- read at their point
  // of use they were 144 + 32 dependent round trips to L2 (29 us for this launch instead of 17).  The small operands of the vector
  // stage and the biases likewise.  (A pack in MFMA-fragment order -- 36 + 8 sixteen-byte loads per lane instead of 176 four-byte
  // ones -- was measured: 17.4 -> 16.6 us here, + 2.8 us in the step's operand gather for the 0.5 M extra slots.  Not kept: what is
  // left of the launch is a sum of small latencies, not load instructions.)
  const int c0 = 64 * hf + 16 * wave;
  float wa[144], wdn[2][16], w9[9];
  const int ci = tid >> 2, sq = 4 * (tid & 3);                     // conv2d: thread = (channel ci, four samples)
#pragma unroll
  for (int t = 0; t < 9; ++t) w9[t] = q.w0[t * 64 + ci];
  const float bias = q.b0[ci];
  const f32x4 bv = *reinterpret_cast<const f32x4*>(q.b1 + c0 + 4 * kq);
  {
    const float* w = q.w1 + (size_t)kq * 128 + c0 + col;          // A[m = col][k = kq] of k-step j: row 4 j + kq of B[576][128]
#pragma unroll
    for (int j = 0; j < 144; ++j) wa[j] = w[(size_t)j * 4 * 128];
    const int krow0 = p * 128 + 64 * hf;                          // first of the slice's rows of B[3200][128]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float* wd = q.wd + (size_t)(krow0 + kq) * 128 + 16 * (2 * wave + t) + col;
#pragma unroll
      for (int j = 0; j < 16; ++j) wdn[t][j] = wd[(size_t)j * 4 * 128];
    }
  }
  for (int i = tid; i < TE_NS * 100; i += 256) {
    const int s = i / 100;
    sm[TE_X + i] = s < ns ? q.x[(size_t)(s0 + s) * 100 + (i - s * 100)] : 0.f;
  }
  __syncthreads();
  // conv2d (3x3, stride 2, TF SAME = pad bottom / right only) for the nine neighbours of p (zeros outside the 5x5 level: conv2d_1's
  // padding); thread = (channel ci, four samples), its nine weights in registers
  {
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
      const int oy = py + tap / 3 - 1, ox = px + tap % 3 - 1;
      const bool in = (unsigned)oy < 5u && (unsigned)ox < 5u;      // block-uniform
      f32x4 y0v = {0.f, 0.f, 0.f, 0.f};
      if (in) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const float* xs = sm + TE_X + (sq + u) * 100;
          float z = bias;
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            const int iy = 2 * oy + ky;
            if (iy >= 10) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
              const int ix = 2 * ox + kx;
              if (ix >= 10) continue;
              z = fmaf(xs[iy * 10 + ix], w9[ky * 3 + kx], z);
            }
          }
          y0v[u] = te_act(z, q.swish[0]);
          if (tap == 4 && hf == 0 && sq + u < ns) {      // the centre pixel's owner keeps the layer's values
            const size_t o = ((size_t)(s0 + sq + u) * 25 + p) * 64 + ci;
            q.z0[o] = z;
            if (q.y0 != q.z0) q.y0[o] = y0v[u];
          }
        }
      }
      *reinterpret_cast<f32x4*>(sm + TE_Y0 + (tap * 64 + ci) * TE_NS + sq) = y0v;
    }
  }
  __syncthreads();
  // conv2d_1 for (p, channels 64 hf + 16 wave + m): D[channel m][sample], K = (tap, ci) = 576 = 144 k-steps of 4
  {
    const float* b = sm + TE_Y0 + kq * TE_NS + col;               // B[k = kq][n = col]
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 144; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[j], b[j * 4 * TE_NS], acc, 0, 0, 0);
    // lane (sample col, row group kq): channels c0 + 4 kq + i
    f32x4 z, y;
#pragma unroll
    for (int i = 0; i < 4; ++i) { z[i] = acc[i] + bv[i]; y[i] = te_act(z[i], q.swish[1]); }
    if (col < ns) {
      const size_t o = ((size_t)(s0 + col) * 25 + p) * 128 + c0 + 4 * kq;   // flattened NHWC (h*5+w)*128 + c
      *reinterpret_cast<f32x4*>(q.z1 + o) = z;
      if (q.y1 != q.z1) *reinterpret_cast<f32x4*>(q.y1 + o) = y;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) sm[TE_Y1 + (16 * wave + 4 * kq + i) * TE_NS + col] = y[i];
  }
  __syncthreads();
  // this slice of the dense layer: D[feature][sample] += sum over the 64 inputs (p, 64 hf + c); wave w: feature tiles 2w, 2w + 1
  {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int f0 = 16 * (2 * wave + t);
      const float* b = sm + TE_Y1 + kq * TE_NS + col;
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int j = 0; j < 16; ++j) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(wdn[t][j], b[j * 4 * TE_NS], acc, 0, 0, 0);
      if (col < ns) *reinterpret_cast<f32x4*>(q.partial + ((size_t)slice * q.n + s0 + col) * 128 + f0 + 4 * kq) = acc;
    }
  }
}

__global__ void __launch_bounds__(128) train_enc_b(TrainEncParams q) {
  __shared__ float y2s[128];
  const int s = blockIdx.x, f = threadIdx.x;
  // the 50 slices in slice order, as five chains of ten (fixed)
  float zz[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 10; ++i)
#pragma unroll
    for (int c = 0; c < 5; ++c) zz[c] += q.partial[((size_t)(10 * c + i) * q.n + s) * 128 + f];
  const float z = ((zz[0] + zz[1]) + (zz[2] + zz[3])) + zz[4] + q.bd[f];
  const float y = te_act(z, q.swish[2]);
  q.z2[(size_t)s * 128 + f] = z;
  if (q.y2 != q.z2) q.y2[(size_t)s * 128 + f] = y;
  y2s[f] = y;
  __syncthreads();
  if (f < q.nl) {
    float a[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < 128; k += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) a[j] = fmaf(y2s[k + j], q.wl[(size_t)(k + j) * q.nl_pad + f], a[j]);
    }
    const float z3 = (a[0] + a[1]) + (a[2] + a[3]) + q.bl[f];
    q.z3[(size_t)s * q.nl + f] = z3;
    if (q.y3 != q.z3) q.y3[(size_t)s * q.nl + f] = te_act(z3, q.swish[3]);
  }
}

bool train_enc_qualifies(const GemmDesc& c1, const GemmDesc& c2, const GemmDesc& de, const GemmDesc& la) {
  const bool conv0 = c1.TY == 3 && c1.TX == 3 && c1.CI == 1 && c1.N == 64 && c1.Npad == 64 && c1.IH == 10 && c1.IW == 10 && c1.MH == 5 && c1.MW == 5 &&
                     c1.ay == 2 && c1.ax == 2 && c1.by == 1 && c1.bx == 1 && c1.cy == 0 && c1.cx == 0 && c1.nphx == 1 && c1.CO == 64 && c1.OC == 64;
  const bool conv1 = c2.TY == 3 && c2.TX == 3 && c2.CI == 64 && c2.N == 128 && c2.Npad == 128 && c2.IH == 5 && c2.IW == 5 && c2.MH == 5 && c2.MW == 5 &&
                     c2.ay == 1 && c2.ax == 1 && c2.by == 1 && c2.bx == 1 && c2.cy == -1 && c2.cx == -1 && c2.nphx == 1 && c2.CO == 128 && c2.OC == 128;
  const bool dense = de.MH == 1 && de.MW == 1 && de.K == 3200 && de.N == 128 && de.Npad == 128 && de.OC == 128;
  const bool lat = la.MH == 1 && la.MW == 1 && la.K == 128 && la.N <= 128 && la.OC == la.N;
  return conv0 && conv1 && dense && lat;
}

hipError_t launch_train_enc(const TrainEncParams& q, hipStream_t s) {
  if (q.n <= 0) return hipSuccess;
  hipLaunchKernelGGL(train_enc_a, dim3(50, (q.n + TE_NS - 1) / TE_NS), dim3(256), 0, s, q);
  hipLaunchKernelGGL(train_enc_b, dim3(q.n), dim3(128), 0, s, q);
  return hipGetLastError();
}

}  // namespace srcfd
