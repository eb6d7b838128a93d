// enc32: the whole encoder_10 in one launch on the f32 (<= 1e-5) path:
//   standardise -> conv2d (3x3 s2, 1->64) -> conv2d_1 (3x3 s1, 64->128) -> flatten + dense (3200->128) -> latent_vector (128->50)
//
// Layer by layer (standardise, a vector-unit conv, three generic implicit GEMMs and two split-K finishes) these cost 0.134 ms per
// 768-sample batch for 3.5 GFLOP: 25 pixels or 1 row per sample leave every launch a latency chain.  Here a workgroup owns
// E3_G = 3 samples (256 workgroups per 768 samples: one per CU) from the 10x10 input to the latent vector, activations in LDS:
//   * conv2d on the vector units (same expressions as conv_ci1/standardize of the layer-by-layer path);
//   * conv2d_1 on v_mfma_f32_16x16x4_f32: a wave owns 16 output channels, its 144 weight fragments (K = 576) stay in registers,
//     the im2col B operand is one ds_read_b128 per four MFMAs (the k order inside a 16-channel group is permuted on the host
//     so that a lane's four consecutive k-steps are four consecutive channels);
//   * dense 3200->128 on the vector units: with 3 samples per workgroup an MFMA tile would be 13/16 padding, and the layer is
//     bound by streaming its 1.6 MB of weights through the CU anyway; 16 K-slices x 32 feature quads, partial sums added in
//     slice order;
//   * latent_vector: 150 dot products of 128.
// Every sample's sums run in an order that does not depend on the batch: results are bit-identical across batch sizes.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <cstdio>
#include <cstdlib>

#include "kernels.h"

namespace srcfd {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int E3_G = ENC32_G;
constexpr int E3_NTHR = 512;
constexpr int E3_P1 = 68;                              // conv2d-output row pitch, floats (64 + 4)
constexpr int E3_ZROW = E3_G * 25;                     // all-zero row
constexpr int E3_P2 = 3200 + 4;                        // conv2d_1-output pitch per sample, floats
constexpr int E3_OFF_A1 = 2048;                        // after x: E3_G * 100 f32
constexpr int E3_OFF_A2 = E3_OFF_A1 + (E3_ZROW + 1) * E3_P1 * 4;
constexpr int E3_OFF_PS = E3_OFF_A2 + E3_G * E3_P2 * 4;            // dense partial sums [16 slices][E3_G][128]
constexpr int E3_OFF_A3 = E3_OFF_PS + 16 * E3_G * 128 * 4;         // dense output [E3_G][128]
constexpr int E3_LDS = E3_OFF_A3 + E3_G * 128 * 4;
static_assert(E3_G * 100 * 4 <= E3_OFF_A1 && E3_G * 25 <= 80 && E3_G * 128 <= E3_NTHR, "workgroup shape");

__device__ __forceinline__ float e3_act(float v, int act) {   // kernels_fp32.hip, act_apply_precise (same expression, same bits)
  return act == SRCFD_ACT_SWISH ? v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(v * -1.4426950408889634f)) : v;
}

#ifdef SRCFD_DIAG
__device__ unsigned long long e3_prof[8 * 8];   // SRCFD_ENC32_PROF: cycle stamps of workgroup 7, [wave][stamp]
#define E3_STAMP() stamp[nstamp++] = __builtin_readcyclecounter()
#else
#define E3_STAMP()
#endif

__global__ void __launch_bounds__(E3_NTHR, 1) enc32(Enc32Params p) {
  extern __shared__ __attribute__((aligned(16))) char e3sm[];
  float* X0 = reinterpret_cast<float*>(e3sm);
  float* A1 = reinterpret_cast<float*>(e3sm + E3_OFF_A1);
  float* A2 = reinterpret_cast<float*>(e3sm + E3_OFF_A2);
  float* PS = reinterpret_cast<float*>(e3sm + E3_OFF_PS);
  float* A3 = reinterpret_cast<float*>(e3sm + E3_OFF_A3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kg = lane >> 4;
  const int s0 = blockIdx.x * E3_G, gv = min(E3_G, p.n - s0);
#ifdef SRCFD_DIAG
  unsigned long long stamp[8];
  int nstamp = 0;
#endif
  E3_STAMP();

  // ---- input: standardise_with_stats (PyCFD_ML_accelerated.py:665-668), the expression of standardize_f32 ----
  if (tid < E3_G * 100) {
    const int g = tid / 100;
    float v = 0.f;
    if (g < gv) {
      v = p.x[(size_t)s0 * 100 + tid];
      if (p.affine) {
        const float mean = p.affine[2 * (s0 + g)];
        float sd = p.affine[2 * (s0 + g) + 1];
        if (sd == 0.f) sd = 1e-8f;
        v = __fdiv_rn(__fsub_rn(v, mean), sd);
      }
    }
    X0[tid] = v;
  }
  if (tid < E3_P1) A1[E3_ZROW * E3_P1 + tid] = 0.f;
  // this thread's conv2d weights: it always works on channels 4 (tid % 16) .. +3 (512 % 16 == 0)
  f32x4 w1r[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w1r[t] = *reinterpret_cast<const f32x4*>(p.w1 + t * 64 + (tid & 15) * 4);
  const f32x4 b1r = *reinterpret_cast<const f32x4*>(p.b1 + (tid & 15) * 4);
  __syncthreads();
  E3_STAMP();

  // conv2d_1's A operands: all 36 fragments (144 registers) of this wave's channel tile, in flight while conv2d runs
  f32x4 w2r[36];
  {
    const f32x4* w2 = reinterpret_cast<const f32x4*>(p.w2f) + (size_t)wave * 36 * 64 + lane;
#pragma unroll
    for (int tq = 0; tq < 36; ++tq) w2r[tq] = w2[tq * 64];
  }

  // ---- conv2d: 3x3 stride 2, TF SAME (pad 0 before / 1 after), 1 -> 64; one item = 4 channels of one pixel; k order (ky, kx)
  // as in the layer-by-layer kernel, one fmaf per tap ----
  constexpr int C1_ITEMS = (E3_G * 25 * 16 + E3_NTHR - 1) / E3_NTHR;
#pragma unroll
  for (int it = 0; it < C1_ITEMS; ++it) {
    const int idx = tid + it * E3_NTHR;
    if (idx >= E3_G * 25 * 16) continue;
    const int pix = (idx >> 4) % 25, g = idx / (25 * 16);
    const int oy = pix / 5, ox = pix - oy * 5;
    f32x4 acc = b1r;
    const float* xs = X0 + g * 100;
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
      const int iy = 2 * oy + ky;
      if (iy >= 10) continue;
#pragma unroll
      for (int kx = 0; kx < 3; ++kx) {
        const int ix = 2 * ox + kx;
        if (ix >= 10) continue;
        const float v = xs[iy * 10 + ix];
        const f32x4 w = w1r[ky * 3 + kx];
#pragma unroll
        for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, w[c], acc[c]);
      }
    }
#pragma unroll
    for (int c = 0; c < 4; ++c) acc[c] = e3_act(acc[c], p.act1);
    *reinterpret_cast<f32x4*>(A1 + (g * 25 + pix) * E3_P1 + (tid & 15) * 4) = acc;
  }
  __syncthreads();
  E3_STAMP();

  // ---- conv2d_1: 3x3 stride 1 pad 1, 64 -> 128.  D[channel 16][pixel 16] per MFMA; wave = channel tile, five pixel tiles.
  // MFMA m = tap*16 + q*4 + j multiplies channels ci = 16 q + 4 kg + j of tap `tap` (host-permuted A fragments, four MFMAs per
  // 16-byte load of either operand).
  {
    int brow[5][9];
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int px = t * 16 + col, g = px / 25, r = px - g * 25, y = r / 5, x = r - y * 5;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        const bool ok = px < E3_G * 25 && (unsigned)iy < 5u && (unsigned)ix < 5u;
        brow[t][tap] = (ok ? g * 25 + iy * 5 + ix : E3_ZROW) * E3_P1 + kg * 4;
      }
    }
    f32x4 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int tq = 0; tq < 36; ++tq) {       // (tap, q): 16 input channels of one tap; fully unrolled: brow[][] stays in registers
      const int tap = tq >> 2, q = tq & 3;
      const f32x4 af = w2r[tq];
#pragma unroll
      for (int t = 0; t < 5; ++t) {
        const f32x4 bf = *reinterpret_cast<const f32x4*>(A1 + brow[t][tap] + q * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[j], bf[j], acc[t], 0, 0, 0);
      }
    }
    E3_STAMP();
    const f32x4 b2 = *reinterpret_cast<const f32x4*>(p.b2 + wave * 16 + kg * 4);
#pragma unroll
    for (int t = 0; t < 5; ++t) {
      const int px = t * 16 + col, g = px / 25, r = px - g * 25;
      f32x4 v = acc[t] + b2;
#pragma unroll
      for (int c = 0; c < 4; ++c) v[c] = e3_act(v[c], p.act2);
      // flattened NHWC sample (sr-ae-conv.ipynb:c166: index (h*5+w)*128 + c)
      if (px < E3_G * 25) *reinterpret_cast<f32x4*>(A2 + g * E3_P2 + r * 128 + wave * 16 + kg * 4) = v;
    }
  }
  __syncthreads();

  E3_STAMP();
  // ---- dense 3200 -> 128: thread = 4 features x K slice of 200 (16 slices), all E3_G samples ----
  {
    const int f4 = (tid & 31) * 4, ks = tid >> 5;
    const float* wp = p.wd + (size_t)(ks * 200) * 128 + f4;
    // A thread's 200 rows are five sub-slices of 40 with an accumulator set each, summed in sub-slice order at the end -- so the
    // ORDER in which a workgroup walks them is free, and workgroup b starts at sub-slice b % 5: in lockstep all 256 workgroups
    // would read the same addresses at the same time and queue on the same few L2 channels (as in enc16: 2x on this phase).
    // Weight rows arrive 8 at a time, the next 8 in flight while these are used (two register sets, swapped by name).
    constexpr int CH = 8, SUB = 5, CPS = 40 / CH;     // chunk rows, sub-slices, chunks per sub-slice
    const int rot = __builtin_amdgcn_readfirstlane(blockIdx.x % SUB);
    auto row0_of = [&](int c) {                       // first row (in the thread's slice) of the c-th chunk walked
      int sub = c / CPS + rot;
      sub = sub >= SUB ? sub - SUB : sub;
      return sub * 40 + (c % CPS) * CH;
    };
    f32x4 sacc[SUB][E3_G];                            // indexed by walk slot (compile time); slot s holds sub-slice (s + rot) % 5
#pragma unroll
    for (int sl = 0; sl < SUB; ++sl)
#pragma unroll
      for (int g = 0; g < E3_G; ++g) sacc[sl][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 wa[CH], wb[CH];
    {
      const int r0 = row0_of(0);
#pragma unroll
      for (int i = 0; i < CH; ++i) wa[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)(r0 + i) * 128);
    }
#pragma unroll
    for (int c = 0; c < SUB * CPS; ++c) {
      if (c + 1 < SUB * CPS) {
        const int r1 = row0_of(c + 1);
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          if (c & 1) wa[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)(r1 + i) * 128);
          else wb[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)(r1 + i) * 128);
        }
      }
      const int r0 = row0_of(c);
#pragma unroll
      for (int g = 0; g < E3_G; ++g) {
#pragma unroll
        for (int h4 = 0; h4 < CH / 4; ++h4) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(A2 + g * E3_P2 + ks * 200 + r0 + h4 * 4);
#pragma unroll
          for (int kk = 0; kk < 4; ++kk) {
            const f32x4 w = (c & 1) ? wb[h4 * 4 + kk] : wa[h4 * 4 + kk];
            sacc[c / CPS][g] = __builtin_elementwise_fma(f32x4{xv[kk], xv[kk], xv[kk], xv[kk]}, w, sacc[c / CPS][g]);
          }
        }
      }
    }
    // sub-slice k was walked in slot (k - rot) mod 5 (rot is wave-uniform); add the sub-slices in their own order
    f32x4 acc[E3_G];
#pragma unroll
    for (int g = 0; g < E3_G; ++g) {
      f32x4 part[SUB];
#pragma unroll
      for (int k = 0; k < SUB; ++k) {
        part[k] = sacc[0][g];
#pragma unroll
        for (int sl = 1; sl < SUB; ++sl) {
          const bool is = (sl + rot) % SUB == k;
#pragma unroll
          for (int r = 0; r < 4; ++r) part[k][r] = is ? sacc[sl][g][r] : part[k][r];
        }
      }
      acc[g] = ((part[0] + part[1]) + (part[2] + part[3])) + part[4];
    }
#pragma unroll
    for (int g = 0; g < E3_G; ++g) *reinterpret_cast<f32x4*>(PS + (ks * E3_G + g) * 128 + f4) = acc[g];
  }
  __syncthreads();
  if (tid < E3_G * 128) {
    const int g = tid >> 7, f = tid & 127;
    float s = PS[g * 128 + f];
#pragma unroll
    for (int ks = 1; ks < 16; ++ks) s += PS[(ks * E3_G + g) * 128 + f];     // slice order
    A3[g * 128 + f] = e3_act(s + p.bd[f], p.act3);
  }
  __syncthreads();

  E3_STAMP();
  // ---- latent_vector 128 -> NL ----
  for (int i = tid; i < E3_G * p.nl; i += E3_NTHR) {
    const int g = i / p.nl, o = i - g * p.nl;
    // four interleaved partial sums, 32 independent loads in flight: as one dependent chain of 128 loads this was 33k cycles
    float s4[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
    for (int k = 0; k < 128; k += 4) {
#pragma unroll
      for (int j = 0; j < 4; ++j) s4[j] = fmaf(A3[g * 128 + k + j], p.wl[(size_t)(k + j) * p.nl_pad + o], s4[j]);
    }
    const float sum = (s4[0] + s4[1]) + (s4[2] + s4[3]);
    if (g < gv) p.z[(size_t)(s0 + g) * p.nl + o] = e3_act(sum + p.bl[o], p.act4);
  }
#ifdef SRCFD_DIAG
  E3_STAMP();
  if (blockIdx.x == 7 && lane == 0)
    for (int i = 0; i < 7; ++i) e3_prof[wave * 8 + i] = stamp[i] - stamp[0];
#endif
}

hipError_t launch_enc32(const Enc32Params& p, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  static thread_local int attr_dev = -1;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  if (attr_dev != dev) {
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(enc32), hipFuncAttributeMaxDynamicSharedMemorySize, E3_LDS);
    if (e != hipSuccess) return e;
    attr_dev = dev;
  }
  hipLaunchKernelGGL(enc32, dim3((p.n + E3_G - 1) / E3_G), dim3(E3_NTHR), E3_LDS, s, p);
#ifdef SRCFD_DIAG
  static const bool prof = getenv("SRCFD_ENC32_PROF") != nullptr;
  static int calls = 0;
  if (prof && ++calls == 12) {
    unsigned long long h[64];
    (void)hipStreamSynchronize(s);
    if (hipMemcpyFromSymbol(h, HIP_SYMBOL(e3_prof), sizeof(h)) == hipSuccess) {
      fprintf(stderr, "enc32 workgroup 7, cycles since entry: staged, conv2d, conv2d_1 MFMA, A2 ready, dense + reduce, end\n");
      for (int w = 0; w < 8; ++w) fprintf(stderr, "  wave %d: %6llu %6llu %6llu %6llu %6llu %6llu\n", w, h[w * 8 + 1], h[w * 8 + 2], h[w * 8 + 3], h[w * 8 + 4], h[w * 8 + 5], h[w * 8 + 6]);
    }
  }
#endif
  return hipGetLastError();
}

}  // namespace srcfd
