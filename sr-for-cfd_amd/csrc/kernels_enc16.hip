// enc16: the whole encoder_10 in one launch, 16-bit operands:
//   standardise -> conv2d (3x3 s2, 1->64, VALU) -> conv2d_1 (3x3 s1, 64->128, MFMA 32x32x16) -> flatten + dense
//   (3200->128, MFMA 16x16x32) -> latent_vector (128->64 padded, MFMA 16x16x32) -> z (n,64)
//
// Why: as four launches (+ a split-K finish) these layers cost 0.060 ms per 768-sample batch for 3.6 GFLOP -- every one of
// them a latency-bound round trip of a few hundred kB through HBM with three samples' worth of work per CU.  Here a
// workgroup owns E_G = 5 samples (125 conv pixels = 4 MFMA column tiles, 98 % full) from the 10x10 input to the latent
// vector; activations live in LDS (53 KB) and the only operand traffic is the weights (1 MB per workgroup, L2 hits after
// the first workgroups), read as ready-made MFMA fragments -- one coalesced 16-byte load per lane and MFMA, no LDS staging.
// Every output column (pixel / sample) of an MFMA accumulates over k in a fixed order, independent of its neighbours:
// results do not depend on the batch size or on a sample's position in the batch.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev16.h"
#include "kernels16.h"

namespace srcfd {

constexpr int E_G = ENC_G;                 // samples per workgroup
constexpr int E_NTHR = 512;
constexpr int E_P1 = 72;                   // conv2d-output row pitch, elements (64 + 8: ds_read_b128 lane groups hit 16 distinct slots)
constexpr int E_ZROW = E_G * 25;           // all-zero row: taps outside the 5x5 image, pixels past the last sample
constexpr int E_P2 = 3200 + 8;             // conv2d_1-output pitch per sample, elements (6416 B = 16 mod 256)
constexpr int E_P3 = 128 + 8;              // dense-output pitch per sample
constexpr int E_OFF_A1 = 2048;                                   // after x: E_G * 100 f32
constexpr int E_OFF_A2 = E_OFF_A1 + (E_ZROW + 1) * E_P1 * 2;
constexpr int E_OFF_A3 = E_OFF_A2 + E_G * E_P2 * 2;
constexpr int E_LDS = E_OFF_A3 + 16 * E_P3 * 2;                  // 16 rows: lanes of absent samples read rows of their own
static_assert(E_G * 100 * 4 <= E_OFF_A1 && E_G * 25 <= 128 && E_G <= 16, "workgroup shape");

typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool F16>
__global__ void __launch_bounds__(E_NTHR, 1) enc16(EncParams p) {
  extern __shared__ __attribute__((aligned(16))) char esm[];
  float* X0 = reinterpret_cast<float*>(esm);
  uint16_t* A1 = reinterpret_cast<uint16_t*>(esm + E_OFF_A1);
  uint16_t* A2 = reinterpret_cast<uint16_t*>(esm + E_OFF_A2);
  uint16_t* A3 = reinterpret_cast<uint16_t*>(esm + E_OFF_A3);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int s0 = blockIdx.x * E_G, gv = min(E_G, p.n - s0);
#ifdef SRCFD_DIAG
  unsigned long long stamp[8];
  int nstamp = 0;
#define ENC_STAMP() stamp[nstamp++] = __builtin_readcyclecounter()
#else
#define ENC_STAMP()
#endif
  ENC_STAMP();

  // The small loads the first phases wait for go first and alone: this thread's input elements and its conv2d weights (it
  // always works on channels 4 (tid % 16) .. +3: 512 % 16 == 0).  The 36 KB per wave of conv2d_1 fragments are issued only
  // after the input is staged: a CU's texture path takes 16 cycles per 1 KB load instruction, and with 8 waves x 36 of them
  // queued ahead the first barrier was reached after 9k cycles (section timers, -DSRCFD_DIAG) instead of ~2k.
  constexpr int XJ = (E_G * 100 + E_NTHR - 1) / E_NTHR;
  float xv[XJ], xmean[XJ], xsd[XJ];
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int i = tid + j * E_NTHR, g = i / 100;
    const bool ok = i < E_G * 100 && g < gv;
    xv[j] = ok ? p.x[(size_t)s0 * 100 + i] : 0.f;
    xmean[j] = (ok && p.affine) ? p.affine[2 * (s0 + g)] : 0.f;
    xsd[j] = (ok && p.affine) ? p.affine[2 * (s0 + g) + 1] : 1.f;
  }
  float4 w1r[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) w1r[t] = *reinterpret_cast<const float4*>(p.w1 + t * 64 + (tid & 15) * 4);
  const float4 b1r = *reinterpret_cast<const float4*>(p.b1 + (tid & 15) * 4);
  // ---- input: standardise_with_stats (PyCFD_ML_accelerated.py:665-668), same expression as enc_conv1_16 ----
#pragma unroll
  for (int j = 0; j < XJ; ++j) {
    const int i = tid + j * E_NTHR, g = i / 100;
    if (i < E_G * 100) {
      float v = xv[j];
      if (p.affine && g < gv) {
        float sd = xsd[j];
        if (sd == 0.f) sd = 1e-8f;
        v = __fdiv_rn(__fsub_rn(v, xmean[j]), sd);
      }
      X0[i] = v;
    }
  }
  if (tid < E_P1 / 2) reinterpret_cast<uint32_t*>(A1 + E_ZROW * E_P1)[tid] = 0;
  __syncthreads();
  ENC_STAMP();

  // ---- conv2d: 3x3 stride 2, TF SAME (pad 0 before / 1 after), 1 -> 64, swish; one item = 4 channels of one pixel.
  // conv2d_1's A operands (all 36 k-steps of this wave's 32-channel tile) are issued nine per item, between the vector work:
  // issued in one burst they hold every wave at the texture path's queue for ~4.6k cycles before conv2d can start.
  const int mt = wave & 3, ntb = (wave >> 2) * 2;
  const u32x4* w2 = reinterpret_cast<const u32x4*>(p.w2f) + (size_t)mt * 36 * 64 + lane;
  u32x4 wf[36];
  constexpr int C1_ITEMS = (E_G * 25 * 16 + E_NTHR - 1) / E_NTHR;
  static_assert(C1_ITEMS * 9 >= 36, "conv2d items carry the 36 fragment loads");
#pragma unroll
  for (int it = 0; it < C1_ITEMS; ++it) {
#pragma unroll
    for (int ks = it * 9; ks < it * 9 + 9 && ks < 36; ++ks) wf[ks] = w2[ks * 64];
    __builtin_amdgcn_sched_barrier(0);
    const int idx = tid + it * E_NTHR;
    if (idx >= E_G * 25 * 16) continue;
    const int c4 = idx & 15 /* == tid & 15 */, pix = (idx >> 4) % 25, g = idx / (25 * 16);
    const int oy = pix / 5, ox = pix - oy * 5;
    float acc[4] = {b1r.x, b1r.y, b1r.z, b1r.w};
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
        const float4 w = w1r[ky * 3 + kx];
        acc[0] = fmaf(v, w.x, acc[0]); acc[1] = fmaf(v, w.y, acc[1]); acc[2] = fmaf(v, w.z, acc[2]); acc[3] = fmaf(v, w.w, acc[3]);
      }
    }
    uint2 o;
    o.x = pack2<F16>(swish_scaled(acc[0]), swish_scaled(acc[1]));
    o.y = pack2<F16>(swish_scaled(acc[2]), swish_scaled(acc[3]));
    *reinterpret_cast<uint2*>(A1 + (g * 25 + pix) * E_P1 + c4 * 4) = o;
  }
  __syncthreads();
  ENC_STAMP();

  // ---- conv2d_1: 3x3 stride 1 pad 1, 64 -> 128, swish.  D[channel][pixel]: A = weight fragments (registers), B = the
  // pixel's 8 consecutive input channels of tap ks / 4, gathered from A1 (zero row outside the image).  Wave = channel
  // tile mt x pixel tiles ntb, ntb + 1.
  {
    int brow[2][9];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int px = (ntb + t) * 32 + l31, g = px / 25, r = px - g * 25, y = r / 5, x = r - y * 5;
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const int iy = y + tap / 3 - 1, ix = x + tap % 3 - 1;
        const bool ok = px < E_G * 25 && (unsigned)iy < 5u && (unsigned)ix < 5u;
        brow[t][tap] = (ok ? g * 25 + iy * 5 + ix : E_ZROW) * E_P1 + h * 8;
      }
    }
    f32x16 acc[2];
    acc[0] = load_bias16(reinterpret_cast<const char*>(p.b2f) + (mt * 2 + h) * 64);
    acc[1] = acc[0];
#pragma unroll
    for (int ks = 0; ks < 36; ++ks) {
      const uint4 af = __builtin_bit_cast(uint4, wf[ks]);
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const uint4 bf = *reinterpret_cast<const uint4*>(A1 + brow[t][ks >> 2] + (ks & 3) * 16);
        acc[t] = mfma32<F16>(af, bf, acc[t]);
      }
    }
    ENC_STAMP();
    // swish, pack, store as the flattened NHWC sample (sr-ae-conv.ipynb:c166: index (h*5+w)*128 + c)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      uint32_t o[8];
      swish_pack16<F16>(acc[t], o);
      const int px = (ntb + t) * 32 + l31, g = px / 25, r = px - g * 25;
      if (px < E_G * 25) {
        uint16_t* dst = A2 + g * E_P2 + r * 128 + mt * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(dst + 8 * q) = make_uint2(o[2 * q], o[2 * q + 1]);
      }
    }
  }

  // ---- dense: 3200 -> 128, swish.  16x16x32 MFMAs: wave = 16 output features over the whole K (no cross-wave sum);
  // A = weight fragments streamed from memory 20 k-steps ahead, B = the sample's 8 consecutive k from A2.
  // Lanes of absent samples (column >= E_G) re-read sample E_G - 1: their columns are never stored.
  const int col = lane & 15, kg = lane >> 4;
  u32x4 wlr[4] = {};   // latent_vector's operands (waves 0-3): loaded with the dense layer's first chunk, consumed after it
  {
    constexpr int CH = 20, NCH = 100 / CH;
    static_assert(NCH == 5, "the rotation and the final sum below are written for five chunks");
    // All workgroups start together and read the same addresses: in lockstep they would queue on the same few L2 channels.
    // Workgroup b walks the five K chunks starting from chunk b % 5.  Each chunk has its own accumulator and the five are
    // summed in chunk order at the end, so the result does not depend on the walk (nor on the workgroup a sample lands in).
    const int rot = __builtin_amdgcn_readfirstlane(blockIdx.x % NCH);
    const u32x4* wd = reinterpret_cast<const u32x4*>(p.wdf) + (size_t)wave * 100 * 64 + lane;
    auto chunk_of = [&](int slot) { const int c = slot + rot; return c >= NCH ? c - NCH : c; };
    u32x4 da[CH], db[CH];
    {
      const u32x4* src = wd + chunk_of(0) * CH * 64;
#pragma unroll
      for (int i = 0; i < CH; ++i) da[i] = src[i * 64];
    }
    if (wave < 4) {
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) wlr[ks] = (reinterpret_cast<const u32x4*>(p.wlf) + (size_t)wave * 4 * 64 + lane)[ks * 64];
    }
    __syncthreads();  // A2 complete
    ENC_STAMP();
    const float4 bias = *reinterpret_cast<const float4*>(p.bd + wave * 16 + kg * 4);
    // one accumulator per 640-deep chunk of K (a single 3200-term f32 chain would also carry ~3x the rounding error of the
    // split-K slabs this kernel replaces)
    f32x4 slot_acc[NCH];
#pragma unroll
    for (int c = 0; c < NCH; ++c) slot_acc[c] = f32x4{0.f, 0.f, 0.f, 0.f};
    const uint16_t* bcol = A2 + min(col, E_G - 1) * E_P2 + kg * 8;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      // the chunk after this one goes in flight before this one is consumed (two register sets, swapped by name)
      if (c + 1 < NCH) {
        const u32x4* src = wd + chunk_of(c + 1) * CH * 64;
#pragma unroll
        for (int i = 0; i < CH; ++i) {
          if (c & 1) da[i] = src[i * 64];
          else db[i] = src[i * 64];
        }
      }
      const uint16_t* bsrc = bcol + chunk_of(c) * CH * 32;
#pragma unroll
      for (int i = 0; i < CH; ++i) {
        const uint4 bf = *reinterpret_cast<const uint4*>(bsrc + i * 32);
        slot_acc[c] = mfma16<F16>(__builtin_bit_cast(uint4, (c & 1) ? db[i] : da[i]), bf, slot_acc[c]);
      }
    }
    // chunk k was walked in slot (k - rot) mod 5; rot is wave-uniform
    f32x4 part[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      part[k] = slot_acc[0];
#pragma unroll
      for (int sl = 1; sl < NCH; ++sl) {
        const bool is = (sl + rot) % NCH == k;
#pragma unroll
        for (int r = 0; r < 4; ++r) part[k][r] = is ? slot_acc[sl][r] : part[k][r];
      }
    }
    f32x4 acc = ((part[0] + part[1]) + (part[2] + part[3])) + part[4];
    acc += f32x4{bias.x, bias.y, bias.z, bias.w};
    const uint2 o = make_uint2(pack2<F16>(act16(acc[0], p.act_dense), act16(acc[1], p.act_dense)),
                               pack2<F16>(act16(acc[2], p.act_dense), act16(acc[3], p.act_dense)));
    *reinterpret_cast<uint2*>(A3 + col * E_P3 + wave * 16 + kg * 4) = o;
  }
  __syncthreads();
  ENC_STAMP();

  // ---- latent_vector: 128 -> 64 (50 zero-padded), waves 0-3 ----
  if (wave < 4) {
    const float4 bias = *reinterpret_cast<const float4*>(p.bl + wave * 16 + kg * 4);
    f32x4 acc = {bias.x, bias.y, bias.z, bias.w};
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const uint4 bf = *reinterpret_cast<const uint4*>(A3 + col * E_P3 + ks * 32 + kg * 8);
      acc = mfma16<F16>(__builtin_bit_cast(uint4, wlr[ks]), bf, acc);
    }
    if (col < gv) {
      const uint2 o = make_uint2(pack2<F16>(act16(acc[0], p.act_latent), act16(acc[1], p.act_latent)),
                                 pack2<F16>(act16(acc[2], p.act_latent), act16(acc[3], p.act_latent)));
      *reinterpret_cast<uint2*>(p.z + (size_t)(s0 + col) * 64 + wave * 16 + kg * 4) = o;
    }
  }
#ifdef SRCFD_DIAG
  ENC_STAMP();
  if (p.prof && blockIdx.x == 7 && lane == 0)
    for (int i = 0; i < 7; ++i) p.prof[wave * 8 + i] = stamp[i] - stamp[0];
#endif
}

hipError_t launch_enc16(bool f16, const EncParams& p, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  void (*fn)(EncParams) = f16 ? enc16<true> : enc16<false>;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(fn), E_LDS);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3((p.n + E_G - 1) / E_G), dim3(E_NTHR), E_LDS, s, p);
  return hipGetLastError();
}

}  // namespace srcfd

namespace srcfd {

// ---------------------------------------------------------------------------
// dense1_16: the decoder's first Dense layer (latent 64 -> 36 864, swish) as its own kernel.
// K is one 64-deep tile, so the generic implicit GEMM spent its time in per-workgroup latency chains (1 728 workgroups:
// stage two tiles through LDS, 16 MFMAs, stage the result, store) -- 0.031 ms for 3.6 GFLOP and a 57 MB write.
// Here a workgroup owns 144 output features (36 864 = 256 x 144: one workgroup per CU, one round) for ALL samples: its
// weights (18 KB) sit in registers as 16x16x32 A fragments for the whole kernel, each wave walks 16-sample tiles
// (B = the latent vectors, read straight from memory / L2), and the swished tile leaves through a wave-private LDS
// transpose as 288-byte runs per sample.  No workgroup barrier anywhere.
// ---------------------------------------------------------------------------
constexpr int D1_FT = 9, D1_NF = D1_FT * 16;      // feature tiles / features per workgroup
constexpr int D1_PITCH = D1_NF * 2 + 16;          // LDS row pitch of the transpose tile, bytes
constexpr int D1_WAVES = 8;
constexpr int D1_LDS = D1_WAVES * 16 * D1_PITCH;

template <bool F16>
__global__ void __launch_bounds__(64 * D1_WAVES, 1) dense1_16(const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt,
                                                              const float* __restrict__ bias, uint16_t* __restrict__ Y, int M, int N, int act) {
  extern __shared__ __attribute__((aligned(16))) char dsm[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, col = lane & 15, kg = lane >> 4;
  const int n0 = blockIdx.x * D1_NF;
  u32x4 wa[D1_FT][2];
  float4 bs[D1_FT];
#pragma unroll
  for (int ft = 0; ft < D1_FT; ++ft) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) wa[ft][ks] = *reinterpret_cast<const u32x4*>(Wt + (size_t)(n0 + ft * 16 + col) * 64 + ks * 32 + kg * 8);
    bs[ft] = *reinterpret_cast<const float4*>(bias + n0 + ft * 16 + kg * 4);
  }
  char* st = dsm + wave * 16 * D1_PITCH;
  const int tiles = (M + 15) / 16;
  u32x4 xb[2], xn[2];
  auto load_x = [&](int t, u32x4 (&dst)[2]) {
    const int sv = min(t * 16 + col, M - 1);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) dst[ks] = *reinterpret_cast<const u32x4*>(X + (size_t)sv * 64 + ks * 32 + kg * 8);
  };
  if (wave < tiles) load_x(wave, xb);
  for (int t = wave; t < tiles; t += D1_WAVES) {
    if (t + D1_WAVES < tiles) load_x(t + D1_WAVES, xn);
#pragma unroll
    for (int ft = 0; ft < D1_FT; ++ft) {
      f32x4 acc = {bs[ft].x, bs[ft].y, bs[ft].z, bs[ft].w};
      acc = mfma16<F16>(__builtin_bit_cast(uint4, wa[ft][0]), __builtin_bit_cast(uint4, xb[0]), acc);
      acc = mfma16<F16>(__builtin_bit_cast(uint4, wa[ft][1]), __builtin_bit_cast(uint4, xb[1]), acc);
      const uint2 o = make_uint2(pack2<F16>(act16(acc[0], act), act16(acc[1], act)), pack2<F16>(act16(acc[2], act), act16(acc[3], act)));
      *reinterpret_cast<uint2*>(st + col * D1_PITCH + ft * 32 + kg * 8) = o;
    }
    // the tile is wave-private: lanes read chunks other lanes of the SAME wave wrote.  The hardware executes one wave's LDS
    // operations in order; the wave-scope fence + barrier keep the compiler from moving these loads above the stores (and, below,
    // the next tile's stores above these loads) -- no instruction is emitted for either
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // write-out: 16 samples x 18 chunks of 16 B, consecutive lanes on consecutive chunks of one sample's 288-byte run
#pragma unroll
    for (int r = 0; r < (16 * D1_NF / 8 + 63) / 64; ++r) {
      const int c = lane + 64 * r, row = c / (D1_NF / 8), ch = c - row * (D1_NF / 8), s = t * 16 + row;
      if (c < 16 * D1_NF / 8 && s < M) {
        const uint4 v = *reinterpret_cast<const uint4*>(st + row * D1_PITCH + ch * 16);
        *reinterpret_cast<uint4*>(Y + (size_t)s * N + n0 + ch * 8) = v;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    xb[0] = xn[0]; xb[1] = xn[1];
  }
}

bool dense1_16_qualifies(const GemmDesc& d, int Kpad) {
  return d.MH == 1 && d.MW == 1 && d.K == 64 && Kpad == 64 && d.N == d.Npad && d.N % D1_NF == 0 && d.OC == d.N && d.CI == 64;
}

hipError_t launch_dense1_16(bool f16, const GemmDesc& d, const uint16_t* X, const uint16_t* Wt, const float* bias, uint16_t* Y, hipStream_t s) {
  if (d.M == 0) return hipSuccess;
  void (*fn)(const uint16_t*, const uint16_t*, const float*, uint16_t*, int, int, int) = f16 ? dense1_16<true> : dense1_16<false>;
  hipLaunchKernelGGL(fn, dim3(d.N / D1_NF), dim3(64 * D1_WAVES), D1_LDS, s, X, Wt, bias, Y, d.M, d.N, d.act);
  return hipGetLastError();
}

}  // namespace srcfd
