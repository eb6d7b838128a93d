// tail32: the f32 (parity, <= 1e-5) path's last four layers as ONE streaming kernel (gfx950 only):
//   ConvT 2x2 s2 64->32 -> ConvT 2x2 s2 32->16 -> ConvT 2x2 s2 16->8 -> Conv 3x3 SAME 8->1
//   + de-standardise + NaN/Inf guard + output cast              (sr-ae-conv.ipynb:c283-286)
//
// Why: as separate launches the 8-channel 400x400 activation (5.12 MB per sample in f32) is written by one kernel and
// re-read by the next -- 2 x 3.9 GB per 768 samples, 3.4 of the f32 path's 5.6 ms.  Here nothing between the 50x50x64
// input and the final image touches HBM.
//
// Structure (the streaming-ring idea of tail16, re-done for f32 operands):
//   * one 512-thread workgroup per CU walks its samples top to bottom in STRIPS of 4 output rows
//     (= one first-layer tap row ty1 of one 50-level row); per round
//       P: every wave takes one item = (16-pixel tile of the 50-level row, first-layer tap column tx1) and runs the
//          three transposed convolutions in registers on v_mfma_f32_16x16x4_f32 (exact f32 products): rows = tap x
//          channel, columns = the wave's 16 pixels, so the activated accumulators of one layer ARE the B operands of the
//          next (lane (pixel n, group rg), register i holds channel 4 rg + i: k-step i of the next layer contracts
//          channels {4 kg + i}, weights packed in that order on the host).  The 4x4x8 output block of every pixel goes
//          to a 10-row LDS ring of the 400-level (f32, 32 B per pixel);
//       D: the 3x3 output conv of the PREVIOUS strip's rows from the ring on the VALU (v_pk_fma_f32 runs at the f32
//          MFMA rate, and the matrix pipe is busy with P), 4 adjacent pixels per lane, + de-standardise + guard,
//          stored as whole 16-byte pieces of contiguous rows;
//     waves 0-3 run P then D, waves 4-7 D then P, so that the two waves of a SIMD are in different phases
//     (matrix work of one beside vector work of the other); one LDS-only barrier per round.
//   * ring layout [row][channel half][x & 7 plane][x >> 3 slot] of 16-byte granules, 54 slots per plane (slot 0 and
//     slots past the image stay zero = the SAME padding of the conv): P writes 16 consecutive slots per instruction, D
//     reads consecutive x across lanes -> both conflict-free; one extra all-zero row stands in for rows outside the image.
//   * small batches: a sample is cut into S segments of consecutive strips ("virtual samples"), each preceded by one
//     warm-up strip that refills the two ring rows its first output row reads; every output pixel goes through the same
//     arithmetic whatever S is (results are bit-identical across batch sizes).
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>

#include "act_device.h"
#include "kernels.h"
#include "kernels16.h"  // lds_attr_once

namespace srcfd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int R_SLOTS = 54;                       // granules per plane: 1 + 50 + padding (plane stride 864 B = 96 mod 256)
constexpr int R_PLANE = R_SLOTS * 16;             // 864
constexpr int R_HALF = 8 * R_PLANE;               // 6912: channels 0-3 | 4-7
constexpr int R_ROW = 2 * R_HALF;                 // 13824
constexpr int R_ROWS = 10;
constexpr int R_ZROW = R_ROWS * R_ROW;            // byte offset of the all-zero row
constexpr int R_WC = (R_ROWS + 1) * R_ROW;        // output-conv weights [9 taps][8] f32 + bias (wave-uniform broadcast reads)
constexpr int R_W2 = R_WC + 320;                  // second / third layer A operands (8 KB + 2 KB): one conflict-free ds_read_b32 per MFMA
constexpr int R_W3 = R_W2 + 4 * 8 * 64 * 4;       // (held in registers they cost 40 of the 256 a wave has: the kernel spilled)
constexpr int R_DUMP = R_W3 + 2 * 4 * 64 * 4;     // 64 granules: where lanes / rounds with nothing to store write (keeps the body branch-free)
constexpr int T32_LDS = R_DUMP + 64 * 16;         // 163648
static_assert(T32_LDS <= 160 * 1024, "tail32 LDS budget");
// X3 layout: no all-zero row (the output conv skips window rows outside the image instead of reading zeros: 13.8 KB), which makes
// room for the second layer's THREE bf16 weight planes (12 KB instead of 8 KB of f32 fragments)
template <bool X3> struct T32Lay {
  static constexpr int ZROW = X3 ? -1 : R_ZROW;                       // "no row": X3 skips it
  static constexpr int WC = X3 ? R_ROWS * R_ROW : R_WC;
  static constexpr int W2 = WC + 320;
  static constexpr int W3 = W2 + (X3 ? 4 * 3 * 64 * 16 : 4 * 8 * 64 * 4);
  static constexpr int DUMP = W3 + 2 * 4 * 64 * 4;
  static constexpr int LDS = DUMP + 64 * 16;
};
static_assert(T32Lay<true>::LDS <= 160 * 1024 && T32Lay<false>::LDS == T32_LDS, "tail32 LDS budget");

__device__ __forceinline__ void lds_barrier32() {  // LDS traffic complete, global loads / stores stay in flight
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

// The three transposed convolutions are swish, the output conv linear (decoder_400; the host falls back to the
// layer-by-layer launches for anything else).  Every swish layer produces u = log2(e) x (first-layer weights and all biases
// scaled on the host) and the epilogue computes u * rcp(1 + exp2(-u)) = log2(e) swish(x): one v_exp_f32 with a negated
// source, one add, one v_rcp_f32, one multiply.  Between swish layers the factors cancel (weights unscaled); the output
// conv's weights absorb the last 1/log2(e).  exp2 and rcp are good to ~1 ulp each, an order of magnitude inside the 1e-5
// bar; for u -> -inf: exp2 = inf, rcp = 0, u * 0 = -0 (no NaN); a NaN stays a NaN.
// On this chip f32 MFMAs and vector instructions of the waves of a SIMD do NOT overlap (tools/microbench8.hip: their times
// add), so every vector instruction dropped here is kernel time: the Newton-refined sigmoid of the layer-by-layer path
// costs 7 instructions per activation, this 4 (3 with the packed add / multiply the compiler forms).
// (`one2` = {1, 1} in a register pair the compiler cannot see through: with a literal 1.0f it forms four v_add_f32 instead of two
// v_pk_add_f32 -- 28 of a round's ~670 vector instructions.)
__device__ __forceinline__ f32x4 swish_l2e(f32x4 u, const f32x2& one2) {
  f32x2 e0, e1;
  e0[0] = __builtin_amdgcn_exp2f(-u[0]); e0[1] = __builtin_amdgcn_exp2f(-u[1]);
  e1[0] = __builtin_amdgcn_exp2f(-u[2]); e1[1] = __builtin_amdgcn_exp2f(-u[3]);
  e0 = e0 + one2; e1 = e1 + one2;
  f32x4 r;
  r[0] = __builtin_amdgcn_rcpf(e0[0]); r[1] = __builtin_amdgcn_rcpf(e0[1]);
  r[2] = __builtin_amdgcn_rcpf(e1[0]); r[3] = __builtin_amdgcn_rcpf(e1[1]);
  return u * r;
}

// One strip of one (virtual) sample as the pipeline sees it.
struct Strip {
  int valid;   // 0: nothing (before the first / after the last job)
  int smp;     // real sample
  int g;       // strip index in the sample, 0 .. 2H-1 (row y = g >> 1, first-layer tap row ty1 = g & 1)
  int warm;    // warm-up strip of a segment: computed for the ring only, its rows belong to the previous segment
};

// TRAIN (train.hip, forward pass of a training step): the epilogue turns the image into the loss gradient instead of
// de-standardising it -- out = dpred = two_scale * (pred - target), and every workgroup leaves the sum of its squared
// errors (float64, fixed order: the row -> workgroup assignment is static) in sse_partial[blockIdx.x].
// X3 (SRCFD_PREC_FP32X3): the first TWO layers (ConvT 64 -> 32 -> 16: two thirds of the kernel's f32 MFMA cycles) as six bf16 MFMAs per product
// on operands split exactly into three bf16 terms, as in kernels_x3.hip: v_mfma_f32_16x16x32_bf16, K = 64 = two k-steps, the hi x hi
// products in an accumulator of their own.  The input tile is split once per round (16 values per lane); the hi and mid weight
// planes of both tap rows are resident (64 registers, what the f32 fragments took), the lo plane of the round's tap row is
// fetched at the start of the round and used last.  bf16 MFMAs run under the vector work of the wave that issues them, f32 ones
// do not: the layer leaves the f32 matrix pipe (1 024 cycles per item) for ~110 vector instructions.
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef short s16x8_t32 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ f32x4 mfma16bf(const u32x4& a, const u32x4& b, const f32x4& c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8_t32, a), __builtin_bit_cast(s16x8_t32, b), c, 0, 0, 0);
}
__device__ __forceinline__ void split8_t32(const f32x4& v0, const f32x4& v1, u32x4& fh, u32x4& fm, u32x4& fl) {
  // two values at a time: the two exact subtractions are packed (v_pk_add_f32 with a negated source)
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x2 x = q < 2 ? f32x2{v0[2 * q], v0[2 * q + 1]} : f32x2{v1[2 * q - 4], v1[2 * q - 3]};
    const u32x2 hb = __builtin_bit_cast(u32x2, x) & 0xffff0000u;
    const f32x2 r1 = x - __builtin_bit_cast(f32x2, hb);
    const u32x2 mb = __builtin_bit_cast(u32x2, r1) & 0xffff0000u;
    const u32x2 lb = __builtin_bit_cast(u32x2, r1 - __builtin_bit_cast(f32x2, mb));
    fh[q] = __builtin_amdgcn_perm(hb[1], hb[0], 0x07060302u);
    fm[q] = __builtin_amdgcn_perm(mb[1], mb[0], 0x07060302u);
    fl[q] = __builtin_amdgcn_perm(lb[1], lb[0], 0x07060302u);
  }
}

// Diagnostic work-skipping switches (Tail32Params::ablate) exist only in a -DSRCFD_DIAG build (make DIAG=1).
#ifdef SRCFD_DIAG
#define T32_ABL(bit) (p.ablate & (bit))
#else
#define T32_ABL(bit) 0
#endif

template <int OUT, bool TRAIN, bool X3 = false>  // OUT: 0 f32, 1 bf16, 2 f16
__global__ void __launch_bounds__(512) tail32(Tail32Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, n = lane & 15, rg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.H, W = p.W, SH = 2 * H, OW = 8 * W, OHs = 8 * H;

  f32x2 one2 = {1.0f, 1.0f};
  asm volatile("" : "+v"(one2));
  typedef T32Lay<X3> LY;
  for (int i = tid; i < LY::WC / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);
  if (tid < 80) reinterpret_cast<float*>(smem + LY::WC)[tid] = tid < 73 ? p.wc[tid] : 0.f;
  if (X3) { for (int i = tid; i < 4 * 3 * 64; i += 512) reinterpret_cast<u32x4*>(smem + LY::W2)[i] = reinterpret_cast<const u32x4*>(p.w2x)[i]; }
  else { for (int i = tid; i < 4 * 8 * 64; i += 512) reinterpret_cast<float*>(smem + LY::W2)[i] = p.w2f[i]; }
  reinterpret_cast<float*>(smem + LY::W3)[tid] = p.w3f[tid];   // 2 * 4 * 64 = 512 floats

  // ---- P role: item = (tile, tx1) ----
  const int tile = wave & 3, tx1 = wave >> 2;
  const int px = 16 * tile + n;
  const bool px_ok = px < W;
  const int pxc = px_ok ? px : W - 1;
  const bool tile_on = 16 * tile < W;   // wave-uniform: tiles wholly past the row do nothing
  float wA[2][2][X3 ? 1 : 16];
  u32x4 wH[2][2][2], wM[2][2][2];          // X3: [ty1][m-tile t][k-step c], hi and mid planes
  const u32x4* w1x = reinterpret_cast<const u32x4*>(p.w1x) + lane;   // [(((ty1*2 + tx1)*2 + t)*2 + c)*3 + plane][64 lanes]
  const float* wB = reinterpret_cast<const float*>(smem + LY::W2) + lane;   // [(2 ty2 + tx2) * 8 + ks][64]
  const u32x4* wBx = reinterpret_cast<const u32x4*>(smem + LY::W2) + lane;    // X3: [(2 ty2 + tx2) * 3 + plane][64]
  const float* wC = reinterpret_cast<const float*>(smem + LY::W3) + lane;   // [u * 4 + i][64]
  f32x4 bA[2], bB, bC;
#pragma unroll
  for (int ty1 = 0; ty1 < 2; ++ty1)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (X3) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
          wH[ty1][t][c] = w1x[((((ty1 * 2 + tx1) * 2 + t) * 2 + c) * 3 + 0) * 64];
          wM[ty1][t][c] = w1x[((((ty1 * 2 + tx1) * 2 + t) * 2 + c) * 3 + 1) * 64];
        }
      } else {
#pragma unroll
        for (int s = 0; s < 16; ++s) wA[ty1][t][s] = p.w1f[((((2 * ty1 + tx1) * 2 + t) * 16) + s) * 64 + lane];
      }
    }
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) bA[t][i] = p.b1[16 * t + 4 * rg + i];
#pragma unroll
  for (int i = 0; i < 4; ++i) { bB[i] = p.b2[4 * rg + i]; bC[i] = p.b3[4 * (rg & 1) + i]; }
  // ring byte offset of this lane's granule, without the row and the 2*tx2 plane term
  const int p_off = (rg & 1) * R_HALF + (4 * tx1 + (rg >> 1)) * R_PLANE + (px + 1) * 16;

  // ---- D role: row q of the strip, 4 adjacent pixels per lane ----
  const int dq = wave >> 1;
  const int db = (wave & 1) * W + lane;           // x-block, 0 .. 2W-1
  const bool d_lane = lane < W;
  const int dbc = d_lane ? db : 0;
  const int d_c = (4 * (dbc & 1)) * R_PLANE + ((dbc >> 1) + 1) * 16;                                   // columns X0 .. X0+3: + j * R_PLANE
  const int d_l = (dbc & 1) ? 3 * R_PLANE + ((dbc >> 1) + 1) * 16 : 7 * R_PLANE + (dbc >> 1) * 16;      // column X0 - 1
  const int d_r = (dbc & 1) ? ((dbc >> 1) + 2) * 16 : 4 * R_PLANE + ((dbc >> 1) + 1) * 16;              // column X0 + 4
  const f32x4* wk = reinterpret_cast<const f32x4*>(smem + LY::WC);   // [tap][half]
  unsigned bad_total = 0;
  double sse = 0.0;

  // ---- job iterator: virtual samples blockIdx.x + j * gridDim.x, each a run of strips.  Plain scalars kept wave-uniform
  // (readfirstlane) so that the whole bookkeeping runs on the scalar unit.
#define UNI(x) __builtin_amdgcn_readfirstlane(x)
  const int S = p.seg > 1 ? p.seg : 1, NV = p.n * S;   // segment s of a sample: strips [s SH / S, (s + 1) SH / S) -- S need not divide SH
  int it_jv = (int)blockIdx.x, it_g1 = 0;
  Strip it{0, 0, 0, 0};
  // first strip of virtual sample it_jv (warm-up strip first, unless the segment starts at the top of the image)
#define JOB_START()                                                                     \
  do {                                                                                  \
    if (it_jv < NV) {                                                                   \
      const int smp_ = it_jv / S, seg_ = it_jv - smp_ * S, g0_ = seg_ * SH / S;         \
      it.valid = 1; it.smp = UNI(smp_); it.g = UNI(g0_ > 0 ? g0_ - 1 : 0); it.warm = UNI(g0_ > 0 ? 1 : 0); \
      it_g1 = UNI((seg_ + 1) * SH / S);                                                 \
    } else it.valid = 0;                                                                \
  } while (0)
  JOB_START();
  Strip cur = it, d1{0, 0, 0, 0}, d2{0, 0, 0, 0};

  auto load_x = [&](const Strip& s, f32x4 (&x)[4]) {
    // f32 MFMAs: channels 16 rg + 0..15 (k-step s contracts channel 16 kg + s); X3: channels 8 rg + 0..7 and 32 + 8 rg + 0..7 (the two
    // 32-deep k-steps of v_mfma_f32_16x16x32_bf16: lane group kg holds k = 8 kg + j)
    const float* src = p.in + ((((size_t)s.smp * H + (s.g >> 1)) * W + pxc) * 64 + (X3 ? 8 : 16) * rg);
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f32x4*>(src + (X3 ? 32 * (q >> 1) + 4 * (q & 1) : 4 * q));
  };
  f32x4 xs[4], xn[4];
#pragma unroll
  for (int q = 0; q < 4; ++q) { xs[q] = f32x4{0, 0, 0, 0}; xn[q] = f32x4{0, 0, 0, 0}; }
  if (cur.valid && tile_on) load_x(cur, xs);
  __syncthreads();

  int ring0 = 0;  // ring row of this round's strip row 0: (4 * round) % 10
  float d_mean = 0.f, d_sd = 1.f;
  int aff_smp = -1;
  for (;;) {
    if (!cur.valid && !d1.valid && !d2.valid) break;
    // the strip after this one: fetch its input now when it is a different 50-level row
    if (it.valid) {                      // one strip further
      if (it.g + 1 < it_g1) { it.g = it.g + 1; it.warm = 0; }
      else { it_jv = UNI(it_jv + (int)gridDim.x); JOB_START(); }
    }
    const Strip nxt = it;
    // Its input row is fetched every round, needed or not (a re-read of the same row hits L2), from inside the block below:
    // a conditional fetch would be its own basic block, and issued up here it would sit in front of the block's first
    // vector-memory wait (which the compiler makes a full vmcnt(0)): a round would start by waiting for its own prefetch.
    const Strip ldx = nxt.valid ? nxt : (cur.valid ? cur : Strip{0, 0, 0, 0});

    // ---- D bookkeeping (scalar unit).  Row q of this wave: q == 0 is row 3 of the strip issued two rounds ago (its lower
    // neighbour is row 0 of last round's strip, or the zero row at the bottom of the image); q = 1..3 is row q-1 of last
    // round's strip. ----
    int d_smp, d_yl, ro0, ro1, ro2;      // output row (sample, local row); ring byte offsets of the three window rows
    bool d_on;
    {
      const int base = ring0 + 2 * R_ROWS - 4;   // ring row of last round's strip row 0 (+ 2 R_ROWS: differences stay positive)
      if (dq == 0) {
        const bool below = d1.valid && d1.smp == d2.smp && d1.g == d2.g + 1;
        d_on = d2.valid && (below || d2.g == SH - 1);
        d_smp = d2.smp; d_yl = 4 * d2.g + 3;
        ro0 = ((base - 2) % R_ROWS) * R_ROW;
        ro1 = ((base - 1) % R_ROWS) * R_ROW;
        ro2 = below ? (base % R_ROWS) * R_ROW : LY::ZROW;
      } else {
        d_on = d1.valid && !d1.warm;
        d_smp = d1.smp; d_yl = 4 * d1.g + dq - 1;
        const bool above = dq > 1 || (d2.valid && d2.smp == d1.smp && d2.g + 1 == d1.g);   // else: top of the image
        ro0 = above ? ((base + dq - 2) % R_ROWS) * R_ROW : LY::ZROW;
        ro1 = ((base + dq - 1) % R_ROWS) * R_ROW;
        ro2 = ((base + dq) % R_ROWS) * R_ROW;
      }
      if (!d_on) { d_smp = 0; d_yl = 0; }
    }
    // de-standardisation scalars: (re)read on the scalar unit only when the sample changes
    if (p.aff_out && d_on && d_smp != aff_smp) {
      typedef const __attribute__((address_space(4))) float* cfp;
      cfp ap = (cfp)(uintptr_t)(p.aff_out + 2 * (size_t)d_smp);
      d_mean = ap[0]; d_sd = ap[1];
      aff_smp = d_smp;
    }
    const bool p_on = cur.valid && tile_on;
    int prow[4];                         // ring byte offsets of the four rows this round's strip writes
#pragma unroll
    for (int j = 0; j < 4; ++j) { const int r = ring0 + j; prow[j] = (r >= R_ROWS ? r - R_ROWS : r) * R_ROW; }

    // ================= one straight-line block: the first layer's 32 MFMAs run beside the output conv's vector work ==========
    f32x4 a0 = bA[0], a1 = bA[1];
    f32x2 acc[4];
    {
      const float cbias = reinterpret_cast<const float*>(smem + LY::WC)[72];
#pragma unroll
      for (int o = 0; o < 4; ++o) acc[o] = f32x2{cbias, 0.f};
    }
    auto layer1 = [&](auto TY1, auto S0, auto S1) {   // k-steps [S0, S1) of first-layer tap row TY1
      constexpr int ty1 = decltype(TY1)::value;
#pragma unroll
      for (int s = decltype(S0)::value; s < decltype(S1)::value; ++s) {
        const float xv = xs[s >> 2][s & 3];
        a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ty1][0][s], xv, a0, 0, 0, 0);
        a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wA[ty1][1][s], xv, a1, 0, 0, 0);
      }
    };
    // X3: the round's input split into three bf16 planes, the lo weight plane of this round's tap row on its way
    u32x4 xh[2], xm[2], xl[2], wL[2][2];
    f32x4 r0 = {0.f, 0.f, 0.f, 0.f}, r1 = {0.f, 0.f, 0.f, 0.f};   // the small products' accumulators (a0 / a1 take hi x hi and the bias)
    if (X3) {
      const int ty1r = cur.g & 1;
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int c = 0; c < 2; ++c) wL[t][c] = w1x[((((ty1r * 2 + tx1) * 2 + t) * 2 + c) * 3 + 2) * 64];
      if (T32_ABL(16)) {
        xh[0] = __builtin_bit_cast(u32x4, xs[0]); xm[0] = __builtin_bit_cast(u32x4, xs[1]); xl[0] = xh[0];
        xh[1] = __builtin_bit_cast(u32x4, xs[2]); xm[1] = __builtin_bit_cast(u32x4, xs[3]); xl[1] = xh[1];
      } else {
      split8_t32(xs[0], xs[1], xh[0], xm[0], xl[0]);
      split8_t32(xs[2], xs[3], xh[1], xm[1], xl[1]);
      }
    }
    auto layer1x = [&](auto TY1, auto PART) {   // PART 0: mid x mid, hi x lo; 1: hi x mid, mid x hi, hi x hi; 2: lo x hi (the late plane), sets meet
      constexpr int ty1 = decltype(TY1)::value;
      constexpr int part = decltype(PART)::value;
      if (T32_ABL(8)) { if (part == 2) { a0 = a0 + __builtin_bit_cast(f32x4, xh[0]); a1 = a1 + __builtin_bit_cast(f32x4, xm[1]) + __builtin_bit_cast(f32x4, xl[0]); } return; }
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        if (part == 0) {
          r0 = mfma16bf(wM[ty1][0][c], xm[c], r0); r1 = mfma16bf(wM[ty1][1][c], xm[c], r1);
          r0 = mfma16bf(wH[ty1][0][c], xl[c], r0); r1 = mfma16bf(wH[ty1][1][c], xl[c], r1);
        } else if (part == 1) {
          r0 = mfma16bf(wH[ty1][0][c], xm[c], r0); r1 = mfma16bf(wH[ty1][1][c], xm[c], r1);
          r0 = mfma16bf(wM[ty1][0][c], xh[c], r0); r1 = mfma16bf(wM[ty1][1][c], xh[c], r1);
          a0 = mfma16bf(wH[ty1][0][c], xh[c], a0); a1 = mfma16bf(wH[ty1][1][c], xh[c], a1);
        } else {
          r0 = mfma16bf(wL[0][c], xh[c], r0); r1 = mfma16bf(wL[1][c], xh[c], r1);
        }
      }
      if (part == 2) { a0 = a0 + r0; a1 = a1 + r1; }
    };
    auto window_row = [&](auto DY, const int ro) {
      constexpr int dy = decltype(DY)::value;
      if (X3 && ro < 0) return;           // a row outside the image (wave-uniform): nothing to add
      if (T32_ABL(64)) return;
      const char* rowp = smem + ro;
      f32x4 lo[6], hi[6];                 // window columns X0 - 1 .. X0 + 4 of this row
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        const char* gp = rowp + (j == 0 ? d_l : (j == 5 ? d_r : d_c + (j - 1) * R_PLANE));
        lo[j] = *reinterpret_cast<const f32x4*>(gp);
        hi[j] = *reinterpret_cast<const f32x4*>(gp + R_HALF);
      }
      if (T32_ABL(2)) {   // keep the reads alive: one add per granule
#pragma unroll
        for (int j = 0; j < 6; ++j) acc[j & 3] = acc[j & 3] + f32x2{lo[j][0], hi[j][3]};
      } else
#pragma unroll
      for (int dx = 0; dx < 3; ++dx) {
        const f32x4 wl = wk[(dy * 3 + dx) * 2], wh = wk[(dy * 3 + dx) * 2 + 1];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
          const int j = o + dx;           // column X0 + o + (dx - 1)
          acc[o] = __builtin_elementwise_fma(f32x2{lo[j][0], lo[j][1]}, f32x2{wl[0], wl[1]}, acc[o]);
          acc[o] = __builtin_elementwise_fma(f32x2{lo[j][2], lo[j][3]}, f32x2{wl[2], wl[3]}, acc[o]);
          acc[o] = __builtin_elementwise_fma(f32x2{hi[j][0], hi[j][1]}, f32x2{wh[0], wh[1]}, acc[o]);
          acc[o] = __builtin_elementwise_fma(f32x2{hi[j][2], hi[j][3]}, f32x2{wh[2], wh[3]}, acc[o]);
        }
      }
      // one window row (48 + 24 registers) in flight at a time: left alone the compiler issues all 54 LDS reads of the
      // three rows first (216 registers) and spills the resident first-layer weights
      asm volatile("" : "+v"(acc[0]), "+v"(acc[1]), "+v"(acc[2]), "+v"(acc[3]) : : "memory");
    };
    // the first layer's MFMAs between the three window rows of the output conv (one copy of the block per tap row: selecting
    // the weights per k-step instead costs 32 vector instructions per item)
    const size_t o_row = ((size_t)d_smp * OHs + d_yl) * OW;
    f32x4 tgt = {0.f, 0.f, 0.f, 0.f};
    auto front = [&](auto TY1) {
      if (X3) layer1x(TY1, std::integral_constant<int, 0>{});
      else layer1(TY1, std::integral_constant<int, 0>{}, std::integral_constant<int, 6>{});
      window_row(std::integral_constant<int, 0>{}, ro0);
      load_x(ldx, xn);   // behind the asm of window_row (a memory clobber): cannot be hoisted above the first MFMA
      if (TRAIN) tgt = *reinterpret_cast<const f32x4*>(p.target + o_row + 4 * (d_lane ? db : 0));
      if (X3) layer1x(TY1, std::integral_constant<int, 1>{});
      else layer1(TY1, std::integral_constant<int, 6>{}, std::integral_constant<int, 11>{});
      window_row(std::integral_constant<int, 1>{}, ro1);
      if (X3) layer1x(TY1, std::integral_constant<int, 2>{});
      else layer1(TY1, std::integral_constant<int, 11>{}, std::integral_constant<int, 16>{});
      window_row(std::integral_constant<int, 2>{}, ro2);
    };
    if (cur.g & 1) front(std::integral_constant<int, 1>{});
    else front(std::integral_constant<int, 0>{});
    {
      float v[4];
      const bool st_on = d_on && d_lane;
#pragma unroll
      for (int o = 0; o < 4; ++o) {
        float t = acc[o][0] + acc[o][1];
        if (TRAIN) {
          const float e = t - tgt[o];
          if (st_on) sse += (double)e * (double)e;
          t = p.two_scale * e;
        }
        if (!TRAIN && p.aff_out) t = __fadd_rn(__fmul_rn(t, d_sd), d_mean);
        if (!TRAIN && p.nan_guard) {
          const bool bad = st_on && !(fabsf(t) <= 3.402823466e38f);
          bad_total += (unsigned)__popcll(__ballot(bad));
          t = bad ? 0.f : t;
        }
        v[o] = t;
      }
      const size_t o0 = o_row + 4 * (d_lane ? db : 0);
      if (st_on) {
        if (OUT == 0) *reinterpret_cast<f32x4*>(reinterpret_cast<float*>(p.out) + o0) = f32x4{v[0], v[1], v[2], v[3]};
        else {
          typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
          typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
          const f32x4 vv = {v[0], v[1], v[2], v[3]};
          uint2 pk;
          if (OUT == 1) pk = __builtin_bit_cast(uint2, __builtin_convertvector(vv, bf16x4));
          else pk = __builtin_bit_cast(uint2, __builtin_convertvector(vv, h16x4));
          *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.out) + o0) = pk;
        }
      }
    }
    // ---- the rest of the chain: second layer (both tap rows), third layer; the activation of one tile runs beside the
    // matrix work of the next (independent) one ----
    if (!T32_ABL(1)) { a0 = swish_l2e(a0, one2); a1 = swish_l2e(a1, one2); }
    f32x4 b[2][2];
    if (X3) {
      // the second layer the same way: its B operand is the 8 activations a lane holds (k index 8 kg + j <-> channel 4 kg + j for
      // j < 4, 16 + 4 kg + j - 4 above: the host packs the weight planes in that order), K = 32 = one k-step, six MFMAs per tap
      u32x4 yh, ym, yl;
      if (T32_ABL(16)) { yh = __builtin_bit_cast(u32x4, a0); ym = __builtin_bit_cast(u32x4, a1); yl = yh; }
      else split8_t32(a0, a1, yh, ym, yl);
#pragma unroll
      for (int ty2 = 0; ty2 < 2; ++ty2)
#pragma unroll
        for (int tx2 = 0; tx2 < 2; ++tx2) {
          const u32x4 wh = wBx[((2 * ty2 + tx2) * 3 + 0) * 64], wm = wBx[((2 * ty2 + tx2) * 3 + 1) * 64], wl = wBx[((2 * ty2 + tx2) * 3 + 2) * 64];
          f32x4 hh = bB, rr = {0.f, 0.f, 0.f, 0.f};
          if (T32_ABL(8)) { b[ty2][tx2] = hh + __builtin_bit_cast(f32x4, yh) + __builtin_bit_cast(f32x4, ym) + __builtin_bit_cast(f32x4, yl) + __builtin_bit_cast(f32x4, wh); continue; }
          rr = mfma16bf(wm, ym, rr); rr = mfma16bf(wh, yl, rr); rr = mfma16bf(wl, yh, rr); rr = mfma16bf(wh, ym, rr); rr = mfma16bf(wm, yh, rr);
          hh = mfma16bf(wh, yh, hh);
          b[ty2][tx2] = hh + rr;
        }
    } else {
#pragma unroll
      for (int ty2 = 0; ty2 < 2; ++ty2) {
        b[ty2][0] = bB; b[ty2][1] = bB;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
          const float bv = ks < 4 ? a0[ks & 3] : a1[ks & 3];
          b[ty2][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[((2 * ty2) * 8 + ks) * 64], bv, b[ty2][0], 0, 0, 0);
          b[ty2][1] = __builtin_amdgcn_mfma_f32_16x16x4f32(wB[((2 * ty2 + 1) * 8 + ks) * 64], bv, b[ty2][1], 0, 0, 0);
        }
      }
    }
    const int st_sel = (px_ok && p_on) ? 0 : 1;   // lanes / rounds with nothing to store write a dump granule instead (no branch)
#pragma unroll
    for (int ty2 = 0; ty2 < 2; ++ty2) {
      if (!T32_ABL(1)) { b[ty2][0] = swish_l2e(b[ty2][0], one2); b[ty2][1] = swish_l2e(b[ty2][1], one2); }
#pragma unroll
      for (int tx2 = 0; tx2 < 2; ++tx2) {
        const f32x4 bs = b[ty2][tx2];
        f32x4 c0 = bC, c1 = bC;
        if (T32_ABL(4)) { c0 = c0 + bs; c1 = c1 - bs; } else
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(wC[i * 64], bs[i], c0, 0, 0, 0);
          c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(wC[(4 + i) * 64], bs[i], c1, 0, 0, 0);
        }
        if (!T32_ABL(1)) { c0 = swish_l2e(c0, one2); c1 = swish_l2e(c1, one2); }
        const int o_a = st_sel ? LY::DUMP + lane * 16 : prow[2 * ty2] + p_off + 2 * tx2 * R_PLANE;
        const int o_b = st_sel ? LY::DUMP + lane * 16 : prow[2 * ty2 + 1] + p_off + 2 * tx2 * R_PLANE;
        if (T32_ABL(32)) { if (c0[0] + c1[1] == 123.456f) *reinterpret_cast<f32x4*>(smem + o_a) = c0; }
        else {
        *reinterpret_cast<f32x4*>(smem + o_a) = c0;
        *reinterpret_cast<f32x4*>(smem + o_b) = c1;
        }
      }
    }

    lds_barrier32();
#pragma unroll
    for (int q = 0; q < 4; ++q) xs[q] = xn[q];
    d2 = d1; d1 = cur; cur = nxt;
    ring0 += 4;
    ring0 = ring0 >= R_ROWS ? ring0 - R_ROWS : ring0;
  }
  if (!TRAIN && p.nan_guard && p.nonfinite && bad_total && lane == 0) atomicAdd(p.nonfinite, (unsigned long long)bad_total);
  if (TRAIN) {   // lanes -> waves -> workgroup, always in the same order
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sse += __shfl_xor(sse, o, 64);
    __syncthreads();   // the ring is dead: its first bytes carry the eight wave sums
    double* red = reinterpret_cast<double*>(smem);
    if (lane == 0) red[wave] = sse;
    __syncthreads();
    if (tid == 0) p.sse_partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7]));
  }
}

hipError_t launch_tail32(const Tail32Params& p, int num_cus, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  void (*fn)(Tail32Params) = p.target ? tail32<0, true>
                             : (p.out_dtype == SRCFD_F32 ? (p.w1x ? tail32<0, false, true> : tail32<0, false>)
                                : (p.out_dtype == SRCFD_BF16 ? (p.w1x ? tail32<1, false, true> : tail32<1, false>) : (p.w1x ? tail32<2, false, true> : tail32<2, false>)));
  if (p.target && p.w1x) return hipErrorInvalidValue;
  if (p.target && (p.out_dtype != SRCFD_F32 || !p.sse_partial)) return hipErrorInvalidValue;
  if ((p.w1x == nullptr) != (p.w2x == nullptr)) return hipErrorInvalidValue;
  const int lds = p.w1x ? T32Lay<true>::LDS : T32_LDS;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(fn), lds);
  if (e != hipSuccess) return e;
  const int S = p.seg > 1 ? p.seg : 1;
  const int blocks = tail32_blocks(p.n, S, num_cus);
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), lds, s, p);
  return hipGetLastError();
}

int tail32_blocks(int n, int seg, int num_cus) { return (int)std::min<int64_t>((int64_t)n * (seg > 1 ? seg : 1), num_cus); }

// Segments per sample for a batch of n samples of 2H strips each: the busiest workgroup walks ceil(n S / CUs) virtual
// samples of ceil(2H / S) (+1 warm-up) strips, + 2 rounds of pipeline depth; more segments must buy 10 % to be taken.  S need
// not divide 2H (32 samples on 256 CUs: 8 segments of 12 or 13 strips, one virtual sample per CU -- 16 rounds; the best
// divisor, 20, takes 20).  Every output pixel goes through the same arithmetic whatever S is.
int tail32_segments(int n, int H, int num_cus) {
  const int SH = 2 * H;
  long best = ((long)(n + num_cus - 1) / num_cus) * SH + 2;
  int seg = 1;
  for (int cand = 2; cand <= SH; ++cand) {
    const long cost = ((long)((long)n * cand + num_cus - 1) / num_cus) * ((SH + cand - 1) / cand + 1) + 2;
    if (cost * 110 < best * 100) { best = cost; seg = cand; }
  }
  return seg;
}

}  // namespace srcfd
