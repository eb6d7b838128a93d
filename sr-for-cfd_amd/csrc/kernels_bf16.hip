// bf16 / f16 kernels of the SR engine (gfx950 only): the throughput path.
//
//   enc_conv1_16   Conv2D 3x3 s2 SAME 1->64 + standardise + swish (VALU)        SURVEY 8a rows a2, a7
//   gemm16         implicit GEMM on v_mfma_f32_32x32x16_{bf16,f16}; same GemmDesc
//                  as the f32 engine; pixels on the MFMA column (lane) axis so a
//                  lane owns 4 consecutive output channels -> 8-byte stores     rows a8-a14
//   tail16         ConvT#2 -> ConvT#3 -> ConvT#4 -> output conv + de-standardise
//                  + NaN guard, one launch, nothing between 50x50x64 and the
//                  400x400 image touches HBM                                     rows a15-a20
//
// Activations carry a factor log2(e): every swish layer's GEMM produces
// u = log2e * x, the epilogue computes u * rcp(1 + exp2(-u)) = log2e * swish(x)
// (v_exp_f32 with a negated source, no extra multiply) and the consumer's
// weights absorb 1/log2e on the host.  For swish->swish layers the two factors
// cancel and the weights are used as they are.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "dev16.h"
#include "kernels16.h"
#include "tail16_layout.h"

namespace srcfd {

// ---------------------------------------------------------------------------
// encoder conv #1: x (n,10,10,1) f32 [+ standardise] -> (n,5,5,64) bf16/f16
// one thread = 4 output channels of one output pixel
// ---------------------------------------------------------------------------
template <bool F16>
__global__ void __launch_bounds__(256) enc_conv1_16(const float* __restrict__ x, const float* __restrict__ affine,
                                                     const float* __restrict__ w /*[9][64] scaled*/, const float* __restrict__ b /*[64] scaled*/,
                                                     uint16_t* __restrict__ y, int n) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= n * 25 * 16) return;
  int c4 = idx & 15, pix = (idx >> 4) % 25, s = idx / (25 * 16);
  int oy = pix / 5, ox = pix - oy * 5;
  float mean = 0.f, sd = 1.f;
  if (affine) { mean = affine[2 * s]; sd = affine[2 * s + 1]; if (sd == 0.f) sd = 1e-8f; }
  float acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) acc[c] = b[c4 * 4 + c];
  const float* xs = x + (size_t)s * 100;
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    int iy = 2 * oy + ky;  // TF SAME, stride 2, 10->5: pad 0 before / 1 after
    if (iy >= 10) continue;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      int ix = 2 * ox + kx;
      if (ix >= 10) continue;
      float v = xs[iy * 10 + ix];
      if (affine) v = __fdiv_rn(__fsub_rn(v, mean), sd);
      const float* wp = w + (ky * 3 + kx) * 64 + c4 * 4;
#pragma unroll
      for (int c = 0; c < 4; ++c) acc[c] = fmaf(v, wp[c], acc[c]);
    }
  }
  uint2 o;
  o.x = pack2<F16>(swish_scaled(acc[0]), swish_scaled(acc[1]));
  o.y = pack2<F16>(swish_scaled(acc[2]), swish_scaled(acc[3]));
  *reinterpret_cast<uint2*>(y + ((size_t)s * 25 + pix) * 64 + c4 * 4) = o;
}

// ---------------------------------------------------------------------------
// implicit GEMM, 16-bit operands.  D[channel][pixel] = Wt[channel][k] * X[pixel][k]
// Block 256 threads; tile 128 pixels x BN channels x 64 k; one LDS stage (37 KB) with the
// next k-tile's global loads held in registers while the current one feeds the MFMAs, and
// <= 128 VGPRs, so four workgroups share a CU: these layers are bound by memory latency
// (a k-tile is ~0.5k cycles of MFMA against ~5k cycles of load latency), and in-flight
// loads per CU is what buys time.  LDS rows are 144 B (64 elements + 8 pad): a
// ds_read_b128 lane group then covers 16 distinct 16-byte slots of the 256-byte bank row.
// SPLITK: blockIdx.z takes a K slice and adds f32 partial sums into `part`
// (zeroed by the caller); splitk_finish16 applies bias + activation.
// ---------------------------------------------------------------------------
constexpr int G_BP = 128, G_BK = 64, G_PITCH = 72;

template <bool F16, int BN, bool SPLITK>
__global__ void __launch_bounds__(256, BN == 128 ? 3 : 4) gemm16(GemmDesc d, const uint16_t* __restrict__ X, const uint16_t* __restrict__ Wt, int Kpad,
                                               const float* __restrict__ bias, uint16_t* __restrict__ Y, float* __restrict__ part, int kslice) {
  constexpr int PT = BN == 128 ? 2 : 1;  // 32-pixel tiles per wave
  constexpr int WCH = BN / 64;           // 16-byte weight chunks per thread and k-tile
  extern __shared__ __attribute__((aligned(16))) char gsm[];
  uint16_t* Xs = reinterpret_cast<uint16_t*>(gsm);                       // [G_BP][G_PITCH]
  uint16_t* Ws = Xs + G_BP * G_PITCH;                                    // [BN][G_PITCH]
  int* row_img = reinterpret_cast<int*>(Ws + BN * G_PITCH);              // [G_BP] x3
  int* row_my = row_img + G_BP;
  int* row_mx = row_my + G_BP;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int wc = BN == 128 ? (wave >> 1) : 0, wp = BN == 128 ? (wave & 1) : wave;
  const int m0 = blockIdx.x * G_BP, n0 = blockIdx.y * BN;
  const int kbeg = SPLITK ? blockIdx.z * kslice : 0;
  const int kend = SPLITK ? min(d.K, kbeg + kslice) : d.K;

  if (tid < G_BP) {
    int m = m0 + tid, img = -1, my = 0, mx = 0;
    if (m < d.M) {
      int per = d.MH * d.MW;
      img = m / per;
      int r = m - img * per;
      my = r / d.MW;
      mx = r - my * d.MW;
    }
    row_img[tid] = img; row_my[tid] = my; row_mx[tid] = mx;
  }
  __syncthreads();

  // this thread's four X chunks: rows (tid>>3) + 32j, 16-byte column tid&7.  Everything that
  // does not change with the k-tile is folded into 32-bit element offsets up front, so a k-tile
  // costs one add (+ two compares when the tap can fall outside the image) per 16-byte load.
  const int xrow = tid >> 3, c8 = tid & 7;
  const bool check = d.TY * d.TX > 1 || d.cy != 0 || d.cx != 0;  // taps that can leave the image
  int xoff[4], xy[4], xx[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    int img = row_img[xrow + 32 * j];
    int by0 = row_my[xrow + 32 * j] * d.ay + d.cy, bx0 = row_mx[xrow + 32 * j] * d.ax + d.cx;
    xy[j] = img >= 0 ? by0 : -(1 << 28);  // rows past M never pass the bounds test
    xx[j] = bx0;
    xoff[j] = ((img * d.IH + by0) * d.IW + bx0) * d.CI + c8 * 8;
    if (!check && img < 0) xoff[j] = c8 * 8;  // dense / k==s layers: read row 0, result is discarded
  }
  int woff[2 * (BN / 64)];
#pragma unroll
  for (int j = 0; j < 2 * (BN / 64); ++j) woff[j] = (n0 + xrow + 32 * j) * Kpad + c8 * 8;
  const int lds_x = xrow * G_PITCH + c8 * 8;  // + 32j rows

  uint4 xr[4], wr0, wr1, wr2 = make_uint4(0, 0, 0, 0), wr3 = make_uint4(0, 0, 0, 0);
  auto g2r = [&](int k0) {
    int tap = k0 / d.CI, ci0 = k0 - tap * d.CI;
    int ty = tap / d.TX, tx = tap - ty * d.TX;
    const int dy = ty * d.by, dx = tx * d.bx;
    const int toff = (dy * d.IW + dx) * d.CI + ci0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      uint4 v = make_uint4(0, 0, 0, 0);
      if (!check || ((unsigned)(xy[j] + dy) < (unsigned)d.IH && (unsigned)(xx[j] + dx) < (unsigned)d.IW))
        v = *reinterpret_cast<const uint4*>(X + (xoff[j] + toff));
      xr[j] = v;
    }
    // named registers, not an array: the array form ended up in scratch (a store/load round trip right
    // after every weight load, which serialised the prefetch)
    wr0 = *reinterpret_cast<const uint4*>(Wt + (woff[0] + k0));
    wr1 = *reinterpret_cast<const uint4*>(Wt + (woff[1] + k0));
    if (WCH == 2) {
      wr2 = *reinterpret_cast<const uint4*>(Wt + (woff[2 % (2 * WCH)] + k0));
      wr3 = *reinterpret_cast<const uint4*>(Wt + (woff[3 % (2 * WCH)] + k0));
    }
  };
  auto r2l = [&]() {
    uint16_t* xs = Xs + lds_x;
    uint16_t* ws = Ws + lds_x;
#pragma unroll
    for (int j = 0; j < 4; ++j) *reinterpret_cast<uint4*>(xs + 32 * j * G_PITCH) = xr[j];
    *reinterpret_cast<uint4*>(ws) = wr0;
    *reinterpret_cast<uint4*>(ws + 32 * G_PITCH) = wr1;
    if (WCH == 2) {
      *reinterpret_cast<uint4*>(ws + 64 * G_PITCH) = wr2;
      *reinterpret_cast<uint4*>(ws + 96 * G_PITCH) = wr3;
    }
  };

  f32x16 acc[2][PT];
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < PT; ++b)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

  g2r(kbeg);
  const uint16_t* xs = Xs + (wp * 32 * PT + l31) * G_PITCH + h * 8;
  const uint16_t* ws = Ws + (wc * 64 + l31) * G_PITCH + h * 8;
  for (int k0 = kbeg; k0 < kend; k0 += G_BK) {
    r2l();
    __syncthreads();
    if (k0 + G_BK < kend) g2r(k0 + G_BK);
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      uint4 af[2], bf[PT];
#pragma unroll
      for (int a = 0; a < 2; ++a) af[a] = *reinterpret_cast<const uint4*>(ws + a * 32 * G_PITCH + kk * 16);
#pragma unroll
      for (int b = 0; b < PT; ++b) bf[b] = *reinterpret_cast<const uint4*>(xs + b * 32 * G_PITCH + kk * 16);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < PT; ++b) acc[a][b] = mfma32<F16>(af[a], bf[b], acc[a][b]);
    }
    __syncthreads();
  }

  // epilogue.  In the accumulator layout a lane owns one pixel (row) and 4 consecutive channels per register quad;
  // stored directly that is one 8-byte piece per lane, rows a whole pixel (or, for Dense layers, a whole sample)
  // apart -- store-issue bound.  So the packed tile goes through LDS (the operand tiles are dead by now) and leaves
  // as 16-byte chunks, 8 or 16 consecutive lanes per output row.
  constexpr int SP = BN * 2 + 16;  // staging row pitch in bytes
  char* stg = gsm;
  if (SPLITK) {
#pragma unroll
    for (int b = 0; b < PT; ++b) {
      const int prow = wp * 32 * PT + b * 32 + l31;
      if (row_img[prow] < 0) continue;
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int nb = n0 + wc * 64 + a * 32;  // wave-uniform
        if (nb >= d.N) continue;
        // one f32 slab per K slice, summed in slice order by splitk_finish16: reproducible bit for bit
        float* pp = part + ((int64_t)blockIdx.z * d.M + (m0 + prow)) * d.Npad + nb + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<float4*>(pp + 8 * q) = make_float4(acc[a][b][4 * q], acc[a][b][4 * q + 1], acc[a][b][4 * q + 2], acc[a][b][4 * q + 3]);
      }
    }
    return;
  }
#pragma unroll
  for (int b = 0; b < PT; ++b) {
    const int prow = wp * 32 * PT + b * 32 + l31;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const int cb = wc * 64 + a * 32;       // channel offset inside the block tile, wave-uniform
      const int nb = n0 + cb;
      if (nb >= d.N) continue;
      const float* bp = bias + nb + 4 * h;
      uint32_t o[8];
      if (d.act == SRCFD_ACT_SWISH) {
        f32x16 u;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * q);
          u[4 * q] = acc[a][b][4 * q] + bv.x; u[4 * q + 1] = acc[a][b][4 * q + 1] + bv.y;
          u[4 * q + 2] = acc[a][b][4 * q + 2] + bv.z; u[4 * q + 3] = acc[a][b][4 * q + 3] + bv.w;
        }
        swish_pack16<F16>(u, o);
      } else {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const float4 bv = *reinterpret_cast<const float4*>(bp + 8 * q);
          o[2 * q] = pack2<F16>(acc[a][b][4 * q] + bv.x, acc[a][b][4 * q + 1] + bv.y);
          o[2 * q + 1] = pack2<F16>(acc[a][b][4 * q + 2] + bv.z, acc[a][b][4 * q + 3] + bv.w);
        }
      }
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<uint2*>(stg + prow * SP + (cb + 8 * q + 4 * h) * 2) = make_uint2(o[2 * q], o[2 * q + 1]);
    }
  }
  __syncthreads();
  constexpr int CPR = BN / 8;  // 16-byte chunks per row
#pragma unroll
  for (int j = 0; j < G_BP * CPR / 256; ++j) {
    const int e = tid + 256 * j, row = e / CPR, c = e - row * CPR;
    const int img = row_img[row], n = n0 + 8 * c;
    if (img < 0 || n >= d.N) continue;
    const int ph = n / d.CO, co = n - ph * d.CO, py = ph / d.nphx, px = ph - py * d.nphx;
    const int off = ((img * d.OH + row_my[row] * d.os + d.oy0 + py) * d.OW + row_mx[row] * d.os + d.ox0 + px) * d.OC + co;
    const uint4 v = *reinterpret_cast<const uint4*>(stg + row * SP + c * 16);
    if (n + 8 <= d.N && (off & 7) == 0) *reinterpret_cast<uint4*>(Y + off) = v;
    else {
      if (n + 4 <= d.N) *reinterpret_cast<uint2*>(Y + off) = make_uint2(v.x, v.y);
      if (n + 8 <= d.N) *reinterpret_cast<uint2*>(Y + off + 4) = make_uint2(v.z, v.w);
    }
  }
}

// split-K epilogue for dense layers (1x1 row grid): y[m][n] = act(sum_z part[z][m][n] + bias[n])
template <bool F16>
__global__ void __launch_bounds__(256) splitk_finish16(const float* __restrict__ part, int nz, const float* __restrict__ bias,
                                                        uint16_t* __restrict__ Y, int M, int N, int Npad, int OC, int act) {
  int idx = blockIdx.x * 256 + threadIdx.x;
  int n4 = N / 4;
  if (idx >= M * n4) return;
  int m = idx / n4, n = (idx - m * n4) * 4;
  float4 v = *reinterpret_cast<const float4*>(bias + n);
  for (int z = 0; z < nz; ++z) {
    const float4 t = *reinterpret_cast<const float4*>(part + ((int64_t)z * M + m) * Npad + n);
    v.x += t.x; v.y += t.y; v.z += t.z; v.w += t.w;
  }
  uint2 o;
  o.x = pack2<F16>(act16(v.x, act), act16(v.y, act));
  o.y = pack2<F16>(act16(v.z, act), act16(v.w, act));
  *reinterpret_cast<uint2*>(Y + (int64_t)m * OC + n) = o;
}

// ---------------------------------------------------------------------------
// fused tail.  One 1024-thread workgroup per CU walks a sample top to bottom
// in strips of one 50x50-level row (= 2 rows at 100x100, 8 rows at 400x400).
// Every round the 16 waves pull work items from an LDS counter:
//   BC(s): 32 pixels of the 100x100 level -> ConvT#3 -> ConvT#4 in registers
//          (the f32 accumulator of one MFMA is the B operand of the next),
//          swish in the epilogues, 400x400x8 rows written to an 18-row LDS ring
//   A(s+1): ConvT#2 for the next strip, global 50x50x64 -> LDS 100x100x32
//   D(s-1): output 3x3 conv of the previous strip as a banded (Toeplitz) MFMA
//          over 2x8-pixel tiles read from the ring, + de-standardise + guard
// ---------------------------------------------------------------------------
// LDS layouts are bank-swizzled for the two access patterns that hit them:
//  ring (400-level, 16 B per pixel): a row is 8 planes (x & 7) of 52 granules
//    (1 + (x >> 3); granules 0 and 51 stay zero: the SAME padding of the output conv
//    at the left / right image edge, read like any other pixel -- no per-lane edge tests).  BC writes a wave of pixels 4 apart in x (-> consecutive
//    granules of two planes, 2-way at worst); D reads 16 tiles 8 apart in x
//    (-> 16 consecutive granules, conflict-free; the row pitch is a multiple of
//    256 B so the two window rows of one ds_read_b128 lane group interleave).
//  L100 (100-level, 4 chunks of 8 channels): [chunk][row][x parity][x >> 1];
//    A writes pixels 2 apart (-> consecutive), BC reads consecutive pixels
//    (parity planes 56 granules = 8 mod 16 apart -> conflict-free).
// Diagnostic work-skipping switches (TailParams::ablate) exist only in a -DSRCFD_DIAG build (make DIAG=1): the shipped
// kernels carry no path that turns work off.
#ifdef SRCFD_DIAG
#define TAIL_ABL(bit) (p.ablate & (bit))
#else
#define TAIL_ABL(bit) 0
#endif

// the 50x50x64 activation is read exactly once, the image written once: non-temporal on both sides (below)
typedef unsigned u32x4nt __attribute__((ext_vector_type(4)));
template <bool F16, int OUT, bool PROF = false, bool SEG = false>  // OUT: 0 f32, 1 bf16, 2 f16; PROF: per-wave section timers (diagnostic); SEG: samples cut into segments
__global__ void __launch_bounds__(1024) tail16(TailParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + T_OFF_RING;
  char* l100 = smem + T_OFF_L100;
  const char* cst = smem + T_OFF_CONST;

  for (int i = tid; i < TAIL_CONST_BYTES / 16; i += 1024)
    reinterpret_cast<uint4*>(smem + T_OFF_CONST)[i] = reinterpret_cast<const uint4*>(p.consts)[i];
  if (tid < 4) reinterpret_cast<int*>(smem + T_OFF_ZERO)[tid] = 0;
  for (int i = tid; i < T_OFF_L100 / 16; i += 1024) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);   // ring incl. its zero granules
  __syncthreads();

  const uint4* wc_f = reinterpret_cast<const uint4*>(cst + TC_OFF_WC);
  const uint4* w3_f = reinterpret_cast<const uint4*>(cst + TC_OFF_W3);
  const uint4* w2_f = reinterpret_cast<const uint4*>(p.w2frags);
  const float conv_bias = *reinterpret_cast<const float*>(cst + TC_OFF_BC);
  const uint4* const w4_f = reinterpret_cast<const uint4*>(cst + TC_OFF_W4);   // ConvT#4 A operand: read per item (one ds_read_b128)
  // D-item lane constants: tile within the item, window sub-pixel (dy,dx) of this lane's k-group,
  // byte offsets of the five window column pairs relative to granule (plane 0, tx)
  const int d_tsel = 4 * (lane & 3) + ((lane & 15) >> 2);
  // (window sub-pixel (dy, dx) of a lane's k-group and the five window column offsets of the seam path are recomputed inside do_d:
  // the hot loop has no register to spare for constants of a path that runs once per sample)

  // D fast path (every row pair that is not at a sample / segment seam): everything that depends on the lane only is
  // computed here, once; a round adds wave-uniform terms.  (These address and epilogue instructions were 65 of a D item's
  // vector instructions -- on this chip vector and matrix instructions of a SIMD do not overlap, so they are kernel time.)
  constexpr int OUTSZ = OUT == 0 ? 4 : 2;
  const int d_rowlane = ((lane >> 4) & 1) * T_ROWP;                  // window row of this lane's k-group, relative to the item's first
  // fast path (operands swapped, below): lane (n = lane & 15, kg = lane >> 4) stores pixels 4 (kg & 1) .. + 3 of row kg >> 1 of tile d_tsel
  const int d_olane = ((lane >> 5) * 400 + 8 * d_tsel + 4 * ((lane >> 4) & 1)) * OUTSZ;   // output byte offset inside the item's row pair
  const bool d_last_on = d_tsel < 2;                                                       // tiles 48, 49: the two that exist in a row pair's fourth item

  // ---- static schedule (per round: 14 BC, 8 A, 16 D items over 16 waves) ----
  // The kernel is VALU-issue bound and wave w runs on SIMD w % 4, so the schedule balances VALU instructions
  // per SIMD (measured with SRCFD_TAIL_PROF: BC ~266, A ~92, D ~65 instructions; the last wave of a SIMD
  // simply drains what the older waves leave): SIMD 0,1 get 4 BC + 2 A + 2 D, SIMD 2,3 get 3 BC + 2 A + 6 D.
  //   A : waves 0-7, item = wave; weights resident, activations prefetched one round ahead
  //   BC: waves 0,1,4,5,8,9,12,13 -> items 0-7; waves 2,3,6,7,10,11 -> items 8-13
  //   D : waves 8,9,12,13 and 2,3,6,7 -> one item each (0-7); waves 10,11,14,15 -> two each (8-15)
  const bool hasA = wave < 8;
  const int a_mt = wave & 3, a_ct = (wave >> 2) & 1;
  const int sq = wave >> 2, sl = wave & 3;       // wave = 4 * sq + sl, SIMD = sl
  int bc_item = -1, d_first = 0, d_cnt = 0;
  if (sl < 2) {                                   // SIMD 0,1
    bc_item = 2 * sq + sl;                        // 0..7
    if (sq >= 2) { d_first = 2 * (sq - 2) + sl; d_cnt = 1; }        // waves 8,9,12,13 -> 0..3
  } else {                                        // SIMD 2,3
    if (sq < 3) bc_item = 8 + 2 * sq + (sl - 2);  // waves 2,3,6,7,10,11 -> 8..13
    if (sq < 2) { d_first = 4 + 2 * sq + (sl - 2); d_cnt = 1; }      // waves 2,3,6,7 -> 4..7
    else { d_first = 8 + 4 * (sq - 2) + 2 * (sl - 2); d_cnt = 2; }   // waves 10,11,14,15 -> 8..15
  }

  int d_b0[2], d_b1[2];   // per D item of this wave: within-row byte offsets of window column pair 0 (pair 4 = + 16) and 1 (pairs 2, 3 = + 2, 4 planes)
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    int tx = 16 * ((d_first + i) & 3) + d_tsel;
    tx = tx < 50 ? tx : 49;
    d_b0[i] = (tx + 1) * 16 + ((lane >> 5) ? 0 : (7 * T_PLANE - 1) * 16);
    d_b1[i] = (tx + 1) * 16 + ((lane >> 5) ? 2 : 1) * T_PLANE * 16;
  }
  unsigned bad_wave = 0;  // non-finite outputs zeroed by the fast path (wave-uniform count)
  // BC item of this wave (bc_item = 2 * pixel tile + ConvT#3 row tile): the lane's 100-level pixel never changes
  int bc_idx = 32 * (bc_item >= 0 ? bc_item >> 1 : 0) + l31;
  const bool bc_valid = bc_idx < 200;
  bc_idx = bc_valid ? bc_idx : 199;
  const bool bc_a = bc_idx >= 100;
  const int bc_x = bc_idx - (bc_a ? 100 : 0);
  const int bc_l0 = l100_off(bc_a ? 1 : 0, bc_x, h);                                              // chunk h; chunk 2 + h is 2 * 212 granules further
  char* const bc_wbase = ring + ((((bc_x & 1) * (4 * T_PLANE) + (bc_x >> 1) + 1) << 4) + 8 * h);  // x = 4 bc_x + c: plane c + 4 (bc_x & 1), granule (bc_x >> 1) + 1

  const int a_px = 32 * a_ct + l31;
  const bool a_valid = a_px < 50;
  const int a_pxc = a_valid ? a_px : 49;
  const unsigned a_lane = (unsigned)(a_pxc * 64 + 8 * h);                       // element offset of this lane's 16-byte pieces inside a 50-level row
  const int a_dst = l100_off(a_mt >> 1, 2 * (a_valid ? a_px : 0) + (a_mt & 1), 0) + 8 * h;   // L100 granule of chunk 0; chunk q is q * 212 granules further
  uint4 wa[4], xb[4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) { wa[kk] = make_uint4(0, 0, 0, 0); xb[kk] = make_uint4(0, 0, 0, 0); }
  if (hasA) {
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) wa[kk] = w2_f[(a_mt * 4 + kk) * 64 + lane];
  }

  const bool ab_sw = TAIL_ABL(1);
  // This workgroup's samples k = 0..K-1 (ids blockIdx.x + k*gridDim.x) are treated as one tall image of
  // 400*K rows: strip g = 50*k + s.  Round r runs A(g=r), BC(g=r-1), D(g=r-2), so the pipeline never
  // drains between samples; D(g) covers rows 8g-1 .. 8g+6 of the tall image, and its first row pair at a
  // sample seam (last row of sample k-1, first row of sample k) is evaluated once per side of the seam.
  // Small batches leave most CUs idle when a workgroup needs a whole sample, so a sample can be cut into S = p.seg
  // segments of L = 50 / S strips ("virtual samples", id = sample * S + segment).  A segment is preceded by one
  // warm-up strip (A and BC only) that refills the two ring rows its first output row pair reads; for segment 0 that
  // strip lies above the image and is skipped (the top row pair takes zeros, as before).  S = 1 is the plain layout.
  const int S = (SEG && p.seg > 1) ? p.seg : 1, L = 50 / S, SL = S > 1 ? L + 1 : 50;  // SEG == false: compile-time S = 1, the plain index math
  const int NV = p.n * S;
  const int K = ((int)blockIdx.x < NV) ? (NV - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int G = SL * K;
  // strip g of this workgroup's tall image -> (sample, actual strip s in -1..49)
  auto strip_of = [&](const int g, int& smp, int& sidx, int& seg_out, int& sl_out) {
    const int kv = g / SL, sl = g - SL * kv;
    const int vs = (int)blockIdx.x + kv * (int)gridDim.x;
    smp = vs / S;
    const int seg = vs - smp * S;
    sidx = S > 1 ? seg * L + sl - 1 : sl;
    seg_out = seg; sl_out = sl;
  };
  if (hasA && K > 0) {
    int smp0, s0, sg0, sl0;
    strip_of(0, smp0, s0, sg0, sl0);
    if (s0 >= 0) {
      const uint16_t* src = p.in + ((size_t)smp0 * 50 + s0) * 3200 + a_lane;   // wave-uniform row + the lane's constant offset
#pragma unroll
      for (int kk = 0; kk < 4; ++kk) xb[kk] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4nt*>(src + 16 * kk)));
    }
  }
  float o_mean = 0.f, o_std = 1.f, o_mean_prev = 0.f, o_std_prev = 1.f;  // de-standardisation of D's sample / the one before
  unsigned bad_count = 0;

  unsigned long long tD = 0, tBC = 0, tA = 0, tBar = 0, tStart = 0, ts[6];
  if (PROF) tStart = __builtin_amdgcn_s_memtime();
  for (int r = 0; r <= G + 2 && K > 0; ++r) {
    // ---------------- A: ConvT#2 for strip g = r (input prefetched last round), then prefetch g+1 ----------------
    // Runs BEFORE this wave's D items (waves 0-7: BC -> A -> D).  vmcnt is one in-order counter for loads and stores: waiting
    // for the prefetched input right after D's global stores (the old order BC -> D -> A) made the wave sit out the completion
    // of stores it had issued a few hundred cycles earlier, at the end of every round, in front of the barrier.  With A first
    // the youngest operations ahead of the wait are last round's stores -- long done -- and this round's stores have until
    // the next round's A.
    unsigned long long a_span = 0;   // diagnostic build: cycles of this round's A stage (it runs inside the D section for waves 0-7)
    auto run_a = [&]() {
      const unsigned long long a_t0 = PROF ? __builtin_amdgcn_s_memtime() : 0;
      int a_smp = 0, a_s = -1, a_sg = 0, a_sl = 0;
      if (r < G) strip_of(r, a_smp, a_s, a_sg, a_sl);
      if (hasA && r < G && !TAIL_ABL(4)) {
        int n_smp = 0, n_s = -1, n_sg = 0, n_sl = 0;
        if (r + 1 < G) strip_of(r + 1, n_smp, n_s, n_sg, n_sl);
        if (a_s < 0) {        // warm-up strip above the image: nothing to compute, only fetch the next strip's input
          if (n_s >= 0) {
            const uint16_t* src = p.in + ((size_t)n_smp * 50 + n_s) * 3200 + a_lane;
  #pragma unroll
            for (int kk = 0; kk < 4; ++kk) xb[kk] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4nt*>(src + 16 * kk)));
          }
        } else {
        f32x16 acc = load_bias16(cst + TC_OFF_B2 + h * 64);
  #pragma unroll
        for (int kk = 0; kk < 4; ++kk) acc = mfma32<F16>(wa[kk], xb[kk], acc);
        if (n_s >= 0) {
          const uint16_t* src = p.in + ((size_t)n_smp * 50 + n_s) * 3200 + a_lane;
  #pragma unroll
          for (int kk = 0; kk < 4; ++kk) xb[kk] = __builtin_bit_cast(uint4, __builtin_nontemporal_load(reinterpret_cast<const u32x4nt*>(src + 16 * kk)));
        }
        uint32_t f2[8];
        swish_pack16<F16>(acc, f2, ab_sw);
        if (a_valid) {
          char* dst = l100 + (r & 1) * T_L100_BUF + a_dst;
  #pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(dst + q * (212 * 16)) = make_uint2(f2[2 * q], f2[2 * q + 1]);
        }
        }
      }
      if (PROF) a_span = __builtin_amdgcn_s_memtime() - a_t0;
    };
    // ---------------- BC: ConvT#3 + ConvT#4 on 32 pixels of the 100-level ----------------
    auto do_bc = [&]() {
      const int g = r - 1;
      if (!(bc_item >= 0 && g >= 0 && g < G && !TAIL_ABL(8))) return;
      { int sm, sx, sg, sl; strip_of(g, sm, sx, sg, sl); if (sx < 0) return; }  // warm-up strip of a top segment: nothing above the image
      const int m3 = bc_item & 1;
      const char* src = l100 + (g & 1) * T_L100_BUF;
      uint4 b0 = *reinterpret_cast<const uint4*>(src + bc_l0), b1 = *reinterpret_cast<const uint4*>(src + bc_l0 + 2 * 212 * 16);
      f32x16 acc3 = load_bias16(cst + TC_OFF_B3 + h * 64);
      acc3 = mfma32<F16>(w3_f[(m3 * 2 + 0) * 64 + lane], b0, acc3);
      acc3 = mfma32<F16>(w3_f[(m3 * 2 + 1) * 64 + lane], b1, acc3);
      uint32_t f3[8];
      const uint4 w4 = w4_f[lane];
      // the item writes ring rows 8g + 4a + 2 m3 + {0, 1} (a = 100-level row of the lane's pixel; the ConvT#3 tap row is m3
      // for both taps tt): two wave-uniform row offsets per value of a, picked per lane, + the lane's constant granule
      const int rbase = (8 * g) % T_RING_ROWS;
      int ra0 = rbase + 2 * m3, ra1 = ra0 + 1, rb0 = ra0 + 4, rb1 = ra0 + 5;
      ra0 = ra0 >= T_RING_ROWS ? ra0 - T_RING_ROWS : ra0; ra1 = ra1 >= T_RING_ROWS ? ra1 - T_RING_ROWS : ra1;
      rb0 = rb0 >= T_RING_ROWS ? rb0 - T_RING_ROWS : rb0; rb1 = rb1 >= T_RING_ROWS ? rb1 - T_RING_ROWS : rb1;
      char* w0 = bc_wbase + (bc_a ? rb0 : ra0) * T_ROWP;
      char* w1 = bc_wbase + (bc_a ? rb1 : ra1) * T_ROWP;
      // Round 3: ConvT#4's two MFMAs ride INSIDE this wave's own swish stream (tools/microbench9.hip: a v_mfma_f32_32x32x16 between the
      // transcendentals of the same wave costs the SIMD 3.6 ns, in front of its own swish block and waited for -- round 2 -- 10.7 ns).
      // Accumulator registers 0-7 of ConvT#3 are tap 0's sixteen channels (k order of w4), 8-15 tap 1's: activate the first half, start
      // tap 0's MFMA, activate the second half under it, start tap 1's, activate tap 0's result under that.  Same instructions per
      // element as the round-2 order: bit-identical.
      f32x16 acc4a, acc4b;
      if (ab_sw) {       // diagnostic (no swish): the round-2 order
        swish_pack16<F16>(acc3, f3, true);
        acc4a = mfma32<F16>(w4, make_uint4(f3[0], f3[1], f3[2], f3[3]), load_bias16(cst + TC_OFF_B4 + h * 64));
        acc4b = mfma32<F16>(w4, make_uint4(f3[4], f3[5], f3[6], f3[7]), load_bias16(cst + TC_OFF_B4 + h * 64));
      } else {
        swish_pack_h<F16, 0, 8>(acc3, f3);
        pin();
        acc4a = mfma32<F16>(w4, make_uint4(f3[0], f3[1], f3[2], f3[3]), load_bias16(cst + TC_OFF_B4 + h * 64));
        pin();
        swish_pack_h<F16, 8, 8>(acc3, f3 + 4);
        pin();
        acc4b = mfma32<F16>(w4, make_uint4(f3[4], f3[5], f3[6], f3[7]), load_bias16(cst + TC_OFF_B4 + h * 64));
        pin();
      }
      auto ring_store = [&](const uint32_t (&f4)[8], const int tt) {
        if (bc_valid) {
#pragma unroll
          for (int q = 0; q < 4; ++q)   // register pair q: row q >> 1 of the tap's 2x2 block, plane 2 tt + (q & 1)
            *reinterpret_cast<uint2*>(((q >> 1) ? w1 : w0) + (2 * tt + (q & 1)) * (T_PLANE * 16)) = make_uint2(f4[2 * q], f4[2 * q + 1]);
        }
      };
      uint32_t f4[8];
      swish_pack16<F16>(acc4a, f4, ab_sw);
      ring_store(f4, 0);
      swish_pack16<F16>(acc4b, f4, ab_sw);
      ring_store(f4, 1);
    };

    // ---------------- D: output conv of strip g = r-2: tall-image rows 8g-1 .. 8g+6 ----------------
    // One item = one output row pair (rp, wave-uniform) x 16 tiles of 2x8 pixels, so row slots and
    // validity live in SGPRs and each lane only adds its own column offset.
    const int gd = r - 2;
    const bool d_on = gd >= 0 && !TAIL_ABL(2);
    const int kd = gd >= 0 ? gd / SL : 0, sd = gd - SL * kd;   // virtual sample / local strip of D's strip
    int sample_d = 0, s_d = 0, seg_d = 0, sl_d = 0;            // real sample and actual strip (valid while gd < G)
    if (gd >= 0 && gd < G) strip_of(gd, sample_d, s_d, seg_d, sl_d);
    // the virtual sample before this one (the last one, in the flush round) ended a real sample iff it was segment S-1
    int prev_smp = 0; bool prev_ends = false;
    if (gd >= 0 && sd == 0 && kd > 0) {
      const int pv = (int)blockIdx.x + (kd - 1) * (int)gridDim.x;
      prev_smp = pv / S;
      prev_ends = pv - prev_smp * S == S - 1;
    }
    if (d_on && sd == 0 && p.aff_out) {  // entering a virtual sample: rotate the de-standardisation scalars (SGPRs)
      o_mean_prev = o_mean; o_std_prev = o_std;
      if (gd < G) {
        o_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.aff_out[2 * sample_d])));
        o_std = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.aff_out[2 * sample_d + 1])));
      }
    }
    const bool d_warm = gd == G || (S > 1 && sd == 0);   // no regular output rows: flush round, or a segment's warm-up strip
    const bool d_top = !d_warm && s_d == 0;              // first strip of a real sample: row pair 0 = (nothing | row 0)
    {
      // fast path: row pair rp of strip gd, all four window rows inside one sample
      auto d_fast = [&](const int it, const int rowoff0, const int rowoff1) {
        const int item = d_first + it, rp = item >> 2, j4 = item & 3;
        if (TAIL_ABL(64) && j4 == 3) return;   // diagnostic: what the 2-of-16-tile items at the end of every row pair cost
        f32x4 acc = {conv_bias, conv_bias, conv_bias, conv_bias};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const int rb = half ? rowoff1 : rowoff0;
          const char* p0 = smem + rb + d_b0[it];
          const char* p1 = smem + rb + d_b1[it];
          uint4 av[5], wv[5];
          av[0] = *reinterpret_cast<const uint4*>(p0);
          av[4] = *reinterpret_cast<const uint4*>(p0 + 16);
          av[1] = *reinterpret_cast<const uint4*>(p1);
          av[2] = *reinterpret_cast<const uint4*>(p1 + 2 * T_PLANE * 16);
          av[3] = *reinterpret_cast<const uint4*>(p1 + 4 * T_PLANE * 16);
#pragma unroll
          for (int cp = 0; cp < 5; ++cp) wv[cp] = wc_f[(half * 5 + cp) * 64 + lane];
          // Operands SWAPPED against the seam path (round 3): the Toeplitz weights are the A operand (rows = the tile's 16 pixels),
          // the window activations the B operand (columns = the item's 16 tiles).  Both operands keep their per-lane registers
          // (lane & 15 indexes the non-k dimension of either), only the accumulator is transposed: register rr of lane group kg
          // is now pixel 4 kg + rr of tile (lane & 15) -- four CONSECUTIVE pixels of one output row, one 16-byte store per lane
          // instead of four 4-byte stores 32 pixels apart.  Same products, same k order per output: bit-identical to the seam path.
#pragma unroll
          for (int cp = 0; cp < 5; ++cp) acc = mfma16<F16>(wv[cp], av[cp], acc);
          __builtin_amdgcn_sched_barrier(0);
        }
        // lane (n = lane & 15, kg): tile 16 j4 + d_tsel(n), output row oy = kg >> 1 of the pair, pixels 4 (kg & 1) .. + 3 of the tile
        char* orow = reinterpret_cast<char*>(p.out) + (((size_t)sample_d * 400 + (8 * s_d - 1 + 2 * rp)) * 400 + 128 * j4) * OUTSZ;
        float v[4];
        // de-standardise with ONE fma per value on this 16-bit path (the f32 parity path keeps numpy's two roundings, kernels_tail32.hip):
        // <= 1 ulp from y * std + mean, four instructions instead of two packed multiplies + two packed adds + four selects
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) v[rr] = __builtin_fmaf(acc[rr], o_std, o_mean);   // without aff_out: std = 1, mean = 0 -> v exactly
        const bool lane_on = j4 < 3 || d_last_on;   // item 3 of a row pair holds tiles 48, 49 only
        if (p.nan_guard) {
          // ONE question of the four values first (0 * v summed is NaN iff one of them is not finite: 3 fma + 1 multiply + 1 compare);
          // the per-value zero-fill and count run only then
          const float t = __builtin_fmaf(v[3], 0.f, __builtin_fmaf(v[2], 0.f, __builtin_fmaf(v[1], 0.f, v[0] * 0.f)));
          if (__ballot(t != t) != 0ull) {
            const unsigned long long on = __ballot(lane_on);
            unsigned nbad = 0;
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const bool ok = fabsf(v[rr]) <= 3.402823466e38f;
              nbad += (unsigned)__popcll(~__ballot(ok) & on);
              v[rr] = ok ? v[rr] : 0.f;
            }
            bad_wave += nbad;
          }
        }
        if (lane_on) {
          char* o = orow + d_olane;
          // the image is written once and never read back on the device: non-temporal stores keep it out of the way of what IS
          // re-read (weights, the next launch's operands); with the non-temporal loads of the input: 1.6 % of the step, and mid16 in front
          // of this kernel runs 2 % faster (profiles/r04/r_nontemporal_ab.txt)
          typedef float f32x4nt __attribute__((ext_vector_type(4)));
          typedef unsigned u32x2nt __attribute__((ext_vector_type(2)));
          if (OUT == 0) __builtin_nontemporal_store(f32x4nt{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4nt*>(o));
          else if (OUT == 1) __builtin_nontemporal_store(u32x2nt{pack2<false>(v[0], v[1]), pack2<false>(v[2], v[3])}, reinterpret_cast<u32x2nt*>(o));
          else __builtin_nontemporal_store(u32x2nt{pack2<true>(v[0], v[1]), pack2<true>(v[2], v[3])}, reinterpret_cast<u32x2nt*>(o));
        }
      };
      auto do_d = [&](const int item) {  // seam rows (sample / segment boundaries): called (not looped) so no conservative vmcnt(0) lands in front of it
        const int rp = item >> 2, j4 = item & 3;
        // rp == 0 at a boundary: the pair is (row 399 of the sample that just ended | row 0 of the one that starts)
        const bool emit_prev = rp == 0 && sd == 0 && prev_ends;
        const bool emit_top = rp == 0 && d_top;
        const bool seam = emit_prev || emit_top || d_warm;
        if (d_warm && !emit_prev) return;              // flush round / warm-up strip: at most the ended sample's row 399
        const int kg = lane >> 4;
        const int d_dy = (lane >> 4) & 1, d_dx = lane >> 5;   // window sub-pixel (dy, dx) of this lane's k-group
        int d_xo[5];                                           // byte offsets of the five window column pairs relative to granule (plane 0, tx)
#pragma unroll
        for (int cp = 0; cp < 5; ++cp)
          d_xo[cp] = 16 * (d_dx ? (cp < 4 ? 2 * cp * T_PLANE : 1) : (cp == 0 ? 7 * T_PLANE - 1 : (2 * cp - 1) * T_PLANE));
        int tx = 16 * j4 + d_tsel;
        tx = tx < 50 ? tx : 49;
        const int sa = (8 * gd + 16 + 2 * rp) % T_RING_ROWS;  // slot of the first window row (tall row 8g-2+2rp)
        int s0 = sa + d_dy, s1 = s0 + 2;
        s0 = s0 >= T_RING_ROWS ? s0 - T_RING_ROWS : s0;
        s1 = s1 >= T_RING_ROWS ? s1 - T_RING_ROWS : s1;
        const int r0 = T_OFF_RING + s0 * T_ROWP + (tx + 1) * 16, r1 = T_OFF_RING + s1 * T_ROWP + (tx + 1) * 16;
        const bool okf = !(d_dx == 0 && tx == 0), okl = !(d_dx == 1 && tx == 49);
        const int oy = (lane >> 3) & 1, ox = lane & 7;
        const int txo = 16 * j4 + kg;
        // window rows 0,1 (k-steps 0-4) / 2,3 (k-steps 5-9); side 0 keeps rows 0,1 (sample kd-1), side 1 rows 2,3
        auto conv = [&](const bool keep01, const bool keep23) {
          f32x4 acc = {conv_bias, conv_bias, conv_bias, conv_bias};
#pragma unroll
          for (int half = 0; half < 2; ++half) {  // two batches of five k-steps keep the live operand set at 40 VGPRs
            uint4 av[5], wv[5];
#pragma unroll
            for (int cp = 0; cp < 5; ++cp) {
              bool ok = half ? keep23 : keep01;
              if (cp == 0) ok = ok && okf;
              if (cp == 4) ok = ok && okl;
              const int off = ok ? (half ? r1 : r0) + d_xo[cp] : T_OFF_ZERO;
              av[cp] = *reinterpret_cast<const uint4*>(smem + off);
              wv[cp] = wc_f[(half * 5 + cp) * 64 + lane];
            }
#pragma unroll
            for (int cp = 0; cp < 5; ++cp) acc = mfma16<F16>(av[cp], wv[cp], acc);
            __builtin_amdgcn_sched_barrier(0);
          }
          return acc;
        };
        // accumulator register rr of lane group kg holds tile 16*j4 + 4rr + kg, pixel (oy,ox) = lane & 15
        auto store = [&](const f32x4& acc, const int smp, const int Y, const float mean, const float sdv, const bool lane_on) {
          if (!lane_on) return;
          const size_t o0 = ((size_t)smp * 400 + Y) * 400 + 8 * txo + ox;
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            if (txo + 4 * rr >= 50) continue;
            float v = acc[rr];
            if (p.aff_out) v = __builtin_fmaf(v, sdv, mean);   // one fma, as the fast path: every row of an image goes through the same arithmetic whatever the segmentation
            if (p.nan_guard && !(fabsf(v) <= 3.402823466e38f)) { v = 0.f; ++bad_count; }
            const size_t o = o0 + 32 * rr;
            if (OUT == 0) reinterpret_cast<float*>(p.out)[o] = v;
            else if (OUT == 1) reinterpret_cast<uint16_t*>(p.out)[o] = (uint16_t)(pack2<false>(v, 0.f) & 0xffff);
            else reinterpret_cast<uint16_t*>(p.out)[o] = (uint16_t)(pack2<true>(v, 0.f) & 0xffff);
          }
        };
        if (!seam) {
          const f32x4 acc = conv(true, true);
          store(acc, sample_d, 8 * s_d - 1 + 2 * rp + oy, o_mean, o_std, true);
        } else {
          if (emit_prev) {   // last row of the sample that ended: rows 398,399 | zeros
            const f32x4 acc = conv(true, false);
            store(acc, prev_smp, 399, o_mean_prev, o_std_prev, oy == 0);
          }
          if (emit_top) {    // first row of this sample: zeros | rows 0,1
            const f32x4 acc = conv(false, true);
            store(acc, sample_d, 0, o_mean, o_std, oy == 1);
          }
        }
      };
      auto run_d = [&]() {
        if (!d_on || d_cnt < 1) return;
        const int rp = d_first >> 2;    // both items of a wave sit in the same row pair
        // wave-uniform: no regular rows at all (flush round / a segment's warm-up strip), or row pair 0 of a strip whose row
        // above belongs to another sample or lies above the image (first strip of a virtual sample; strip 0 of a real one)
        const bool seam_round = d_warm || (rp == 0 && (sd == 0 || s_d == 0));
        if (seam_round) {
          do_d(d_first);
          if (d_cnt >= 2) do_d(d_first + 1);
          return;
        }
        // ring byte offsets of window rows (d_dy) and (d_dy + 2) of the row pair: uniform slot + lane row, wrapped at 18 rows
        // (unsigned min: v - 18 rows wraps around to a huge value unless v is past the end)
        const int sa = (8 * gd + 16 + 2 * rp) % T_RING_ROWS;
        unsigned v0 = (unsigned)(sa * T_ROWP + d_rowlane);
        v0 = min(v0, v0 - (unsigned)(T_RING_ROWS * T_ROWP));
        unsigned v1 = v0 + 2u * T_ROWP;
        v1 = min(v1, v1 - (unsigned)(T_RING_ROWS * T_ROWP));
        d_fast(0, (int)v0, (int)v1);
        if (d_cnt >= 2) d_fast(1, (int)v0, (int)v1);
      };
      // stagger: the BC-only-plus-D waves do their latency-bound D items first, so their VALU-heavy
      // BC items overlap the tail (A items) of the waves that started with BC
      if (PROF) ts[0] = __builtin_amdgcn_s_memtime();
      const bool d_first_order = TAIL_ABL(16) ? false : (TAIL_ABL(32) ? true : wave >= 8);
      if (d_first_order) {
        if (!TAIL_ABL(128)) __builtin_amdgcn_s_setprio(3);  // D is a latency chain with few instructions: let it through
        run_d();
        __builtin_amdgcn_s_setprio(0);
      }
      if (PROF) ts[1] = __builtin_amdgcn_s_memtime();
      do_bc();
      if (PROF) ts[2] = __builtin_amdgcn_s_memtime();
      if (!d_first_order) {
        run_a();
        if (!TAIL_ABL(128)) __builtin_amdgcn_s_setprio(3);
        run_d();
        __builtin_amdgcn_s_setprio(0);
      }
      if (PROF) ts[3] = __builtin_amdgcn_s_memtime();
      if (d_first_order) run_a();   // (waves 8-15 have no A items; the diagnostic orders keep the call)
    }

    if (PROF) ts[4] = __builtin_amdgcn_s_memtime();
    lds_barrier();
    if (PROF) {
      ts[5] = __builtin_amdgcn_s_memtime();
      const bool a_in_d = !(TAIL_ABL(16) ? false : (TAIL_ABL(32) ? true : wave >= 8));   // A ran between BC and D (inside ts[2]..ts[3])
      tD += (ts[1] - ts[0]) + (ts[3] - ts[2]) - (a_in_d ? a_span : 0); tBC += ts[2] - ts[1]; tA += a_span; tBar += ts[5] - ts[4];
    }
  }
  if (PROF && p.prof && blockIdx.x == 0 && lane == 0) {
    unsigned long long* o = p.prof + wave * 5;
    o[0] = tD; o[1] = tBC; o[2] = tA; o[3] = tBar; o[4] = __builtin_amdgcn_s_memtime() - tStart;
  }
  if (p.nan_guard && p.nonfinite && bad_count) atomicAdd(p.nonfinite, (unsigned long long)bad_count);
  if (p.nan_guard && p.nonfinite && bad_wave && lane == 0) atomicAdd(p.nonfinite, (unsigned long long)bad_wave);
}

// ---------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------
hipError_t launch_enc_conv1_16(bool f16, const float* x, const float* affine, const float* w, const float* b, uint16_t* y, int n, hipStream_t s) {
  if (n == 0) return hipSuccess;
  dim3 grid((n * 25 * 16 + 255) / 256);
  if (f16) hipLaunchKernelGGL(enc_conv1_16<true>, grid, dim3(256), 0, s, x, affine, w, b, y, n);
  else hipLaunchKernelGGL(enc_conv1_16<false>, grid, dim3(256), 0, s, x, affine, w, b, y, n);
  return hipGetLastError();
}

static int gemm16_lds(int bn) { return (G_BP + bn) * G_PITCH * 2 + 3 * G_BP * 4; }

hipError_t launch_gemm16(bool f16, const GemmDesc& d, const uint16_t* X, const uint16_t* Wt, int Kpad, const float* bias, uint16_t* Y,
                         float* part, int splits, hipStream_t s) {
  if (d.M == 0) return hipSuccess;
  const bool wide = d.Npad % 128 == 0;
  const int bn = wide ? 128 : 64;
  const bool sk = splits > 1 && part != nullptr;
  int kslice = d.K;
  if (sk) kslice = ((d.K / G_BK + splits - 1) / splits) * G_BK;
  const int nz = sk ? (d.K + kslice - 1) / kslice : 1;
  dim3 grid((d.M + G_BP - 1) / G_BP, d.Npad / bn, nz);
  void (*fn)(GemmDesc, const uint16_t*, const uint16_t*, int, const float*, uint16_t*, float*, int) = nullptr;
  int slot = (f16 ? 4 : 0) + (wide ? 2 : 0) + (sk ? 1 : 0);
  switch (slot) {
    case 0: fn = gemm16<false, 64, false>; break;
    case 1: fn = gemm16<false, 64, true>; break;
    case 2: fn = gemm16<false, 128, false>; break;
    case 3: fn = gemm16<false, 128, true>; break;
    case 4: fn = gemm16<true, 64, false>; break;
    case 5: fn = gemm16<true, 64, true>; break;
    case 6: fn = gemm16<true, 128, false>; break;
    default: fn = gemm16<true, 128, true>; break;
  }
  const int lds = gemm16_lds(bn);
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(fn), lds);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, grid, dim3(256), lds, s, d, X, Wt, Kpad, bias, Y, part, kslice);
  e = hipGetLastError();
  if (e != hipSuccess || !sk) return e;
  const int total = d.M * (d.N / 4);
  if (f16) hipLaunchKernelGGL(splitk_finish16<true>, dim3((total + 255) / 256), dim3(256), 0, s, part, nz, bias, Y, d.M, d.N, d.Npad, d.OC, d.act);
  else hipLaunchKernelGGL(splitk_finish16<false>, dim3((total + 255) / 256), dim3(256), 0, s, part, nz, bias, Y, d.M, d.N, d.Npad, d.OC, d.act);
  return hipGetLastError();
}

int tail_lds_bytes() { return T_LDS_BYTES; }

hipError_t launch_tail16(bool f16, const TailParams& p, int blocks, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  void (*fn)(TailParams) = nullptr;
  const bool seg = p.seg > 1;
#define PICK(F, O) fn = seg ? tail16<F, O, false, true> : tail16<F, O, false, false>
  if (f16) { if (p.out_dtype == SRCFD_F32) PICK(true, 0); else if (p.out_dtype == SRCFD_BF16) PICK(true, 1); else PICK(true, 2); }
  else { if (p.out_dtype == SRCFD_F32) PICK(false, 0); else if (p.out_dtype == SRCFD_BF16) PICK(false, 1); else PICK(false, 2); }
#undef PICK
  if (p.prof && !f16 && p.out_dtype == SRCFD_F32) fn = seg ? tail16<false, 0, true, true> : tail16<false, 0, true, false>;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(fn), T_LDS_BYTES);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(1024), T_LDS_BYTES, s, p);
  return hipGetLastError();
}

}  // namespace srcfd
