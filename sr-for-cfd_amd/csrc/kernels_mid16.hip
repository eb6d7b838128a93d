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

#ifdef SRCFD_DIAG
#define MID_ABL(bit) (p.ablate & (bit))
#else
#define MID_ABL(bit) 0   // diagnostic switches exist only in a -DSRCFD_DIAG build
#endif

constexpr int M_PITCH = 72;  // LDS row pitch in elements (144 B)

// NW waves per workgroup, each owning 32 output pixels x 128 channels.  The weight tile (16 KB per
// stage) is shared by all NW waves, so bytes pulled through L2 per output pixel fall as 1/NW:
// with 4 waves this kernel sat at ~17 GB/s per CU of operand traffic (the per-CU load rate), not on the MFMAs.
template <int NW> struct MidCfg {
  static constexpr int PX = 32 * NW;                    // output pixels per workgroup
  // input-pixel slab bound, rows: a 12-wide phase packs PX pixels into PX/12+1 output rows (one input row each),
  // plus the row above the first one and one row of slack for the partial first/last rows
  static constexpr int PATCH = (PX / 12 + 3) * 12 + 16;
  static constexpr int OFF_W = (PATCH + 1) * M_PITCH * 2;       // bytes; row PATCH is all zero
  static constexpr int MAIN_END = OFF_W + 2 * 16384;            // two weight tiles of [128 rows][64 k] at 128 B per row (XOR-swizzled)
  static constexpr int T1_END = 32768 + NW * 32 * 144;          // ConvT#1 stage: half of its operands (4 tiles x 8 KB) + per-wave store tiles
  static constexpr int OFF_META = MAIN_END > T1_END ? MAIN_END : T1_END;
  static constexpr int LDS = OFF_META + 3 * PX * 4;
  static constexpr int NTHR = 64 * NW;
  static constexpr int WCH = 1024 / NTHR;               // weight-tile 16-byte chunks per thread
  static constexpr int PCH = (PATCH * 8 + NTHR - 1) / NTHR;  // patch chunks per thread
};

template <bool F16, int NW>
__global__ void __launch_bounds__(64 * NW, NW == 8 ? 4 : 1) mid16(MidParams p) {
  using C = MidCfg<NW>;
  static_assert(C::OFF_META >= 32768 + NW * 32 * 144, "ConvT#1 operand tiles (4 x 8 KB) + per-wave store tiles are staged over the patch + weight tiles");
  extern __shared__ __attribute__((aligned(16))) char msm[];
  uint16_t* Ps = reinterpret_cast<uint16_t*>(msm);
  uint16_t* Ws = reinterpret_cast<uint16_t*>(msm + C::OFF_W);
  int* row_img = reinterpret_cast<int*>(msm + C::OFF_META);
  int* row_my = row_img + C::PX;
  int* row_mx = row_my + C::PX;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, l31 = lane & 31;
  const int phase = blockIdx.y, blk = blockIdx.x;
#ifdef SRCFD_DIAG
  unsigned long long tstamp[6];
  tstamp[0] = __builtin_amdgcn_s_memtime();
#define MID_STAMP(i) do { if (p.prof) { __builtin_amdgcn_sched_barrier(0); tstamp[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define MID_STAMP(i) do { } while (0)
#endif
  const int py = phase >> 1, px = phase & 1;
  const int TY = py ? 1 : 2, TX = px ? 1 : 2, NT = TY * TX;
  const int MH = py ? 12 : 13, MW = px ? 12 : 13, per = MH * MW;
  const int M = p.n * per, m0 = blk * C::PX;
  if (m0 >= M) return;
  const uint16_t* Wt = p.w0[phase];
  const int Kp = p.kpad[phase];

  if (tid < C::PX) {
    int m = m0 + tid, img = -1, my = 0, mx = 0;
    if (m < M) { img = m / per; int r = m - img * per; my = r / MW; mx = r - my * MW; }
    row_img[tid] = img; row_my[tid] = my; row_mx[tid] = mx;
  }
  if (tid < M_PITCH / 2) reinterpret_cast<uint32_t*>(Ps + C::PATCH * M_PITCH)[tid] = 0;
  __syncthreads();

  MID_STAMP(1);
  // contiguous slab of input pixels (tensor index img*144 + iy*12 + ix) this workgroup touches
  const int mlast = min(m0 + C::PX - 1, M - 1) - m0;
  const int lo = row_img[0] * 144 + max(row_my[0] - 1, 0) * 12;
  const int hi = row_img[mlast] * 144 + min(row_my[mlast], 11) * 12 + 11;
  const int NP = min(hi - lo + 1, C::PATCH);  // <= PATCH by construction; the clamp only guards LDS

  // this lane's pixel and the patch row each tap reads
  const int prow = wave * 32 + l31;
  const int img = row_img[prow], my = row_my[prow], mx = row_mx[prow];
  int trow[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int ty = t / TX, tx = t - ty * TX;  // only t < NT is used
    const int iy = my - ty, ix = mx - tx;
    int r = img * 144 + iy * 12 + ix - lo;
    const bool ok = img >= 0 && t < NT && (unsigned)iy < 12u && (unsigned)ix < 12u && (unsigned)r < (unsigned)C::PATCH;
    trow[t] = (ok ? r : C::PATCH) * M_PITCH + h * 8;
  }

  // staging roles: 16-byte column c8 of rows xrow + (NTHR/8)*j
  constexpr int RSTEP = C::NTHR / 8;
  const int xrow = tid >> 3, c8 = tid & 7;
  // native vector type: arrays of HIP's uint4 struct were left in scratch by the compiler (store/reload around every
  // prefetch), which serialised the global loads
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pr[C::PCH];
  // Weight tiles go global -> LDS directly (global_load_lds: no registers, no ds_write pass), two buffers.  The LDS image
  // is lane-linear (chunk index = row * 8 + slot = tid + NTHR * j), so the bank swizzle sits on the SOURCE address: slot
  // `sl` of row r holds k-chunk sl ^ ((r >> 1) & 7), and the fragment reads below apply the same XOR.
  // The builtin form of the instruction is known to hipcc as a pending LDS write: it then puts `s_waitcnt vmcnt(0)` in front of
  // the next ds_read -- here the fragment reads of the tile being COMPUTED -- which drains the prefetch before the MFMAs start
  // (seen in the ISA; the kernel gained nothing from the prefetch).  As an asm statement the load is outside hipcc's
  // bookkeeping: the stage loop waits for it itself (vmcnt(0) in front of the barrier that publishes the tile).
  const unsigned lds_w0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(msm + C::OFF_W);
  // Addresses of the staging loads are a wave-uniform base (SGPR pair) + a 32-bit per-lane offset: as 64-bit per-lane pointers (round 2)
  // the five patch pointers alone took ten registers, hipcc spilled them, and every reload in the loop came with an
  // s_waitcnt vmcnt(0) that drained the weight tile in flight (seen in the ISA, round 3)
  unsigned woff[C::WCH];
#pragma unroll
  for (int j = 0; j < C::WCH; ++j) {
    const int r = xrow + RSTEP * j;
    woff[j] = (unsigned)((r * Kp + ((c8 ^ ((r >> 1) & 7)) * 8)) * 2);
  }
  auto g2l_w = [&](int t, int c, int buf) {
    const uint16_t* base = Wt + t * 256 + c * 64;     // wave-uniform
#pragma unroll
    for (int j = 0; j < C::WCH; ++j) {
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_w0 + (unsigned)(buf * 16384 + (wave * 64 + C::NTHR * j) * 16));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"(woff[j]), "s"(dst), "s"(base) : "memory");
    }
  };
  const unsigned poff0 = (unsigned)(xrow * 512 + c8 * 16);   // byte offset of this thread's first patch chunk inside the slab
  auto g2r_p = [&](int c) {
    const char* base = reinterpret_cast<const char*>(p.in + (size_t)lo * 256 + c * 64);   // wave-uniform
#pragma unroll
    for (int j = 0; j < C::PCH; ++j) {
      const int r = xrow + RSTEP * j;
      pr[j] = r < NP ? *reinterpret_cast<const u32x4*>(base + (poff0 + (unsigned)(RSTEP * j * 512))) : u32x4{0, 0, 0, 0};
    }
  };
  // ConvT#1's 64 KB of A operands (8 tiles of 8 KB = 512 chunks of 16 B, one per (tap, 32-channel half)) go through the LDS
  // the main loop has left, in two halves of four tiles, by global_load_lds (no registers: the accumulators, the packed
  // B operands and the swish temporaries fill the file in that stage); the waves work through a half without
  // synchronising with each other
  constexpr int W1CH = 2048 / C::NTHR;
  static_assert(W1CH * C::NTHR == 2048, "a half of ConvT#1's operands divides evenly over the workgroup");
  const u32x4* w1g = reinterpret_cast<const u32x4*>(p.w1f);
  auto g2l_w1 = [&](int half) {   // lane-linear LDS image: wave w's j-th instruction fills chunks (w*64 + NTHR*j) .. +63
    // (asm, like the weight tiles above: with the builtin pending, hipcc put vmcnt(0) in front of every tile's first LDS read in
    // the ConvT#1 loop, and that counter also holds the previous tap's global stores -- each tile waited for them to complete)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)msm;
#pragma unroll
    for (int j = 0; j < W1CH; ++j) {
      const u32x4* base = w1g + half * 2048 + C::NTHR * j;   // wave-uniform
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds0 + (unsigned)((wave * 64 + C::NTHR * j) * 16));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"((unsigned)(tid * 16)), "s"(dst), "s"(base) : "memory");
    }
  };
  auto r2l_p = [&]() {
#pragma unroll
    for (int j = 0; j < C::PCH; ++j) {
      const int r = xrow + RSTEP * j;
      if (r < C::PATCH) *reinterpret_cast<u32x4*>(Ps + r * M_PITCH + c8 * 8) = pr[j];
    }
  };

  // accumulators start at the bias (ConvT#0 bias as an MFMA C operand)
  f32x16 acc[4];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) acc[mt] = load_bias16(reinterpret_cast<const char*>(p.b0f) + (mt * 2 + h) * 64);

  const int NS = 4 * NT;  // stages: 64-channel chunk c = s / NT (outer), tap t = s % NT (inner)
  if (!MID_ABL(4)) { g2r_p(0); g2l_w(0, 0, 0); }
  // A-fragment reads: row mt*32 + l31, k-chunk 2 kk + h at slot (2 kk + h) ^ ((l31 >> 1) & 7) = (2 kk) ^ wbase
  const int wbase = h ^ ((l31 >> 1) & 7);
  const uint16_t* wsl = Ws + l31 * 64;
  for (int s = 0; s < NS; ++s) {
    const int c = s / NT, t = s - c * NT;
    if (t == 0) {
      if (s > 0) __syncthreads();   // every wave is past the previous chunk's last MFMAs before its patch is overwritten
      r2l_p();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // tile s (asm global_load_lds, issued a stage ago) has landed in this wave's share
    __syncthreads();   // ... and in everybody's; the patch is visible
    if (s == 0) MID_STAMP(2);
    if (s + 1 < NS && !MID_ABL(4)) {
      const int c1 = (s + 1) / NT, t1 = (s + 1) - c1 * NT;
      if (t1 == 0) g2r_p(c1);
      g2l_w(t1, c1, (s + 1) & 1);   // into the buffer stage s - 1 read: all waves finished it before the barrier above
    }
    const uint16_t* bsrc = Ps + trow[t];
    const uint16_t* wsb = wsl + (s & 1) * 8192;
    if (!MID_ABL(1))
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) {
      const uint4 bf = *reinterpret_cast<const uint4*>(bsrc + kk * 16);
      const int wo = ((2 * kk) ^ wbase) * 8;
      uint4 af[4];
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const uint4*>(wsb + mt * 32 * 64 + wo);
#pragma unroll
      for (int mt = 0; mt < 4; ++mt) acc[mt] = mfma32<F16>(af[mt], bf, acc[mt]);
    }
  }
  __syncthreads();   // the ConvT#1 stage re-uses the patch and weight space
  MID_STAMP(3);

  // ---- ConvT#0 epilogue: swish, pack; the packed accumulators are ConvT#1's B operands ----
  // ConvT#1's first operand half is in flight during the swish below
  if (!MID_ABL(2)) g2l_w1(0);
  uint32_t fb[4][8];
#pragma unroll
  for (int mt = 0; mt < 4; ++mt) swish_pack16<F16>(acc[mt], fb[mt]);

  MID_STAMP(4);
  // ---- ConvT#1: 8 tiles of 32 rows (tap = j8 >> 1, channels 32*(j8&1)..+31), K = 128 = 8 k-steps.
  // Its 64 KB of A operands go through the (now free) LDS one 8 KB tile at a time (double-buffered), shared by all waves.
  // Each tap's 32 pixels x 64 channels are transposed through a wave-private LDS tile so that the
  // global stores are whole 128-byte pixel rows (16 B per lane, 8 lanes per pixel) instead of
  // 8-byte pieces 512 B apart -- the scattered form was store-issue bound (~0.06 ms per batch).
  uint4* w1s = reinterpret_cast<uint4*>(msm);          // four 8 KB operand tiles
  char* stage = msm + 32768 + wave * (32 * 144);       // [32 pixels][144 B]
  const int Y = 2 * my + py, X = 2 * mx + px;  // 25x25-level pixel
  const int obase = img >= 0 ? ((img * 50 + 2 * Y) * 50 + 2 * X) * 64 : -1;  // element offset of tap (0,0)
  if (!MID_ABL(2)) {
    // ConvT#1's bias fragments go to LDS (the row tables there are dead): read from memory at the top of every tile they made
    // the tile's first MFMA wait on vmcnt -- an in-order counter that also holds the previous tap's global stores
    if (tid < 64) reinterpret_cast<float*>(msm + C::OFF_META)[tid] = p.b1f[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the first operand half has landed
    __syncthreads();
    // Round 3: software-pipelined by one tile.  The eight MFMAs of tile jj + 1 are issued INSIDE the swish block of tile jj, one after
    // every four transcendentals of this wave's own stream (dev16.h swish_pack_s; tools/microbench9.hip: an MFMA there costs the SIMD
    // about its 8 issue cycles instead of its 32), each with its A fragment read from LDS one hook earlier.  Tiles 0 and 4 (the first
    // of either operand half) are issued in front of the loop / behind the reload.  Same instructions per element: bit-identical.
    f32x2 one2 = {1.0f, 1.0f};
    asm volatile("" : "+v"(one2));
    f32x16 a1[2];
    auto bfrag = [&](const int s) { return make_uint4(fb[s >> 1][4 * (s & 1)], fb[s >> 1][4 * (s & 1) + 1], fb[s >> 1][4 * (s & 1) + 2], fb[s >> 1][4 * (s & 1) + 3]); };
    auto issue_tile = [&](const int jj, f32x16& a) {       // all eight MFMAs back to back (nothing to hide them under)
      a = load_bias16(msm + C::OFF_META + ((jj & 1) * 2 + h) * 64);
      const uint4* wt = w1s + (jj & 3) * 512 + lane;
#pragma unroll
      for (int s = 0; s < 8; ++s) a = mfma32<F16>(wt[s * 64], bfrag(s), a);
    };
    auto tile_out = [&](const int jj, const uint32_t (&o)[8]) {
      const int tap = jj >> 1, jh = jj & 1;
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<uint2*>(stage + l31 * 144 + 64 * jh + 16 * q + 8 * h) = make_uint2(o[2 * q], o[2 * q + 1]);
      if (jh == 1) {
        // coalesced write-out of this tap: lane -> (pixel = lane/8 + 8r, 16-byte chunk = lane%8)
        const int toff = ((tap >> 1) * 50 + (tap & 1)) * 64;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int pix = (lane >> 3) + 8 * r;
          const int ob = __shfl(obase, pix, 64);
          const uint4 v = *reinterpret_cast<const uint4*>(stage + pix * 144 + (lane & 7) * 16);
          if (ob >= 0 && !MID_ABL(8)) *reinterpret_cast<uint4*>(p.out + ob + toff + (lane & 7) * 8) = v;
        }
      }
    };
    issue_tile(0, a1[0]);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
      uint32_t o[8];
      constexpr int dummy = 0; (void)dummy;
      if (jj + 1 < 8 && jj + 1 != 4 && !MID_ABL(16)) {
        f32x16& an = a1[(jj + 1) & 1];
        an = load_bias16(msm + C::OFF_META + (((jj + 1) & 1) * 2 + h) * 64);
        const uint4* wt = w1s + ((jj + 1) & 3) * 512 + lane;
        uint4 wf = wt[0];
        swish_pack_s<F16, 0, 16>(a1[jj & 1], o, one2, [&](auto k) {
          constexpr int S = decltype(k)::value;
          an = mfma32<F16>(wf, bfrag(S), an);
          if constexpr (S < 7) wf = wt[(S + 1) * 64];      // the next fragment, into the register the MFMA has just read
        });
      } else {
        swish_pack16<F16>(a1[jj & 1], o, MID_ABL(16));
        if (jj + 1 < 8 && jj + 1 != 4) issue_tile(jj + 1, a1[(jj + 1) & 1]);   // (diagnostic no-swish build)
      }
      tile_out(jj, o);
      if (jj == 3) {   // second half of the operands: the only point of the stage where the waves meet (its load latency is exposed once)
        __syncthreads();
        g2l_w1(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue_tile(4, a1[0]);
      }
    }
  }
#ifdef SRCFD_DIAG
  if (p.prof && tid == 0) {
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    atomicAdd(p.prof + phase * 6 + 0, 1ull);
    atomicAdd(p.prof + phase * 6 + 1, tstamp[1] - tstamp[0]);
    atomicAdd(p.prof + phase * 6 + 2, tstamp[2] - tstamp[1]);
    atomicAdd(p.prof + phase * 6 + 3, tstamp[3] - tstamp[2]);
    atomicAdd(p.prof + phase * 6 + 4, tstamp[4] - tstamp[3]);
    atomicAdd(p.prof + phase * 6 + 5, t5 - tstamp[4]);
  }
#endif
}

template <bool F16, int NW>
static hipError_t launch_mid16_nw(const MidParams& p, hipStream_t s) {
  using C = MidCfg<NW>;
  void (*fn)(MidParams) = mid16<F16, NW>;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(fn), C::LDS);
  if (e != hipSuccess) return e;
  const int blocks = (p.n * 169 + C::PX - 1) / C::PX;  // the largest phase (13x13 pixels per sample)
  hipLaunchKernelGGL(fn, dim3(blocks, 4), dim3(C::NTHR), C::LDS, s, p);
  return hipGetLastError();
}

hipError_t launch_mid16(bool f16, const MidParams& p, int waves, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  if (waves == 4) return f16 ? launch_mid16_nw<true, 4>(p, s) : launch_mid16_nw<false, 4>(p, s);
  if (waves == 16) return f16 ? launch_mid16_nw<true, 16>(p, s) : launch_mid16_nw<false, 16>(p, s);
  return f16 ? launch_mid16_nw<true, 8>(p, s) : launch_mid16_nw<false, 8>(p, s);
}

}  // namespace srcfd
