// mid16: ConvT#0 (3x3, stride 2, 256->128, 12x12 -> 25x25) chained into ConvT#1 (2x2,
// stride 2, 128->64, -> 50x50) in one kernel (SURVEY.md 8a rows a13 + a14), 16-bit operands.
//
// Why not the generic implicit GEMM: measured on MI355X it spent ~0.45 ms/batch on these two
// layers, bound by re-fetching the same input pixels from beyond L2 once per kernel tap, and by
// writing / re-reading the 25x25x128 activation.  Here
//   * ConvT#0 is split into its four output phases; a workgroup owns 256 (or 512) consecutive output
//     pixels of one phase and stages the contiguous slab of input pixels they touch ("patch") in LDS
//     once per 64-channel chunk; every tap then gathers its B operand from LDS with a per-lane row
//     index, so the input is read from memory once per phase, not once per tap;
//   * a wave owns 32 or 64 pixels (PT = 1 or 2 pixel tiles) x all 128 channels, so after bias + swish
//     the accumulators, packed to 16 bits, ARE the B operands of ConvT#1 (k order permuted on the
//     host to the accumulator's register order): 64 more MFMAs per pixel tile produce the four 2x2
//     taps x 64 channels, and only the 50x50x64 result goes to HBM;
//   * workgroup shapes: 4 waves x 64 pixels, two workgroups per CU (shipped);
//     8 x 64, one per CU, three weight tiles in flight; 8 x 32, two per CU (rounds 1-2).
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
// PT pixel tiles per wave: 1 = two workgroups of 256 pixels per CU (128 registers); 2 = one workgroup of 512 pixels per CU, a wave
// owns 64 pixels (256 registers): every weight fragment read from LDS feeds two MFMAs instead of one, 0.75 KB of LDS reads per MFMA
// instead of 1.25 -- with one tile per wave the main loop sits on the LDS port (16 waves x 20 KB per stage = 320 KB at 128 B/clk =
// 2.5 k cycles + bank conflicts against 2.0 k of MFMA; measured 3.0 k, profiles/r03/d_...).
template <int NW, int PT = 1> struct MidCfg {
  static constexpr int PX = 32 * NW * PT;               // output pixels per workgroup
  // input-pixel slab bound, rows: a 12-wide phase packs PX pixels into PX/12+1 output rows (one input row each),
  // plus the row above the first one and one row of slack for the partial first/last rows
  static constexpr int PATCH = (PX / 12 + 3) * 12 + 16;
  static constexpr int OFF_W = (PATCH + 1) * M_PITCH * 2;       // bytes; row PATCH is all zero
  static constexpr int NBUF = (PT == 2 && NW == 8) ? 3 : 2;                  // weight tiles in flight + the one being read: with one workgroup per CU nobody else covers a tile's load latency, so it is fetched two stages ahead
  static constexpr int MAIN_END = OFF_W + NBUF * 16384;         // weight tiles of [128 rows][64 k] at 128 B per row (XOR-swizzled)
  static constexpr int W1_LOADS = (PT == 2 && NW == 8) ? 1 : 2;              // ConvT#1's 64 KB of operands: in two halves, or (the larger workgroup has the LDS) at once
  static constexpr int W1_BYTES = 65536 / W1_LOADS;
  static constexpr int T1_END = W1_BYTES + NW * 32 * 144;       // ConvT#1 stage: its operands (8 KB tiles) + per-wave store tiles
  static constexpr int OFF_META = MAIN_END > T1_END ? MAIN_END : T1_END;
  static constexpr int LDS = OFF_META + 3 * PX * 4;
  static_assert(PT == 1 || PT == 2, "one or two 32-pixel tiles per wave");
  static constexpr int NTHR = 64 * NW;
  static constexpr int WCH = 1024 / NTHR;               // weight-tile 16-byte chunks per thread
  static constexpr int PCH = (PATCH * 8 + NTHR - 1) / NTHR;  // patch chunks per thread
};

template <bool F16, int NW, int PT>
__global__ void __launch_bounds__(64 * NW, PT == 2 ? 2 : (NW == 8 ? 4 : 1)) mid16(MidParams p) {
  using C = MidCfg<NW, PT>;
  static_assert(C::OFF_META >= C::T1_END, "ConvT#1 operand tiles + per-wave store tiles are staged over the patch + weight tiles");
  extern __shared__ __attribute__((aligned(16))) char msm[];
  uint16_t* Ps = reinterpret_cast<uint16_t*>(msm);
  uint16_t* Ws = reinterpret_cast<uint16_t*>(msm + C::OFF_W);
  int* row_img = reinterpret_cast<int*>(msm + C::OFF_META);
  int* row_my = row_img + C::PX;
  int* row_mx = row_my + C::PX;

  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), h = lane >> 5, l31 = lane & 31;
  // linear workgroup id -> (output phase, block of the phase).  order 0: phase by phase, longest first (phase 0: four taps ... phase 3: one).
  // order 1: phases 0 and 3 alternate, then 1 and 2: the workgroups that start together then reach their store phase at different times.
  int phase = 0, blk = 0;
  {
    int id = blockIdx.x;
    const int nb0 = p.nblk[0], nb1 = p.nblk[1], nb2 = p.nblk[2], nb3 = p.nblk[3];
    if (p.order == 0) {
      if (id < nb0) { phase = 0; blk = id; }
      else if (id < nb0 + nb1) { phase = 1; blk = id - nb0; }
      else if (id < nb0 + nb1 + nb2) { phase = 2; blk = id - nb0 - nb1; }
      else { phase = 3; blk = id - nb0 - nb1 - nb2; }
    } else {
      const int n03 = min(nb0, nb3), n12 = min(nb1, nb2);
      if (id < 2 * n03) { phase = (id & 1) ? 3 : 0; blk = id >> 1; }
      else if ((id -= 2 * n03) < nb0 - n03) { phase = 0; blk = n03 + id; }
      else if ((id -= nb0 - n03) < nb3 - n03) { phase = 3; blk = n03 + id; }
      else if ((id -= nb3 - n03) < 2 * n12) { phase = (id & 1) ? 2 : 1; blk = id >> 1; }
      else if ((id -= 2 * n12) < nb1 - n12) { phase = 1; blk = n12 + id; }
      else { phase = 2; blk = n12 + id - (nb1 - n12); }
    }
  }
#ifdef SRCFD_DIAG
  unsigned long long tstamp[6], t_sync = 0, t_issue = 0, t_mma = 0, t_a = 0, t_b = 0;
  tstamp[0] = __builtin_amdgcn_s_memtime();
#define MID_T(v) do { if (p.prof) { __builtin_amdgcn_sched_barrier(0); v = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#define MID_STAMP(i) do { if (p.prof) { __builtin_amdgcn_sched_barrier(0); tstamp[i] = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define MID_STAMP(i) do { } while (0)
#define MID_T(v) do { } while (0)
#endif
  const int py = phase >> 1, px = phase & 1;
  const int TY = py ? 1 : 2, TX = px ? 1 : 2, NT = TY * TX;
  const int MH = py ? 12 : 13, MW = px ? 12 : 13, per = MH * MW;
  const int M = p.n * per, m0 = blk * C::PX;
  if (m0 >= M) return;
  const uint16_t* Wt = p.w0t[phase];   // stage tiles, 16 KB each, already in LDS order

  for (int i = tid; i < C::PX; i += C::NTHR) {
    int m = m0 + i, img = -1, my = 0, mx = 0;
    if (m < M) { img = m / per; int r = m - img * per; my = r / MW; mx = r - my * MW; }
    row_img[i] = img; row_my[i] = my; row_mx[i] = mx;
  }
  if (tid < M_PITCH / 2) reinterpret_cast<uint32_t*>(Ps + C::PATCH * M_PITCH)[tid] = 0;
  __syncthreads();

  MID_STAMP(1);
  // contiguous slab of input pixels (tensor index img*144 + iy*12 + ix) this workgroup touches
  const int mlast = min(m0 + C::PX - 1, M - 1) - m0;
  const int lo = row_img[0] * 144 + max(row_my[0] - 1, 0) * 12;
  const int hi = row_img[mlast] * 144 + min(row_my[mlast], 11) * 12 + 11;
  const int NP = min(hi - lo + 1, C::PATCH);  // <= PATCH by construction; the clamp only guards LDS

  // this lane's pixel(s) and the patch row each tap reads
  int img[PT], my[PT], mx[PT], trow[PT][4];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int prow = (wave * PT + pt) * 32 + l31;
    img[pt] = row_img[prow]; my[pt] = row_my[prow]; mx[pt] = row_mx[prow];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int ty = t / TX, tx = t - ty * TX;  // only t < NT is used
      const int iy = my[pt] - ty, ix = mx[pt] - tx;
      int r = img[pt] * 144 + iy * 12 + ix - lo;
      const bool ok = img[pt] >= 0 && t < NT && (unsigned)iy < 12u && (unsigned)ix < 12u && (unsigned)r < (unsigned)C::PATCH;
      trow[pt][t] = (ok ? r : C::PATCH) * M_PITCH + h * 8;
    }
  }

  // staging roles: 16-byte column c8 of rows xrow + (NTHR/8)*j
  constexpr int RSTEP = C::NTHR / 8;
  const int xrow = tid >> 3, c8 = tid & 7;
  // native vector type: arrays of HIP's uint4 struct were left in scratch by the compiler (store/reload around every
  // prefetch), which serialised the global loads
  typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
  u32x4 pr[C::PCH];
  // Weight tiles go global -> LDS directly (global_load_lds: no registers, no ds_write pass), two or three buffers.  The LDS image
  // is lane-linear (chunk index = row * 8 + slot = tid + NTHR * j) and the host stores every stage's tile in exactly that order
  // (fused_bf16.hip: 16 KB of consecutive memory per stage), bank swizzle included: slot `sl` of row r holds k-chunk
  // sl ^ ((r >> 1) & 7), and the fragment reads below apply the same XOR.
  // The builtin form of the instruction is known to hipcc as a pending LDS write: it then puts `s_waitcnt vmcnt(0)` in front of
  // the next ds_read -- here the fragment reads of the tile being COMPUTED -- which drains the prefetch before the MFMAs start
  // (seen in the ISA; the kernel gained nothing from the prefetch).  As an asm statement the load is outside hipcc's
  // bookkeeping: the stage loop waits for it itself (vmcnt(0) in front of the barrier that publishes the tile).
  const unsigned lds_w0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)(msm + C::OFF_W);
  // Addresses of the staging loads are a wave-uniform base (SGPR pair) + a 32-bit per-lane offset: as 64-bit per-lane pointers (round 2)
  // the five patch pointers alone took ten registers, hipcc spilled them, and every reload in the loop came with an
  // s_waitcnt vmcnt(0) that drained the weight tile in flight (seen in the ISA, round 3)
  auto g2l_w = [&](int st, int buf) {   // stage st = chunk * taps + tap
#pragma unroll
    for (int j = 0; j < C::WCH; ++j) {
      const uint16_t* base = Wt + (size_t)st * 8192 + C::NTHR * j * 8;     // wave-uniform
      const unsigned dst = __builtin_amdgcn_readfirstlane(lds_w0 + (unsigned)(buf * 16384 + (wave * 64 + C::NTHR * j) * 16));
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %3\n\ts_mov_b32 m0, %0"
                   : "=&s"(keep) : "v"((unsigned)(tid * 16)), "s"(dst), "s"(base) : "memory");
    }
  };
  auto g2r_p = [&](int c) {
    const char* base = reinterpret_cast<const char*>(p.in + (size_t)lo * 256 + c * 64);   // wave-uniform
#pragma unroll
    for (int j = 0; j < C::PCH; ++j) {
      // rows >= NP of the patch are never gathered (trow points past-the-slab pixels at the zero row), so they may hold anything: the
      // load is unconditional from a clamped row -- a zero fill of the staging registers made hipcc wait for every load in flight
      // (vmcnt(0), the weight tiles included) in front of the v_mov
      const int r = min(xrow + RSTEP * j, NP - 1);
      pr[j] = *reinterpret_cast<const u32x4*>(base + (unsigned)(r * 512 + c8 * 16));
    }
  };
  // ConvT#1's 64 KB of A operands (8 tiles of 8 KB = 512 chunks of 16 B, one per (tap, 32-channel half)) go through the LDS
  // the main loop has left, in two halves of four tiles, by global_load_lds (no registers: the accumulators, the packed
  // B operands and the swish temporaries fill the file in that stage); the waves work through a half without
  // synchronising with each other
  constexpr int W1CHUNKS = C::W1_BYTES / 16;
  constexpr int W1CH = W1CHUNKS / C::NTHR;
  static_assert(W1CH * C::NTHR == W1CHUNKS, "a load of ConvT#1's operands divides evenly over the workgroup");
  const u32x4* w1g = reinterpret_cast<const u32x4*>(p.w1f);
  auto g2l_w1 = [&](int half) {   // lane-linear LDS image: wave w's j-th instruction fills chunks (w*64 + NTHR*j) .. +63
    // (asm, like the weight tiles above: with the builtin pending, hipcc put vmcnt(0) in front of every tile's first LDS read in
    // the ConvT#1 loop, and that counter also holds the previous tap's global stores -- each tile waited for them to complete)
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)msm;
#pragma unroll
    for (int j = 0; j < W1CH; ++j) {
      const u32x4* base = w1g + half * W1CHUNKS + C::NTHR * j;   // wave-uniform
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
  f32x16 acc[PT][4];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[pt][mt] = load_bias16(reinterpret_cast<const char*>(p.b0f) + (mt * 2 + h) * 64);

  const int NS = 4 * NT;  // stages: 64-channel chunk c = s / NT (outer), tap t = s % NT (inner)
  constexpr int LA = C::NBUF - 1;   // stages a weight tile is fetched ahead of its use
  if (!MID_ABL(4)) {
    g2r_p(0);
    g2l_w(0, 0);
    if (LA == 2 && NS > 1) g2l_w(1, 1);
  }
  int bcur = 0;                     // ring slot of stage s
  // A-fragment reads: row mt*32 + l31, k-chunk 2 kk + h at slot (2 kk + h) ^ ((l31 >> 1) & 7) = (2 kk) ^ wbase
  const int wbase = h ^ ((l31 >> 1) & 7);
  const uint16_t* wsl = Ws + l31 * 64;
  for (int s = 0; s < NS; ++s) {
    const int c = s / NT, t = s - c * NT;
#ifdef SRCFD_DIAG
    MID_T(t_a);
#endif
    if (t == 0) {
      if (s > 0) __syncthreads();   // every wave is past the previous chunk's last MFMAs before its patch is overwritten
      r2l_p();
    }
    // tile s (asm global_load_lds, issued LA stages ago) has landed in this wave's share: with LA == 2 tile s + 1 (the WCH youngest
    // loads) may still be in flight, except behind a patch load (t == 0: hipcc's own wait for the patch registers in r2l_p covered everything)
    if (LA == 2 && s + 1 < NS && t != 0) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(C::WCH) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();   // ... and in everybody's; the patch is visible
    if (s == 0) MID_STAMP(2);
#ifdef SRCFD_DIAG
    MID_T(t_b); if (s > 0) t_sync += t_b - t_a;
#endif
    int bnext = bcur + LA; bnext = bnext >= C::NBUF ? bnext - C::NBUF : bnext;
    auto issue_loads = [&]() {
      if (MID_ABL(4)) return;
      const int c1 = (s + 1) / NT, t1 = (s + 1) - c1 * NT;
      if (s + 1 < NS && t1 == 0) g2r_p(c1);   // in front of the tile: hipcc's wait for the patch registers (next stage) is vmcnt(0) as far as it knows
      if (s + LA < NS) g2l_w(s + LA, bnext);   // into the buffer stage s - 1 read: all waves finished it before the barrier above
    };
    // One workgroup per CU: the loads of later stages are not what a wave leaving the barrier should spend its first cycles on -- its
    // own fragment reads and MFMAs are; the loads go out a few units into the stage (they have two stages to land)
    if (PT == 1) issue_loads();
#ifdef SRCFD_DIAG
    MID_T(t_a); t_issue += t_a - t_b;
#endif
    const uint16_t* bsrc[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) bsrc[pt] = Ps + trow[pt][t];
    const uint16_t* wsb = wsl + bcur * 8192;
    bcur = bcur + 1 == C::NBUF ? 0 : bcur + 1;
    if (!MID_ABL(1)) {
      if constexpr (PT == 2) {
        // Two waves per SIMD: nobody else fills the LDS round trip in front of a k-step's MFMAs, and left to itself hipcc re-reads every
        // weight fragment into the same three register quads (two MFMAs, then a full LDS latency: the matrix pipe idled two thirds of a
        // stage).  So the stage is software-pipelined by hand in units of one weight fragment (two MFMAs, 64 cycles): the fragment of
        // unit u + 3 is read into a ring of four quads while unit u runs, the two pixel fragments of k-step kk + 1 at the start of kk.
        // (Double-buffering all six fragments of a k-step -- 48 registers -- spilled, and a reload in the ConvT#1 stage waits on vmcnt
        // behind that stage's global stores.)
        uint4 bf[2][PT], af[4];
        auto rd_a = [&](const int u) {   // unit u = (k-step u >> 2, channel tile u & 3)
          af[u & 3] = *reinterpret_cast<const uint4*>(wsb + (u & 3) * 32 * 64 + ((2 * (u >> 2)) ^ wbase) * 8);
        };
        auto rd_b = [&](const int kk) {
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) bf[kk & 1][pt] = *reinterpret_cast<const uint4*>(bsrc[pt] + kk * 16);
        };
        rd_b(0); rd_a(0); rd_a(1); rd_a(2);
#pragma unroll
        for (int u = 0; u < 16; ++u) {
          if (u + 3 < 16) rd_a(u + 3);
          if ((u & 3) == 0 && u < 12) rd_b((u >> 2) + 1);
          if (u == 2) issue_loads();
          pin();
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) acc[pt][u & 3] = mfma32<F16>(af[u & 3], bf[(u >> 2) & 1][pt], acc[pt][u & 3]);
          pin();
        }
      } else {
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          uint4 bf[PT];
#pragma unroll
          for (int pt = 0; pt < PT; ++pt) bf[pt] = *reinterpret_cast<const uint4*>(bsrc[pt] + kk * 16);
          const int wo = ((2 * kk) ^ wbase) * 8;
          uint4 af[4];
#pragma unroll
          for (int mt = 0; mt < 4; ++mt) af[mt] = *reinterpret_cast<const uint4*>(wsb + mt * 32 * 64 + wo);
#pragma unroll
          for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int pt = 0; pt < PT; ++pt) acc[pt][mt] = mfma32<F16>(af[mt], bf[pt], acc[pt][mt]);
        }
      }
    }
#ifdef SRCFD_DIAG
    MID_T(t_b); t_mma += t_b - t_a;
#endif
  }
  __syncthreads();   // the ConvT#1 stage re-uses the patch and weight space
  MID_STAMP(3);

  // ---- ConvT#0 epilogue: swish, pack; the packed accumulators are ConvT#1's B operands ----
  // ConvT#1's first operand half is in flight during the swish below
  if (!MID_ABL(2)) g2l_w1(0);
  uint32_t fb[PT][4][8];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt)
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) swish_pack16<F16>(acc[pt][mt], fb[pt][mt]);

  MID_STAMP(4);
  // ---- ConvT#1: 8 tiles of 32 rows (tap = j8 >> 1, channels 32*(j8&1)..+31), K = 128 = 8 k-steps.
  // Its 64 KB of A operands go through the (now free) LDS one 8 KB tile at a time (double-buffered), shared by all waves.
  // Each tap's 32 pixels x 64 channels are transposed through a wave-private LDS tile so that the
  // global stores are whole 128-byte pixel rows (16 B per lane, 8 lanes per pixel) instead of
  // 8-byte pieces 512 B apart -- the scattered form was store-issue bound (~0.06 ms per batch).
  uint4* w1s = reinterpret_cast<uint4*>(msm);          // 8 KB operand tiles (four or all eight resident)
  char* stage = msm + C::W1_BYTES + wave * (32 * 144);  // [32 pixels][144 B]
  int obase[PT];                                        // element offset of tap (0,0) of the lane's pixel(s)
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int Y = 2 * my[pt] + py, X = 2 * mx[pt] + px;  // 25x25-level pixel
    obase[pt] = img[pt] >= 0 ? ((img[pt] * 50 + 2 * Y) * 50 + 2 * X) * 64 : -1;
  }
  if (!MID_ABL(2)) {
    // ConvT#1's bias fragments go to LDS (the row tables there are dead): read from memory at the top of every tile they made
    // the tile's first MFMA wait on vmcnt -- an in-order counter that also holds the previous tap's global stores
    if (tid < 64) reinterpret_cast<float*>(msm + C::OFF_META)[tid] = p.b1f[tid];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's share of the first operand load has landed
    __syncthreads();
    // Round 3: software-pipelined by one tile.  The eight MFMAs of the next tile are issued INSIDE the swish block of the current one, one
    // after every four transcendentals of this wave's own stream (dev16.h swish_pack_s; tools/microbench9.hip: an MFMA there costs the SIMD
    // about its 8 issue cycles instead of its 32), each with its A fragment read from LDS one hook earlier.  The first tile of an operand
    // load is issued in front of the loop / behind the reload.  Same instructions per element: bit-identical.
    // Item i of the stage = (operand load i / (TPL * PT), pixel tile pt, operand tile jj): with two pixel tiles per wave both run
    // through the resident operand tiles before the next load.
    f32x2 one2 = {1.0f, 1.0f};
    asm volatile("" : "+v"(one2));
    f32x16 a1[2];
    constexpr int TPL = 8 / C::W1_LOADS;     // operand tiles per load
    constexpr int NI = 8 * PT;               // items
    auto item_pt = [](const int i) { return (i % (TPL * PT)) / TPL; };
    auto item_jj = [](const int i) { return (i / (TPL * PT)) * TPL + i % TPL; };
    auto bfrag = [&](const int pt, const int s) { return make_uint4(fb[pt][s >> 1][4 * (s & 1)], fb[pt][s >> 1][4 * (s & 1) + 1], fb[pt][s >> 1][4 * (s & 1) + 2], fb[pt][s >> 1][4 * (s & 1) + 3]); };
    auto issue_tile = [&](const int pt, const int jj, f32x16& a) {       // all eight MFMAs back to back (nothing to hide them under)
      a = load_bias16(msm + C::OFF_META + ((jj & 1) * 2 + h) * 64);
      const uint4* wt = w1s + (jj % TPL) * 512 + lane;
#pragma unroll
      for (int s = 0; s < 8; ++s) a = mfma32<F16>(wt[s * 64], bfrag(pt, s), a);
    };
    auto tile_out = [&](const int pt, const int jj, const uint32_t (&o)[8]) {
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
          const int ob = __shfl(obase[pt], pix, 64);
          const uint4 v = *reinterpret_cast<const uint4*>(stage + pix * 144 + (lane & 7) * 16);
          if (ob >= 0 && !MID_ABL(8)) *reinterpret_cast<uint4*>(p.out + ob + toff + (lane & 7) * 8) = v;
        }
      }
    };
    issue_tile(0, 0, a1[0]);
#pragma unroll
    for (int i = 0; i < NI; ++i) {
      uint32_t o[8];
      const int pt = item_pt(i), jj = item_jj(i);
      const bool reload_next = C::W1_LOADS == 2 && i + 1 == NI / 2;   // the next item is the first of the second operand half
      if (i + 1 < NI && !reload_next && !MID_ABL(16)) {
        const int ptn = item_pt(i + 1), jjn = item_jj(i + 1);
        f32x16& an = a1[(i + 1) & 1];
        an = load_bias16(msm + C::OFF_META + ((jjn & 1) * 2 + h) * 64);
        const uint4* wt = w1s + (jjn % TPL) * 512 + lane;
        uint4 wf = wt[0];
        swish_pack_s<F16, 0, 16>(a1[i & 1], o, one2, [&](auto k) {
          constexpr int S = decltype(k)::value;
          an = mfma32<F16>(wf, bfrag(ptn, S), an);
          if constexpr (S < 7) wf = wt[(S + 1) * 64];      // the next fragment, into the register the MFMA has just read
        });
      } else {
        swish_pack16<F16>(a1[i & 1], o, MID_ABL(16));
        if (i + 1 < NI && !reload_next) issue_tile(item_pt(i + 1), item_jj(i + 1), a1[(i + 1) & 1]);   // (diagnostic no-swish build)
      }
      tile_out(pt, jj, o);
      if (reload_next) {   // second half of the operands: the only point of the stage where the waves meet (its load latency is exposed once)
        __syncthreads();
        g2l_w1(1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        issue_tile(item_pt(i + 1), item_jj(i + 1), a1[(i + 1) & 1]);
      }
    }
  }
#ifdef SRCFD_DIAG
  if (p.prof && blk == 3 && phase == 0 && lane == 0 && wave < 8) {   // one workgroup of the longest phase, every wave
    p.prof[36 + wave * 3 + 0] = t_sync; p.prof[36 + wave * 3 + 1] = t_issue; p.prof[36 + wave * 3 + 2] = t_mma;
  }
  if (p.prof && tid == 0) {
    const unsigned long long t5 = __builtin_amdgcn_s_memtime();
    atomicAdd(p.prof + phase * 6 + 0, 1ull);
    atomicAdd(p.prof + phase * 6 + 1, tstamp[1] - tstamp[0]);
    atomicAdd(p.prof + phase * 6 + 2, tstamp[2] - tstamp[1]);
    atomicAdd(p.prof + phase * 6 + 3, tstamp[3] - tstamp[2]);
    atomicAdd(p.prof + phase * 6 + 4, tstamp[4] - tstamp[3]);
    atomicAdd(p.prof + phase * 6 + 5, t5 - tstamp[4]);
    atomicAdd(p.prof + 24 + phase * 3 + 0, t_sync);
    atomicAdd(p.prof + 24 + phase * 3 + 1, t_issue);
    atomicAdd(p.prof + 24 + phase * 3 + 2, t_mma);
  }
#endif
}

template <bool F16, int NW, int PT = 1>
static hipError_t launch_mid16_nw(const MidParams& p, hipStream_t s) {
  using C = MidCfg<NW, PT>;
  void (*fn)(MidParams) = mid16<F16, NW, PT>;
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(fn), C::LDS);
  if (e != hipSuccess) return e;
  MidParams q = p;
  const int per[4] = {169, 156, 156, 144};   // pixels per sample and output phase (13x13, 13x12, 12x13, 12x12)
  int total = 0;
  for (int ph = 0; ph < 4; ++ph) { q.nblk[ph] = (p.n * per[ph] + C::PX - 1) / C::PX; total += q.nblk[ph]; }
  hipLaunchKernelGGL(fn, dim3(total), dim3(C::NTHR), C::LDS, s, q);
  return hipGetLastError();
}

hipError_t launch_mid16(bool f16, const MidParams& p, int waves, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  if (waves == 4) return f16 ? launch_mid16_nw<true, 4>(p, s) : launch_mid16_nw<false, 4>(p, s);
  if (waves == 16) return f16 ? launch_mid16_nw<true, 16>(p, s) : launch_mid16_nw<false, 16>(p, s);
  if (waves == 42) return f16 ? launch_mid16_nw<true, 4, 2>(p, s) : launch_mid16_nw<false, 4, 2>(p, s);   // 4 waves x 2 pixel tiles: 256 pixels per workgroup, two per CU, one wave of either on a SIMD
  if (waves == 82) return f16 ? launch_mid16_nw<true, 8, 2>(p, s) : launch_mid16_nw<false, 8, 2>(p, s);   // 8 waves x 2 pixel tiles: 512 pixels per workgroup, one per CU
  return f16 ? launch_mid16_nw<true, 8>(p, s) : launch_mid16_nw<false, 8>(p, s);
}

}  // namespace srcfd
