// tail16s: the SECOND implementation of the 16-bit tail (ConvT#2 -> ConvT#3 -> ConvT#4 -> 3x3 output conv + de-standardise + NaN
// guard in one streaming launch; sr-ae-conv.ipynb:c283-286, PyCFD_ML_accelerated.py:671-673,869-876).
// NOT the shipped kernel: the default is `tail16` in kernels_bf16.hip; this one runs only under SRCFD_TAIL=s (the A/B arm of the
// parity tests, which compare the two bit for bit) and measured 15 % slower (experiments: 4.1c).
//
// Same data flow and LDS layout as tail16: one workgroup per CU walks its samples as one tall image in strips of one 50-level row;
// round r runs A (ConvT#2, strip r, global -> 100-level LDS tile), BC (ConvT#3 -> ConvT#4 in registers, strip r-1, -> 18-row
// ring of the 400x400x8 level) and D (banded-MFMA output conv of strip r-2 from the ring) between two barriers.
//
// What differs is HOW a round is issued (round-3 experiment; tools/microbench9.hip, profiles/r03):
//  * 8 waves of 256 registers instead of 16 of 128.  The kernel is bound by the vector unit's swish stream (two transcendentals
//    per activation); at two waves per SIMD that stream runs within 2 % of its four-wave rate, and a wave now has the registers to
//    hold the NEXT stage's operands and accumulators beside the accumulator it is activating.
//  * Every MFMA is issued INSIDE a swish block of the same wave (swish_pack_s: a hook after every four transcendentals): a
//    v_mfma_f32_32x32x16 costs the SIMD 3.6 ns there against 10.7 ns in front of its own swish block (round 2), the 16x16x32
//    MFMAs of the output conv and their LDS reads ride the same way.  Order is pinned with sched_barrier; the MFMAs stay
//    builtins so hipcc pads their hazards.
//  * One static, SIMD-balanced schedule: waves 0-3 ("X") take pixel tile w of ConvT#3/#4 for both tap rows (96 wave-registers of
//    swish) and three D items; waves 4-7 ("Y") take one full BC item, one half item (the seventh, 8-pixel tile split by tap),
//    both column tiles of one ConvT#2 tap (weights resident) and one D item (104 wave-registers).  Waves w and w + 4 share a SIMD.
//    A BC item is split by ConvT#4 tap without redundant vector work: only the half of ConvT#3's accumulator that feeds the tap
//    is activated.
//  * D's epilogue (fma, guard, store) has no LDS dependency: it is deferred across the barrier and runs at the top of the next
//    round, under the LDS reads and first MFMAs that every wave starts a round with.
// Rounds in which a wave's stages are not all regular (pipeline fill / drain, sample and segment seams) take a plain sequential
// body; both bodies do the same arithmetic per output.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "dev16.h"
#include "kernels16.h"
#include "tail16_layout.h"

namespace srcfd {


// Diagnostic work-skipping switches (TailParams::ablate, SRCFD_TAIL_ABLATE) exist only in a -DSRCFD_DIAG build (make DIAG=1): the
// shipped kernel carries no path that turns work off.  tail16s: 1 Y: no input prefetch, 2 Y: no deferred store, 4 Y: no D item,
// 8 X: no deferred stores, 16 X: no D items, 32 X: no deferred epilogue at all, 64 no swish (conversions only)
#ifdef SRCFD_DIAG
#define TS_ABL(bit) (p.ablate & (bit))
#else
#define TS_ABL(bit) 0
#endif

template <bool F16, int OUT, bool SEG, bool PROF = false>  // OUT: 0 f32, 1 bf16, 2 f16; SEG: samples cut into segments (small batches); PROF: per-wave round timers (DIAG builds)
__global__ void __launch_bounds__(512) tail16s(TailParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, l31 = lane & 31;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  char* ring = smem + T_OFF_RING;
  char* l100 = smem + T_OFF_L100;
  const char* cst = smem + T_OFF_CONST;

  for (int i = tid; i < TAIL_CONST_BYTES / 16; i += 512)
    reinterpret_cast<uint4*>(smem + T_OFF_CONST)[i] = reinterpret_cast<const uint4*>(p.consts)[i];
  if (tid < 4) reinterpret_cast<int*>(smem + T_OFF_ZERO)[tid] = 0;
  for (int i = tid; i < T_OFF_L100 / 16; i += 512) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0, 0, 0, 0);   // ring incl. its zero granules
  __syncthreads();

  const uint4* wc_f = reinterpret_cast<const uint4*>(cst + TC_OFF_WC);
  const uint4* w3_f = reinterpret_cast<const uint4*>(cst + TC_OFF_W3);
  const uint4* w2_f = reinterpret_cast<const uint4*>(p.w2frags);
  const float conv_bias = *reinterpret_cast<const float*>(cst + TC_OFF_BC);
  const uint4 w4 = reinterpret_cast<const uint4*>(cst + TC_OFF_W4)[lane];   // ConvT#4 A operand, resident
  constexpr int OUTSZ = OUT == 0 ? 4 : 2;
  f32x2 one2 = {1.0f, 1.0f};
  asm volatile("" : "+v"(one2));   // opaque: stays in its register pair

  // ---- this wave's items ----
  const bool clsY = wave >= 4;
  const int wq = wave & 3;
  // BC pixel tiles: X: tile wq, both ConvT#3 tap rows m3; Y: full item (tile 4 + (wq >> 1), m3 = wq & 1) and the half item
  // (tile 6, m3 = wq >> 1, ConvT#4 tap tt = wq & 1)
  const int pt_f = clsY ? 4 + (wq >> 1) : wq;
  const int m3_f = wq & 1;                 // Y's full item
  const int m3_h = wq >> 1, tt_h = wq & 1; // Y's half item
  struct PixTile { int l0; int woff; bool valid, a; };
  auto pix_tile = [&](const int pt) {
    PixTile t;
    int idx = 32 * pt + l31;
    t.valid = idx < 200;
    idx = t.valid ? idx : 199;
    t.a = idx >= 100;
    const int x = idx - (t.a ? 100 : 0);
    t.l0 = l100_off(t.a ? 1 : 0, x, h);                                                   // chunk h; chunk 2 + h is 2 * 212 granules further
    t.woff = ((((x & 1) * (4 * T_PLANE) + (x >> 1) + 1) << 4) + 8 * h) + (t.a ? 4 * T_ROWP : 0);   // ring byte offset: x400 = 4 x + c: plane c + 4 (x & 1), granule (x >> 1) + 1; rows + 4 a
    return t;
  };
  const PixTile tf = pix_tile(pt_f), th = pix_tile(6);
  // A (Y only): ConvT#2 tap a_mt = wq, both 32-pixel column tiles of the 50-level row
  const int a_mt = wq;
  unsigned a_lane[2];
  int a_dst[2];
  bool a_valid[2];
#pragma unroll
  for (int ct = 0; ct < 2; ++ct) {
    const int px = 32 * ct + l31;
    a_valid[ct] = px < 50;
    a_lane[ct] = (unsigned)((a_valid[ct] ? px : 49) * 64 + 8 * h);
    a_dst[ct] = l100_off(a_mt >> 1, 2 * (a_valid[ct] ? px : 0) + (a_mt & 1), 0) + 8 * h;
  }
  uint4 wa[4];
  // ConvT#2's input of the NEXT strip, read from global memory a round ahead.  The loads are inline asm into native vector registers:
  // as plain loads hipcc parks them in temporaries and copies them into the loop-carried registers right behind the loads --
  // with an s_waitcnt vmcnt(0) in front of the copies, i.e. the wave sat out the HBM latency it was meant to hide (4.7 k of its
  // 10.4 k cycles per round, SRCFD_TAIL_PROF).  a_wait() -- s_waitcnt vmcnt(0) naming the registers -- stands in front of their use.
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  u32x4 xb[2][4];
#pragma unroll
  for (int kk = 0; kk < 4; ++kk) {
    wa[kk] = clsY ? w2_f[(a_mt * 4 + kk) * 64 + lane] : make_uint4(0, 0, 0, 0);
    xb[0][kk] = u32x4{0, 0, 0, 0}; xb[1][kk] = u32x4{0, 0, 0, 0};
  }
  // SYNC = false (fast rounds): the loads only; the data is waited for by a_wait() at the top of the NEXT round.  SYNC = true (the
  // first strip, and the rounds that take the plain body): loads and s_waitcnt in ONE statement, so that whatever register copies
  // hipcc places behind the statement read landed data (it does place some at the join of the two bodies).
  auto a_load_async = [&](const unsigned row_elems) {   // row_elems: element offset of the 50-level row, (sample * 50 + strip) * 3200  (< 2^31: at most 1024 samples per launch)
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const uint16_t* src = p.in + (size_t)(row_elems + a_lane[ct]);
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                   "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96"
                   : "=&v"(xb[ct][0]), "=&v"(xb[ct][1]), "=&v"(xb[ct][2]), "=&v"(xb[ct][3]) : "v"(src) : "memory");
    }
  };
  auto a_load_sync = [&](const unsigned row_elems) {
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) {
      const uint16_t* src = p.in + (size_t)(row_elems + a_lane[ct]);
      asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:32\n\t"
                   "global_load_dwordx4 %2, %4, off offset:64\n\tglobal_load_dwordx4 %3, %4, off offset:96\n\ts_waitcnt vmcnt(0)"
                   : "=&v"(xb[ct][0]), "=&v"(xb[ct][1]), "=&v"(xb[ct][2]), "=&v"(xb[ct][3]) : "v"(src) : "memory");
    }
  };
  auto a_wait = [&]() {      // everything this wave has in the vector-memory queue: the loads of the previous round (and its stores)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    pin();
  };
  // D: Y wave: item wq (row pair 0); X wave: items 4 + 3 wq .. + 2.  item = 4 rp + j4.
  const int d_first = clsY ? wq : 4 + 3 * wq, d_cnt = clsY ? 1 : 3;
  const int d_tsel = 4 * (lane & 3) + ((lane & 15) >> 2);
  const int d_olane = ((lane >> 5) * 400 + 8 * d_tsel + 4 * ((lane >> 4) & 1)) * OUTSZ;
  const bool d_last_on = d_tsel < 2;
  // lane constants of the D fast path: ring byte offset of (window row dy, column family 0) of item slot i relative to the slot of the
  // item's first window row, and the distance to column family 1 (both column families of a row are one register + an immediate)
  int d_c0[3];
#pragma unroll
  for (int i = 0; i < 3; ++i) {
    int tx = 16 * ((d_first + i) & 3) + d_tsel;
    tx = tx < 50 ? tx : 49;
    d_c0[i] = ((lane >> 4) & 1) * T_ROWP + (tx + 1) * 16 + ((lane >> 5) ? 0 : (7 * T_PLANE - 1) * 16);
  }
  const int d_delta = ((lane >> 5) ? 2 : 1) * T_PLANE * 16 - ((lane >> 5) ? 0 : (7 * T_PLANE - 1) * 16);

  // ---- the workgroup's tall image (as in round 2) ----
  const int S = (SEG && p.seg > 1) ? p.seg : 1, L = 50 / S, SL = S > 1 ? L + 1 : 50;
  const int NV = p.n * S;
  const int K = ((int)blockIdx.x < NV) ? (NV - 1 - (int)blockIdx.x) / (int)gridDim.x + 1 : 0;
  const int G = SL * K;
  // Strip g of the tall image -> (sample, strip index in the sample, ...).  Round 2 recomputed this with divisions for every stage of
  // every round (224 scalar instructions per wave and round: with two waves per SIMD nothing hides them); here the state of strip
  // r + 1 is ADVANCED once per round and the stages take theirs from a four-deep shift register.
  struct Strip { int smp, s, sl, seg; bool valid; };
  auto strip_at = [&](const int g) {
    Strip t;
    t.valid = g >= 0 && g < G;
    const int gg = t.valid ? g : 0;
    const int kv = gg / SL;
    t.sl = gg - SL * kv;
    const int vs = (int)blockIdx.x + kv * (int)gridDim.x;
    t.smp = vs / S;
    t.seg = vs - t.smp * S;
    t.s = S > 1 ? t.seg * L + t.sl - 1 : t.sl;
    return t;
  };
  Strip stN = strip_at(1), stA = strip_at(0), stB = strip_at(-1), stD = strip_at(-1), stP = strip_at(-1);   // strips r+1, r, r-1, r-2, r-3
  int kvN = 1 / SL;                                           // virtual-sample index of stN
  int slotA = 0, slotB = 0, slotD = 0;                        // (8 g) % 18 of the strips of A, BC, D
  if (clsY && K > 0 && stA.valid && stA.s >= 0) a_load_sync((unsigned)(stA.smp * 50 + stA.s) * 3200u);

  float o_mean = 0.f, o_std = 1.f, o_mean_prev = 0.f, o_std_prev = 1.f;
  unsigned bad_count = 0, bad_wave = 0;
  // deferred D epilogues (fast body): accumulators and where they go
  f32x4 pend_acc[3];
  bool pend = false;
  unsigned pend_obase = 0;          // element offset of (sample, row 8 s - 1) of the pending strip
  float pend_mean = 0.f, pend_std = 1.f;
  int d_ooff[3];                    // per item: element offset of its row pair and 16-tile group inside the strip's rows: rp * 800 + 128 j4
#pragma unroll
  for (int i = 0; i < 3; ++i) d_ooff[i] = ((d_first + i) >> 2) * 800 + 128 * ((d_first + i) & 3);
  const unsigned long long d_on_mask[2] = {~0ull, __ballot(d_last_on)};   // lanes that hold pixels: every lane, or tiles 48, 49 only (item 3 of a row pair)

  // D epilogue of one fast-path item, in two parts: the vector work (de-standardise, guard) and the store.  Lane (n = lane & 15, kg):
  // tile 16 j4 + d_tsel, row kg >> 1 of the pair, pixels 4 (kg & 1) .. + 3.  The guard first asks ONE question of the four values
  // (0 * v summed is NaN iff one of them is not finite: 4 fma + 1 compare); the per-value zero-fill and count run only then.
  auto d_finish = [&](const f32x4& acc, const int item, const float mean, const float sdv, float (&v)[4]) {
#pragma unroll
    for (int rr = 0; rr < 4; ++rr) v[rr] = __builtin_fmaf(acc[rr], sdv, mean);   // without aff_out: std = 1, mean = 0 -> v exactly
    if (p.nan_guard) {
      const float t = __builtin_fmaf(v[3], 0.f, __builtin_fmaf(v[2], 0.f, __builtin_fmaf(v[1], 0.f, v[0] * 0.f)));
      if (__ballot(t != t) != 0ull) {
        const unsigned long long on = d_on_mask[(item & 3) == 3 ? 1 : 0];
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
  };
  auto d_store = [&](const float (&v)[4], const int item, const unsigned obase, const int ooff) {
    const unsigned eo = obase + (unsigned)ooff;                     // < 2^28 elements: at most 1024 samples per launch
    if ((item & 3) < 3 || d_last_on) {
      char* o = reinterpret_cast<char*>(p.out) + (size_t)(eo * OUTSZ + (unsigned)d_olane);
      if (OUT == 0) *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
      else if (OUT == 1) *reinterpret_cast<uint2*>(o) = make_uint2(pack2<false>(v[0], v[1]), pack2<false>(v[2], v[3]));
      else *reinterpret_cast<uint2*>(o) = make_uint2(pack2<true>(v[0], v[1]), pack2<true>(v[2], v[3]));
    }
  };
  auto flush_pending = [&]() {
    if (!pend) return;
    if (TS_ABL(32) && !clsY) { pend = false; return; }
    float v[4];
    if (clsY) { d_finish(pend_acc[0], d_first, pend_mean, pend_std, v); d_store(v, d_first, pend_obase, d_ooff[0]); }
    else {
#pragma unroll
      for (int i = 0; i < 3; ++i) { d_finish(pend_acc[i], d_first + i, pend_mean, pend_std, v); if (!TS_ABL(8)) d_store(v, d_first + i, pend_obase, d_ooff[i]); else asm volatile("" :: "v"(v[0]), "v"(v[1]), "v"(v[2]), "v"(v[3])); }
    }
    pend = false;
  };

  // diagnostic build: cycles between leaving a barrier and arriving at the next (work) and inside the barrier (wait), fast / slow rounds apart
  unsigned long long t_work[2] = {0, 0}, t_wait[2] = {0, 0}, n_rounds[2] = {0, 0}, t_mark = 0;
  unsigned long long t_blk[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, t_last = 0;   // fast rounds: cycles per section (top, B1 .. B8, rest)
  auto stamp = [&](const int i) {
    if (PROF) {
      pin();
      const unsigned long long t = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      t_blk[i] += t - t_last; t_last = t;
      pin();
    }
  };
  if (PROF) t_mark = __builtin_amdgcn_s_memtime();
  for (int r = 0; r <= G + 2 && K > 0; ++r) {
    // ---------------- stage state of this round (wave-uniform): A strip r, BC strip r - 1, D strip r - 2 ----------------
    const bool a_on = clsY && stA.valid, a_compute = a_on && stA.s >= 0;
    const int gb = r - 1;
    const bool bc_on = stB.valid && stB.s >= 0;       // a top segment's warm-up strip lies above the image
    const int gd = r - 2;
    const bool d_on = gd >= 0;
    const int sd = stD.valid ? stD.sl : 0;            // (the flush round gd == G counts as the first strip of a virtual sample, as in round 2)
    const int sample_d = stD.smp, s_d = stD.s;
    const int prev_smp = stP.smp;
    const bool prev_ends = d_on && sd == 0 && stP.valid && stP.seg == S - 1;   // the virtual sample before this one ended a real sample
    if (d_on && sd == 0 && p.aff_out) {  // entering a virtual sample: rotate the de-standardisation scalars
      o_mean_prev = o_mean; o_std_prev = o_std;
      if (stD.valid) {
        o_mean = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.aff_out[2 * sample_d])));
        o_std = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, p.aff_out[2 * sample_d + 1])));
      }
    }
    const bool d_warm = gd == G || (S > 1 && sd == 0);   // no regular output rows: flush round, or a segment's warm-up strip
    const bool d_top = !d_warm && s_d == 0;              // first strip of a real sample: row pair 0 = (nothing | row 0)
    // row pair 0 is a seam (or absent) whenever the row above belongs to another sample or lies above the image
    const bool d_regular = d_on && !d_warm && (clsY ? !(sd == 0 || s_d == 0) : true);
    const unsigned d_obase = (unsigned)(sample_d * 160000 + (8 * s_d - 1) * 400);   // used by regular rounds only (8 s - 1 + 2 rp >= 1 there)

    // ---------------- shared pieces ----------------
    // the next strip's input, one round ahead, into registers
    auto a_prefetch_async = [&]() { if (a_on && stN.valid && stN.s >= 0) a_load_async((unsigned)(stN.smp * 50 + stN.s) * 3200u); };
    auto a_prefetch_sync = [&]() { if (a_on && stN.valid && stN.s >= 0) a_load_sync((unsigned)(stN.smp * 50 + stN.s) * 3200u); };
    auto a_store = [&](const uint32_t (&f2)[8], const int ct) {
      if (a_valid[ct]) {
        char* dst = l100 + (r & 1) * T_L100_BUF + a_dst[ct];
#pragma unroll
        for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(dst + q * (212 * 16)) = make_uint2(f2[2 * q], f2[2 * q + 1]);
      }
    };
    // ring rows of a BC item: 8g + 4a + 2 m3 + {0, 1} (a = 100-level row of the lane's pixel)
    // ring rows of a BC item: (8 g + 4 a + 2 m3 + {0, 1}) mod 18 (a = 100-level row of the lane's pixel): the lane's constant part
    // (granule, 4 a rows) + the wave-uniform slot, wrapped at 18 rows with one unsigned min (v - 18 rows wraps around to a huge value
    // unless v is past the end)
    auto bc_rows = [&](const PixTile& t, const int m3, char*& w0, char*& w1) {
      unsigned v0 = (unsigned)(t.woff + (slotB + 2 * m3) * T_ROWP);
      v0 = min(v0, v0 - (unsigned)(T_RING_ROWS * T_ROWP));
      unsigned v1 = v0 + (unsigned)T_ROWP;
      v1 = min(v1, v1 - (unsigned)(T_RING_ROWS * T_ROWP));
      w0 = ring + v0; w1 = ring + v1;
    };
    auto ring_store = [&](const PixTile& t, char* w0, char* w1, const uint32_t (&f4)[8], const int tt) {
      if (t.valid) {
#pragma unroll
        for (int q = 0; q < 4; ++q)   // register pair q: row q >> 1 of the tap's 2x2 block, plane 2 tt + (q & 1)
          *reinterpret_cast<uint2*>(((q >> 1) ? w1 : w0) + (2 * tt + (q & 1)) * (T_PLANE * 16)) = make_uint2(f4[2 * q], f4[2 * q + 1]);
      }
    };
    auto c3_operands = [&](const PixTile& t, uint4& b0, uint4& b1) {
      const char* src = l100 + (gb & 1) * T_L100_BUF;
      b0 = *reinterpret_cast<const uint4*>(src + t.l0);
      b1 = *reinterpret_cast<const uint4*>(src + t.l0 + 2 * 212 * 16);
    };
    // D: LDS addresses of an item's window pieces.  The window rows of row pair rp sit in ring slots (8 gd + 16 + 2 rp + k) % 18,
    // k = 0..3 (scalar); a lane's k-group reads rows dy and dy + 2: two selects, then the four base addresses (row x column pair
    // family) every piece of the item is an immediate offset from
    struct DAddr { const char *p0a, *p1a, *p0b, *p1b; };
    auto d_addr = [&](const int it) {
      const int rp = (d_first + it) >> 2;
      int sa = slotD + 16 + 2 * rp;                     // slot of the first window row: (8 gd + 16 + 2 rp) % 18
      sa = sa >= 2 * T_RING_ROWS ? sa - 2 * T_RING_ROWS : sa >= T_RING_ROWS ? sa - T_RING_ROWS : sa;
      unsigned v0 = (unsigned)(sa * T_ROWP + d_c0[it]);                 // + the lane's window row dy and first column family
      v0 = min(v0, v0 - (unsigned)(T_RING_ROWS * T_ROWP));
      unsigned v1 = v0 + 2u * T_ROWP;                                   // window rows dy + 2
      v1 = min(v1, v1 - (unsigned)(T_RING_ROWS * T_ROWP));
      DAddr a;
      a.p0a = smem + v0; a.p1a = smem + v0 + d_delta;
      a.p0b = smem + v1; a.p1b = smem + v1 + d_delta;
      return a;
    };
    // operand pair cp (0..4) of window-row half `half`: the activation window piece and the Toeplitz weights
    auto d_pair = [&](const DAddr& a, const int half, const int cp, uint4& av, uint4& wv) {
      const char* p0 = half ? a.p0b : a.p0a;
      const char* p1 = half ? a.p1b : a.p1a;
      av = *reinterpret_cast<const uint4*>(cp == 0 ? p0 : cp == 4 ? p0 + 16 : p1 + (cp - 1) * (2 * T_PLANE * 16));
      wv = wc_f[(half * 5 + cp) * 64 + lane];
    };

    // ---------------- slow body: one stage after the other (pipeline fill / drain, seams) ----------------
    auto bc_item_seq = [&](const PixTile& t, const int m3, const bool do_t0, const bool do_t1) {
      uint4 b0, b1;
      c3_operands(t, b0, b1);
      f32x16 acc3 = load_bias16(cst + TC_OFF_B3 + h * 64);
      acc3 = mfma32<F16>(w3_f[(m3 * 2 + 0) * 64 + lane], b0, acc3);
      acc3 = mfma32<F16>(w3_f[(m3 * 2 + 1) * 64 + lane], b1, acc3);
      uint32_t f3[8];
      swish_pack16<F16>(acc3, f3);
      char *w0, *w1;
      bc_rows(t, m3, w0, w1);
#pragma unroll
      for (int tt = 0; tt < 2; ++tt) {
        if (!(tt ? do_t1 : do_t0)) continue;
        const uint4 bf = make_uint4(f3[4 * tt], f3[4 * tt + 1], f3[4 * tt + 2], f3[4 * tt + 3]);
        f32x16 acc4 = mfma32<F16>(w4, bf, load_bias16(cst + TC_OFF_B4 + h * 64));
        uint32_t f4[8];
        swish_pack16<F16>(acc4, f4);
        ring_store(t, w0, w1, f4, tt);
      }
    };
    // general D item (also the seam rows of a sample / segment boundary): as in round 2
    auto do_d = [&](const int item) {
      const int rp = item >> 2, j4 = item & 3;
      const bool emit_prev = rp == 0 && sd == 0 && prev_ends;   // (row 399 of the sample that just ended | nothing)
      const bool emit_top = rp == 0 && d_top;                   // (nothing | row 0 of the sample that starts)
      const bool seam = emit_prev || emit_top || d_warm;
      if (d_warm && !emit_prev) return;
      const int kg = lane >> 4;
      const int d_dy = (lane >> 4) & 1, d_dx = lane >> 5;       // window sub-pixel (dy, dx) of this lane's k-group
      int d_xo[5];                                              // byte offsets of the five window column pairs relative to granule (plane 0, tx)
#pragma unroll
      for (int cp = 0; cp < 5; ++cp)
        d_xo[cp] = 16 * (d_dx ? (cp < 4 ? 2 * cp * T_PLANE : 1) : (cp == 0 ? 7 * T_PLANE - 1 : (2 * cp - 1) * T_PLANE));
      int tx = 16 * j4 + d_tsel;
      tx = tx < 50 ? tx : 49;
      int sa = slotD + 16 + 2 * rp;                             // slot of the first window row (tall row 8g-2+2rp): (8 gd + 16 + 2 rp) % 18
      sa = sa >= 2 * T_RING_ROWS ? sa - 2 * T_RING_ROWS : sa >= T_RING_ROWS ? sa - T_RING_ROWS : sa;
      int s0 = sa + d_dy, s1 = s0 + 2;
      s0 = s0 >= T_RING_ROWS ? s0 - T_RING_ROWS : s0;
      s1 = s1 >= T_RING_ROWS ? s1 - T_RING_ROWS : s1;
      const int r0 = T_OFF_RING + s0 * T_ROWP + (tx + 1) * 16, r1 = T_OFF_RING + s1 * T_ROWP + (tx + 1) * 16;
      const bool okf = !(d_dx == 0 && tx == 0), okl = !(d_dx == 1 && tx == 49);
      const int oy = (lane >> 3) & 1, ox = lane & 7;
      const int txo = 16 * j4 + kg;
      auto conv = [&](const bool keep01, const bool keep23) {
        f32x4 acc = {conv_bias, conv_bias, conv_bias, conv_bias};
#pragma unroll
        for (int half = 0; half < 2; ++half) {
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
          float v = __builtin_fmaf(acc[rr], sdv, mean);
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
    auto slow_body = [&]() {
      flush_pending();
      if (a_on) {
        if (a_compute) {
          f32x16 acc[2];
          a_wait();
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            acc[ct] = load_bias16(cst + TC_OFF_B2 + h * 64);
#pragma unroll
            for (int kk = 0; kk < 4; ++kk) acc[ct] = mfma32<F16>(wa[kk], __builtin_bit_cast(uint4, xb[ct][kk]), acc[ct]);
          }
          asm volatile("s_nop 7\n\ts_nop 7" ::: "memory");   // the MFMAs above have read xb before the (inline asm, unmodelled) loads below may overwrite it
          a_prefetch_sync();
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            uint32_t f2[8];
            swish_pack16<F16>(acc[ct], f2);
            a_store(f2, ct);
          }
        } else a_prefetch_sync();
      }
      if (bc_on) {
        if (clsY) { bc_item_seq(tf, m3_f, true, true); bc_item_seq(th, m3_h, tt_h == 0, tt_h == 1); }
        else { bc_item_seq(tf, 0, true, true); bc_item_seq(tf, 1, true, true); }
      }
      if (d_on) {
        do_d(d_first);
        if (d_cnt > 1) { do_d(d_first + 1); do_d(d_first + 2); }
      }
    };

    // ---------------- fast body, X: pixel tile wq for both tap rows + three D items ----------------
    // Blocks in issue order (a block = one swish_pack_s; [..] = what its hooks carry, 4 hooks per 8-register block, 8 per 16):
    //   top: ConvT#3 tap row 0 (2 MFMAs), last round's D epilogues
    //   B1 swish(acc3a lo) -> MFMA ConvT#4 tap 0      B2 swish(acc3a hi) -> MFMA tap 1
    //   B3 swish(acc4[0]) [ConvT#3 tap row 1]          B4 swish(acc4[1])
    //   B5 swish(acc3b lo) -> MFMA tap 0               B6 swish(acc3b hi) -> MFMA tap 1
    //   B7 swish(acc4[2])                               B8 swish(acc4[3])
    // D: operand pair q (item q / 10, window-row half (q / 5) & 1, piece q % 5) is read from LDS at hook q + 1 and multiplied at
    // hook q + 4 (three pairs in flight); hooks are numbered 0..47 through the blocks.
    auto fast_x = [&]() {
      uint4 b0, b1;
      c3_operands(tf, b0, b1);
      uint4 av[4], wv[4];
      const DAddr da[3] = {d_addr(0), d_addr(1), d_addr(2)};
      f32x16 acc3a = load_bias16(cst + TC_OFF_B3 + h * 64), acc3b;
      const uint4 w3a0 = w3_f[0 * 64 + lane], w3a1 = w3_f[1 * 64 + lane];
      d_pair(da[0], 0, 0, av[0], wv[0]);
      f32x16 acc4[2];
      acc4[0] = load_bias16(cst + TC_OFF_B4 + h * 64);   // every accumulator's bias is read a block ahead of its MFMA: with two waves
      pin();                                             // per SIMD nothing else covers an LDS round trip in front of an MFMA
      flush_pending();          // last round's D epilogues: vector work + stores with no dependency on anything this round has started,
      pin();                    // under the LDS reads above
      acc3a = mfma32<F16>(w3a0, b0, acc3a);
      acc3a = mfma32<F16>(w3a1, b1, acc3a);
      pin();
      stamp(0);
      f32x4 dacc[3];
      auto sched = [&](auto s) {
        constexpr int H = decltype(s)::value;
        if constexpr (H >= 1 && H <= 29) {             // D operand pair q = H is read from LDS here ...  (pair 0: at the top)
          constexpr int Q = H, IT = Q / 10, HF = (Q / 5) & 1, CP = Q % 5;
          if (!TS_ABL(16)) d_pair(da[IT], HF, CP, av[Q & 3], wv[Q & 3]);
        }
        if constexpr (H >= 3 && H <= 32) {             // ... and multiplied three hooks later
          constexpr int Q = H - 3, IT = Q / 10, HF = (Q / 5) & 1, CP = Q % 5;
          if constexpr (CP == 0 && HF == 0) dacc[IT] = f32x4{conv_bias, conv_bias, conv_bias, conv_bias};
          if (!TS_ABL(16)) dacc[IT] = mfma16<F16>(wv[Q & 3], av[Q & 3], dacc[IT]);
        }
        if constexpr (H == 0) acc4[1] = load_bias16(cst + TC_OFF_B4 + h * 64);
        if constexpr (H == 6) acc3b = load_bias16(cst + TC_OFF_B3 + h * 64);
        if constexpr (H == 8) acc3b = mfma32<F16>(w3_f[2 * 64 + lane], b0, acc3b);
        if constexpr (H == 9) acc3b = mfma32<F16>(w3_f[3 * 64 + lane], b1, acc3b);
      };
      uint32_t f3[8], f4[8];
      char *w0, *w1;
      bc_rows(tf, 0, w0, w1);
      swish_pack_s<F16, 0, 8>(acc3a, f3, one2, [&](auto k) { sched(ic<0 + decltype(k)::value>()); });
      pin(); acc4[0] = mfma32<F16>(w4, make_uint4(f3[0], f3[1], f3[2], f3[3]), acc4[0]); pin();
      stamp(1);
      swish_pack_s<F16, 8, 8>(acc3a, f3 + 4, one2, [&](auto k) { sched(ic<4 + decltype(k)::value>()); });
      pin(); acc4[1] = mfma32<F16>(w4, make_uint4(f3[4], f3[5], f3[6], f3[7]), acc4[1]); pin();
      stamp(2);
      swish_pack_s<F16, 0, 16>(acc4[0], f4, one2, [&](auto k) { sched(ic<8 + decltype(k)::value>()); });
      ring_store(tf, w0, w1, f4, 0);
      stamp(3);
      acc4[0] = load_bias16(cst + TC_OFF_B4 + h * 64);
      swish_pack_s<F16, 0, 16>(acc4[1], f4, one2, [&](auto k) { sched(ic<16 + decltype(k)::value>()); });
      ring_store(tf, w0, w1, f4, 1);
      stamp(4);
      acc4[1] = load_bias16(cst + TC_OFF_B4 + h * 64);
      bc_rows(tf, 1, w0, w1);
      swish_pack_s<F16, 0, 8>(acc3b, f3, one2, [&](auto k) { sched(ic<24 + decltype(k)::value>()); });
      pin(); acc4[0] = mfma32<F16>(w4, make_uint4(f3[0], f3[1], f3[2], f3[3]), acc4[0]); pin();
      stamp(5);
      swish_pack_s<F16, 8, 8>(acc3b, f3 + 4, one2, [&](auto k) { sched(ic<28 + decltype(k)::value>()); });
      pin(); acc4[1] = mfma32<F16>(w4, make_uint4(f3[4], f3[5], f3[6], f3[7]), acc4[1]); pin();
      stamp(6);
      swish_pack_s<F16, 0, 16>(acc4[0], f4, one2, [&](auto k) { sched(ic<32 + decltype(k)::value>()); });
      ring_store(tf, w0, w1, f4, 0);
      stamp(7);
      swish_pack_s<F16, 0, 16>(acc4[1], f4, one2, [&](auto) {});
      ring_store(tf, w0, w1, f4, 1);
      stamp(8);
#pragma unroll
      for (int i = 0; i < 3; ++i) pend_acc[i] = dacc[i];
      pend = true; pend_obase = d_obase; pend_mean = o_mean; pend_std = o_std;
    };

    // ---------------- fast body, Y: one full BC item, the half item, ConvT#2 tap wq (both column tiles), one D item ----------------
    //   top: ConvT#3 of the full item (2 MFMAs), last round's D epilogue
    //   B1 swish(acc3f lo) [ConvT#2 column tile 0, 4 MFMAs] -> MFMA ConvT#4 tap 0
    //   B2 swish(acc3f hi) [ConvT#2 column tile 1, 4 MFMAs] -> MFMA tap 1
    //   B3 swish(acc_a[0]) [next strip's input; D pairs]      B4 swish(acc_a[1]) [D]      -> 100-level tile
    //   B5 swish(acc4a) [ConvT#3 of the half item; D]
    //   B6 swish(acc3h, the half item's eight registers) -> MFMA ConvT#4 of its tap
    //   B7 swish(acc4b)                                         B8 swish(acc4h)
    auto fast_y = [&]() {
      uint4 b0, b1, hb0, hb1;
      c3_operands(tf, b0, b1);
      c3_operands(th, hb0, hb1);
      uint4 av[4], wv[4];
      const DAddr da = d_addr(0);
      f32x16 acc3f = load_bias16(cst + TC_OFF_B3 + h * 64), acc3h, acc_a[2], acc4a, acc4b, acc4h;
      const uint4 w3f0 = w3_f[(m3_f * 2 + 0) * 64 + lane], w3f1 = w3_f[(m3_f * 2 + 1) * 64 + lane];
      acc_a[0] = load_bias16(cst + TC_OFF_B2 + h * 64);
      acc_a[1] = load_bias16(cst + TC_OFF_B2 + h * 64);
      acc4a = load_bias16(cst + TC_OFF_B4 + h * 64);
      pin();
      // last round's D epilogue: its vector part here, under the LDS reads above; its STORE only behind the ConvT#2 MFMAs (hook 8) --
      // vmcnt counts loads and stores in one queue, and the wait for this round's prefetched input (issued a round ago, long
      // complete) must not find a store that was issued a moment ago in front of it
      float pv[4];
      const bool had_pend = pend;
      const unsigned pv_obase = pend_obase;
      if (pend) { d_finish(pend_acc[0], d_first, pend_mean, pend_std, pv); pend = false; }
      a_wait();                 // this round's ConvT#2 input: read a round ago
      pin();
      acc3f = mfma32<F16>(w3f0, b0, acc3f);
      acc3f = mfma32<F16>(w3f1, b1, acc3f);
      pin();
      stamp(0);
      f32x4 dacc;
      auto sched = [&](auto s) {
        constexpr int H = decltype(s)::value;
        if constexpr (H < 8) {                         // B1, B2: ConvT#2
          constexpr int CT = H >> 2, KK = H & 3;
          acc_a[CT] = mfma32<F16>(wa[KK], __builtin_bit_cast(uint4, xb[CT][KK]), acc_a[CT]);
        }
        if constexpr (H == 1) acc4b = load_bias16(cst + TC_OFF_B4 + h * 64);
        if constexpr (H == 8) { if (had_pend && !TS_ABL(2)) d_store(pv, d_first, pv_obase, d_ooff[0]); if (!TS_ABL(1)) a_prefetch_async(); }
        if constexpr (H >= 9 && H <= 18) {             // D pair q = H - 9 read ...
          constexpr int Q = H - 9, HF = Q / 5, CP = Q % 5;
          if (!TS_ABL(4)) d_pair(da, HF, CP, av[Q & 3], wv[Q & 3]);
        }
        if constexpr (H >= 12 && H <= 21) {            // ... and multiplied three hooks later
          constexpr int Q = H - 12;
          if constexpr (Q == 0) dacc = f32x4{conv_bias, conv_bias, conv_bias, conv_bias};
          if (!TS_ABL(4)) dacc = mfma16<F16>(wv[Q & 3], av[Q & 3], dacc);
        }
        if constexpr (H == 24) acc3h = load_bias16(cst + TC_OFF_B3 + h * 64);
        if constexpr (H == 26) acc3h = mfma32<F16>(w3_f[(m3_h * 2 + 0) * 64 + lane], hb0, acc3h);
        if constexpr (H == 27) acc3h = mfma32<F16>(w3_f[(m3_h * 2 + 1) * 64 + lane], hb1, acc3h);
        if constexpr (H == 30) acc4h = load_bias16(cst + TC_OFF_B4 + h * 64);
      };
      uint32_t f3[8], f3h[4], f4[8], f2[8];
      char *w0f, *w1f, *w0h, *w1h;
      bc_rows(tf, m3_f, w0f, w1f);
      swish_pack_s<F16, 0, 8>(acc3f, f3, one2, [&](auto k) { sched(ic<0 + decltype(k)::value>()); });
      pin(); acc4a = mfma32<F16>(w4, make_uint4(f3[0], f3[1], f3[2], f3[3]), acc4a); pin();
      stamp(1);
      swish_pack_s<F16, 8, 8>(acc3f, f3 + 4, one2, [&](auto k) { sched(ic<4 + decltype(k)::value>()); });
      pin(); acc4b = mfma32<F16>(w4, make_uint4(f3[4], f3[5], f3[6], f3[7]), acc4b); pin();
      stamp(2);
      swish_pack_s<F16, 0, 16>(acc_a[0], f2, one2, [&](auto k) { sched(ic<8 + decltype(k)::value>()); });
      a_store(f2, 0);
      stamp(3);
      swish_pack_s<F16, 0, 16>(acc_a[1], f2, one2, [&](auto k) { sched(ic<16 + decltype(k)::value>()); });
      a_store(f2, 1);
      stamp(4);
      swish_pack_s<F16, 0, 16>(acc4a, f4, one2, [&](auto k) { sched(ic<24 + decltype(k)::value>()); });
      ring_store(tf, w0f, w1f, f4, 0);
      stamp(5);
      // the half item activates only the eight accumulator registers of ITS ConvT#4 tap: registers 8 tt .. 8 tt + 7 (tt is wave-uniform)
      f32x16 sel;
#pragma unroll
      for (int i = 0; i < 8; ++i) sel[i] = tt_h ? acc3h[8 + i] : acc3h[i];
      swish_pack_s<F16, 0, 8>(sel, f3h, one2, [&](auto) {});
      pin(); acc4h = mfma32<F16>(w4, make_uint4(f3h[0], f3h[1], f3h[2], f3h[3]), acc4h); pin();
      stamp(6);
      swish_pack_s<F16, 0, 16>(acc4b, f4, one2, [&](auto) {});
      ring_store(tf, w0f, w1f, f4, 1);
      stamp(7);
      bc_rows(th, m3_h, w0h, w1h);
      swish_pack_s<F16, 0, 16>(acc4h, f4, one2, [&](auto) {});
      ring_store(th, w0h, w1h, f4, tt_h);
      stamp(8);
      pend_acc[0] = dacc;
      pend = true; pend_obase = d_obase; pend_mean = o_mean; pend_std = o_std;
    };

    const bool fast = clsY ? (a_compute && bc_on && d_regular) : (bc_on && d_regular);
    if (PROF) t_last = __builtin_amdgcn_s_memtime();
    if (clsY) {
      if (TS_ABL(256)) {} else if (fast) fast_y(); else slow_body();      // diagnostic 256 / 128: this class of waves does nothing
    } else {
      if (TS_ABL(128)) {} else if (fast) fast_x(); else slow_body();
    }
    // advance the strip state: every stage moves one strip on, strip r + 2 is derived from r + 1
    stP = stD; stD = stB; stB = stA; stA = stN;
    slotD = slotB; slotB = slotA; slotA = slotA + 8 >= T_RING_ROWS ? slotA + 8 - T_RING_ROWS : slotA + 8;
    {
      stN.valid = r + 2 < G;
      if (++stN.sl == SL) {
        stN.sl = 0; ++kvN;
        const int vs = (int)blockIdx.x + kvN * (int)gridDim.x;
        stN.smp = vs / S;
        stN.seg = vs - stN.smp * S;
      }
      stN.s = S > 1 ? stN.seg * L + stN.sl - 1 : stN.sl;
    }
    if (PROF) {
      const unsigned long long t1 = __builtin_amdgcn_s_memtime();
      lds_barrier();
      const unsigned long long t2 = __builtin_amdgcn_s_memtime();
      t_work[fast ? 0 : 1] += t1 - t_mark; t_wait[fast ? 0 : 1] += t2 - t1; n_rounds[fast ? 0 : 1] += 1;
      t_mark = t2;
    } else lds_barrier();
  }
  if (PROF && p.prof && blockIdx.x == 0 && lane == 0) {
    unsigned long long* o = p.prof + wave * 6;
    o[0] = t_work[0]; o[1] = t_wait[0]; o[2] = n_rounds[0]; o[3] = t_work[1]; o[4] = t_wait[1]; o[5] = n_rounds[1];
    for (int i = 0; i < 10; ++i) p.prof[48 + wave * 10 + i] = t_blk[i];
  }
  flush_pending();
  if (p.nan_guard && p.nonfinite && bad_count) atomicAdd(p.nonfinite, (unsigned long long)bad_count);
  if (p.nan_guard && p.nonfinite && bad_wave && lane == 0) atomicAdd(p.nonfinite, (unsigned long long)bad_wave);
}

hipError_t launch_tail16s(bool f16, const TailParams& p, int blocks, hipStream_t s) {
  if (p.n == 0) return hipSuccess;
  void (*fn)(TailParams) = nullptr;
  const bool seg = p.seg > 1;
#define PICK(F, O) fn = seg ? tail16s<F, O, true> : tail16s<F, O, false>
  if (f16) { if (p.out_dtype == SRCFD_F32) PICK(true, 0); else if (p.out_dtype == SRCFD_BF16) PICK(true, 1); else PICK(true, 2); }
  else { if (p.out_dtype == SRCFD_F32) PICK(false, 0); else if (p.out_dtype == SRCFD_BF16) PICK(false, 1); else PICK(false, 2); }
#undef PICK
#ifdef SRCFD_DIAG
  if (p.prof && !f16 && p.out_dtype == SRCFD_F32) fn = seg ? tail16s<false, 0, true, true> : tail16s<false, 0, false, true>;
#endif
  hipError_t e = lds_attr_once(reinterpret_cast<const void*>(fn), T_LDS_BYTES);
  if (e != hipSuccess) return e;
  hipLaunchKernelGGL(fn, dim3(blocks), dim3(512), T_LDS_BYTES, s, p);
  return hipGetLastError();
}

}  // namespace srcfd
