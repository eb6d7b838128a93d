// LDS layout of the fused 16-bit tail, shared by its two implementations: `tail16` in kernels_bf16.hip (the shipped kernel) and
// `tail16s` in kernels_tail16s.hip (SRCFD_TAIL=s, the A/B arm the parity tests compare it with bit for bit; measured slower).
//  ring (400-level, 16 B per pixel): a row is 8 planes (x & 7) of 52 granules (1 + (x >> 3); granules 0 and 51 stay zero: the
//    SAME padding of the output conv at the left / right image edge, read like any other pixel -- no per-lane edge tests).
//    BC writes a wave of pixels 4 apart in x (-> consecutive granules of two planes, 2-way at worst); D reads 16 tiles 8 apart
//    in x (-> 16 consecutive granules, conflict-free; the row pitch is a multiple of 256 B so the two window rows of one
//    ds_read_b128 lane group interleave).
//  L100 (100-level, 4 chunks of 8 channels): [chunk][row][x parity][x >> 1]; A writes pixels 2 apart (-> consecutive), BC reads
//    consecutive pixels (parity planes 56 granules = 8 mod 16 apart -> conflict-free).
#pragma once
#include "dev16.h"
#include "kernels16.h"

namespace srcfd {

constexpr int T_RING_ROWS = 18, T_PLANE = 52, T_ROWP = 8 * T_PLANE * 16;   // 50 granules + a zero granule at either end of a plane
constexpr int T_OFF_RING = 0;
constexpr int T_OFF_L100 = T_RING_ROWS * T_ROWP;            // 119808
constexpr int T_L100_BUF = 4 * 212 * 16;                    // 13568
constexpr int T_OFF_CONST = T_OFF_L100 + 2 * T_L100_BUF;    // blob copied from TailParams::consts
constexpr int T_OFF_CTR = T_OFF_CONST + TAIL_CONST_BYTES;
constexpr int T_OFF_ZERO = T_OFF_CTR + 16;                  // 16 zero bytes: what out-of-image window pixels read
constexpr int T_LDS_BYTES = T_OFF_ZERO + 16;
static_assert(T_LDS_BYTES <= 160 * 1024, "tail kernel LDS budget");

__device__ __forceinline__ int ring_off(int Y, int X) { return (Y % T_RING_ROWS) * T_ROWP + (((X & 7) * T_PLANE + (X >> 3) + 1) << 4); }
__device__ __forceinline__ int l100_off(int a, int x, int chunk) { return (chunk * 212 + a * 106 + (x & 1) * 56 + (x >> 1)) << 4; }

// block-wide barrier that leaves global loads / stores in flight: only LDS traffic
// has to be complete before the other waves may look at it
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  asm volatile("" ::: "memory");
}

}  // namespace srcfd
