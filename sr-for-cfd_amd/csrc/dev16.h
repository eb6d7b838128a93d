// Device-side helpers shared by the 16-bit kernels (kernels_bf16.hip, kernels_mid16.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

#include "../../include/srcfd.h"

namespace srcfd {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h16x2 __attribute__((ext_vector_type(2)));

template <bool F16>
__device__ __forceinline__ f32x16 mfma32(const uint4& a, const uint4& b, const f32x16& c) {
  if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma16(const uint4& a, const uint4& b, const f32x4& c) {
  if (F16) return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h16x8, a), __builtin_bit_cast(h16x8, b), c, 0, 0, 0);
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(s16x8, a), __builtin_bit_cast(s16x8, b), c, 0, 0, 0);
}

template <bool F16>
__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  f32x2 v = {lo, hi};
  if (F16) return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, h16x2));
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));  // v_cvt_pk_bf16_f32, RNE
}

// u = log2e * x  ->  log2e * swish(x)
__device__ __forceinline__ float swish_scaled(float u) {
  float e = __builtin_amdgcn_exp2f(-u);
  return u * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float act16(float u, int act) { return act == SRCFD_ACT_SWISH ? swish_scaled(u) : u; }

// 16 swish + pack, issued as four batches of 16 independent instructions (exp, add, rcp, mul):
// hipcc interleaves the four dependent steps of neighbouring elements, and with only four waves per
// SIMD the in-order issue then stalls on every step (measured 25 cycles per 64 activations); in
// batch order no instruction waits on one issued fewer than 16 slots earlier (~14 cycles).
// Measured alternatives that did NOT help at this kernel's 4 waves per SIMD (tools/microbench4.hip, profiles/r01):
// v_pk_add_f32 / v_pk_mul_f32 halve the instruction count but issue every 14 cycles per wave against 5.8 for the
// plain forms; batches of 8 instead of 16 look 34 % faster when all waves run in lockstep and identical in the
// real kernels, where the waves are in different phases.
template <bool F16>
__device__ __forceinline__ void swish_pack16(const f32x16& dd, uint32_t (&o)[8], bool skip = false) {
  if (skip) {
#pragma unroll
    for (int i = 0; i < 8; ++i) o[i] = pack2<F16>(dd[2 * i], dd[2 * i + 1]);
    return;
  }
  // exp2 and rcp per element (quarter rate, 8.5 cycles per wave instruction); the add and the multiply as packed f32
  // pairs: 5.5 cycles per v_pk_* instruction = 2.7 per element against 4.5 for the plain forms (tools/microbench8.hip)
  f32x2 u2[8], e2[8];
  const f32x2 one2 = {1.0f, 1.0f};
#pragma unroll
  for (int i = 0; i < 8; ++i) u2[i] = f32x2{dd[2 * i], dd[2 * i + 1]};
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    e2[i].x = __builtin_amdgcn_exp2f(-u2[i].x);  // builtin: hipcc pads the MFMA -> VALU read hazard itself
    e2[i].y = __builtin_amdgcn_exp2f(-u2[i].y);
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(e2[i]) : "v"(one2));
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].x));
    asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].y));
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(e2[i]) : "v"(u2[i]));
#pragma unroll
  for (int i = 0; i < 8; ++i) o[i] = pack2<F16>(e2[i].x, e2[i].y);
}

// The same swish + pack over NR (8 or 16) accumulator registers starting at R0, with four places where the caller may put
// instructions of ANOTHER dependency chain -- MFMAs of the next stage -- into this wave's own instruction stream: after the first
// and the second half of the exps, and after the first and the second half of the rcps.  Round 3 (tools/microbench9.hip,
// profiles/r03): a v_mfma_f32_32x32x16_bf16 issued between the transcendentals of the same wave costs 3.6 ns of SIMD time against
// 10.7 ns when it is issued in front of the block and waited for -- the matrix pipe runs under the wave's own vector stream, which
// it does not do under another wave's (profiles/r02/d_...).  A hook is `[&] { pin(); acc = mfma32(...); pin(); }` with
// pin() = __builtin_amdgcn_sched_barrier(0): hipcc may otherwise move the builtin MFMA anywhere its operands allow.  The MFMA stays a
// builtin so that hipcc pads its hazards (VALU write -> MFMA operand, MFMA result -> VALU read) itself.
struct no_hook { __device__ __forceinline__ void operator()() const {} };
__device__ __forceinline__ void pin() { __builtin_amdgcn_sched_barrier(0); }

template <bool F16, int R0, int NR, class H0 = no_hook, class H1 = no_hook, class H2 = no_hook, class H3 = no_hook>
__device__ __forceinline__ void swish_pack_h(const f32x16& dd, uint32_t* o, H0&& h0 = no_hook(), H1&& h1 = no_hook(), H2&& h2 = no_hook(),
                                             H3&& h3 = no_hook()) {
  constexpr int NP = NR / 2;
  f32x2 u2[NP], e2[NP];
  const f32x2 one2 = {1.0f, 1.0f};
#pragma unroll
  for (int i = 0; i < NP; ++i) u2[i] = f32x2{dd[R0 + 2 * i], dd[R0 + 2 * i + 1]};
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) {
    e2[i].x = __builtin_amdgcn_exp2f(-u2[i].x);
    e2[i].y = __builtin_amdgcn_exp2f(-u2[i].y);
  }
  h0();
#pragma unroll
  for (int i = NP / 2; i < NP; ++i) {
    e2[i].x = __builtin_amdgcn_exp2f(-u2[i].x);
    e2[i].y = __builtin_amdgcn_exp2f(-u2[i].y);
  }
  h1();
#pragma unroll
  for (int i = 0; i < NP; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(e2[i]) : "v"(one2));
#pragma unroll
  for (int i = 0; i < NP / 2; ++i) {
    asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].x));
    asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].y));
  }
  h2();
#pragma unroll
  for (int i = NP / 2; i < NP; ++i) {
    asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].x));
    asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].y));
  }
  h3();
#pragma unroll
  for (int i = 0; i < NP; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(e2[i]) : "v"(u2[i]));
#pragma unroll
  for (int i = 0; i < NP; ++i) o[i] = pack2<F16>(e2[i].x, e2[i].y);
}

template <int K> using ic = std::integral_constant<int, K>;

// swish + pack over NR (8 or 16) accumulator registers starting at R0; hook(ic<k>) after every four transcendentals
// (k = 0 .. NR/2 - 1: first the exps, then the rcps).  Same instructions per element as swish_pack16 (dev16.h): bit-identical.
template <bool F16, int R0, int NR, class H>
__device__ __forceinline__ void swish_pack_s(const f32x16& dd, uint32_t* o, const f32x2& one2, H&& hook) {   // one2 = {1, 1}, kept in a register pair by the caller (hipcc otherwise rebuilds it from scalars in front of every block)
  constexpr int NP = NR / 2, NG = NP / 2;
  f32x2 u2[NP], e2[NP];
#pragma unroll
  for (int i = 0; i < NP; ++i) u2[i] = f32x2{dd[R0 + 2 * i], dd[R0 + 2 * i + 1]};
  auto exps = [&](auto g) {
    constexpr int G = decltype(g)::value;
#pragma unroll
    for (int i = 2 * G; i < 2 * G + 2; ++i) {
      e2[i].x = __builtin_amdgcn_exp2f(-u2[i].x);
      e2[i].y = __builtin_amdgcn_exp2f(-u2[i].y);
    }
    pin();
    hook(ic<G>());
    pin();
  };
  auto rcps = [&](auto g) {
    constexpr int G = decltype(g)::value;
#pragma unroll
    for (int i = 2 * G; i < 2 * G + 2; ++i) {
      asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].x));
      asm volatile("v_rcp_f32 %0, %0" : "+v"(e2[i].y));
    }
    pin();
    hook(ic<NG + G>());
    pin();
  };
  exps(ic<0>()); exps(ic<1>());
  if constexpr (NG == 4) { exps(ic<2>()); exps(ic<3>()); }
#pragma unroll
  for (int i = 0; i < NP; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(e2[i]) : "v"(one2));
  rcps(ic<0>()); rcps(ic<1>());
  if constexpr (NG == 4) { rcps(ic<2>()); rcps(ic<3>()); }
#pragma unroll
  for (int i = 0; i < NP; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(e2[i]) : "v"(u2[i]));
#pragma unroll
  for (int i = 0; i < NP; ++i) o[i] = pack2<F16>(e2[i].x, e2[i].y);
}

__device__ __forceinline__ f32x16 load_bias16(const char* base) {
  f32x16 r;
  const float4* p = reinterpret_cast<const float4*>(base);
#pragma unroll
  for (int i = 0; i < 4; ++i) { float4 v = p[i]; r[4 * i] = v.x; r[4 * i + 1] = v.y; r[4 * i + 2] = v.z; r[4 * i + 3] = v.w; }
  return r;
}

}  // namespace srcfd
