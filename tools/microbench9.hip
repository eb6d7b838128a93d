// Microbench v9: does a v_mfma_f32_32x32x16_bf16 hide under the swish stream of the SAME wave?  
// One workgroup per CU, 4 / 8 / 16 waves (1 / 2 / 4 per SIMD).  Every instruction of the measured loop is its own
// `asm volatile` statement, so the order below IS the order in the binary (checked in the .s).
// A "BC-like item" = what tail16's BC stage does for 32 pixels: 2 chained MFMAs (ConvT#3) -> swish of 16 registers -> pack
// -> 2 x (1 MFMA (ConvT#4) -> swish of 16 registers -> pack): 4 MFMAs and 48 wave-registers of swish
// (per register: v_exp_f32, add, v_rcp_f32, multiply; per pair one v_cvt_pk_bf16_f32).
//   P 0: the 4 MFMAs only                      P 1: the 3 swish blocks only
//   P 2: the item as tail16 r2 issues it: MFMA(s), wait, swish block, pack, MFMA, wait, swish block, ...
//   P 3: software-pipelined by one block: the ConvT#4 MFMAs of item i-1 sit INSIDE the first swish block of item i
//        (at slots S1, S2 of its 56), the ConvT#3 MFMAs of item i+1 inside the second block; same instruction multiset as P 2
//   P 5: a D-like item alone (20 ds_read_b128 + 10 chained v_mfma_f32_16x16x32_bf16 + 4-instruction epilogue)
//   P 6: the D-like item's MFMAs spread through the three swish blocks of a BC-like item whose own MFMAs are placed as in P 3
// PK: add / multiply as v_pk_add_f32 / v_pk_mul_f32 (8 + 8 per block) or as v_add_f32 / v_mul_f32 (16 + 16)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#define MFMA32(acc, a, b) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define MFMA16(acc, a, b) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
#define NOP12() asm volatile("s_nop 7\n\ts_nop 3")

// one swish block over 16 accumulator registers, slot by slot; hook(s) runs after slot s
template <bool PK, class H>
__device__ __forceinline__ void swish_block(const f32x16& u, float (&e)[16], uint32_t (&f)[8], H&& hook) {
  const f32x2 one2 = {1.0f, 1.0f};
  int s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) { asm volatile("v_exp_f32 %0, -%1" : "=v"(e[i]) : "v"(u[i])); hook(s++); }
  if (PK) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x2 t = {e[2 * i], e[2 * i + 1]};
      asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(t) : "v"(one2));
      e[2 * i] = t.x; e[2 * i + 1] = t.y; hook(s++);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[i])); hook(s++); }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) { asm volatile("v_rcp_f32 %0, %0" : "+v"(e[i])); hook(s++); }
  if (PK) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      f32x2 t = {e[2 * i], e[2 * i + 1]}, uu = {u[2 * i], u[2 * i + 1]};
      asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(t) : "v"(uu));
      e[2 * i] = t.x; e[2 * i + 1] = t.y; hook(s++);
    }
  } else {
#pragma unroll
    for (int i = 0; i < 16; ++i) { asm volatile("v_mul_f32 %0, %0, %1" : "+v"(e[i]) : "v"(u[i])); hook(s++); }
  }
#pragma unroll
  for (int i = 0; i < 8; ++i) { asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(f[i]) : "v"(e[2 * i]), "v"(e[2 * i + 1])); hook(s++); }
}

template <int P, bool PK, int S1, int S2, int NW>
__global__ void __launch_bounds__(64 * NW) k(uint32_t* out, unsigned long long* cyc, int iters) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 4096; i += 64 * NW) reinterpret_cast<uint4*>(smem)[i] = make_uint4(0x3c003c00u + i, 0x3c003c01u, 0x3c003c02u, 0x3c003c03u);
  __syncthreads();
  const i32x4 w3a = reinterpret_cast<const i32x4*>(smem)[lane], w3b = reinterpret_cast<const i32x4*>(smem)[64 + lane];
  const i32x4 w4 = reinterpret_cast<const i32x4*>(smem)[128 + lane];
  i32x4 b0 = reinterpret_cast<const i32x4*>(smem)[192 + tid], b1 = reinterpret_cast<const i32x4*>(smem)[1300 + tid];
  f32x16 acc3, acc4a, acc4b;
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc3[i] = 0.01f * (lane + i); acc4a[i] = 0.02f * i; acc4b[i] = -0.03f * i; }
  f32x4 dacc = {0, 0, 0, 0};
  float e[16];
  uint32_t f3[8], f4[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { f3[i] = 0x3c003c00u + lane; f4[i] = 0; }
  const int dbase0 = (lane & 15) * 16 + (lane >> 4) * 1024;   // conflict-free 16-byte reads
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  for (int it = 0; it < iters; ++it) {
    const int dbase = dbase0 + ((it & 3) << 8), wbase = 16 * lane + ((it & 1) << 12);   // loop-variant: the reads stay in the loop
    if (P == 0) {
      MFMA32(acc3, w3a, b0); MFMA32(acc3, w3b, b1);
      i32x4 lo = {(int)f3[0], (int)f3[1], (int)f3[2], (int)f3[3]}, hi = {(int)f3[4], (int)f3[5], (int)f3[6], (int)f3[7]};
      MFMA32(acc4a, w4, lo); MFMA32(acc4b, w4, hi);
    } else if (P == 1) {
      swish_block<PK>(acc3, e, f3, [](int) {});
      swish_block<PK>(acc4a, e, f4, [](int) {});
      swish_block<PK>(acc4b, e, f4, [](int) {});
    } else if (P == 2) {
      MFMA32(acc3, w3a, b0); MFMA32(acc3, w3b, b1); NOP12();
      swish_block<PK>(acc3, e, f3, [](int) {});
      i32x4 lo = {(int)f3[0], (int)f3[1], (int)f3[2], (int)f3[3]}, hi = {(int)f3[4], (int)f3[5], (int)f3[6], (int)f3[7]};
      asm volatile("s_nop 1");
      MFMA32(acc4a, w4, lo); NOP12();
      swish_block<PK>(acc4a, e, f4, [](int) {});
      MFMA32(acc4b, w4, hi); NOP12();
      swish_block<PK>(acc4b, e, f4, [](int) {});
    } else if (P == 3 || P == 6) {
      // D-like item riding along (P 6): ten (activation, weight) operand pairs; pair c is read from LDS at slot 16 c + 2 of the
      // item's 168 and multiplied 12 slots later, so at most two pairs (16 VGPRs) are live
      i32x4 da[10], dw[10];
      auto d_hook = [&](int gs) {   // gs: slot index over the three blocks
        if (P != 6) return;
#pragma unroll
        for (int c = 0; c < 10; ++c) {
          if (gs == 16 * c + 2) { da[c] = *reinterpret_cast<const i32x4*>(smem + 4096 + dbase + 4096 * (c % 5) + 2048 * (c / 5)); dw[c] = *reinterpret_cast<const i32x4*>(smem + 32768 + 1024 * c + wbase); }
          if (gs == 16 * c + 14) MFMA16(dacc, da[c], dw[c]);
        }
      };
      i32x4 lo = {(int)f3[0], (int)f3[1], (int)f3[2], (int)f3[3]}, hi = {(int)f3[4], (int)f3[5], (int)f3[6], (int)f3[7]};
      swish_block<PK>(acc3, e, f3, [&](int s) {
        if (s == S1) MFMA32(acc4a, w4, lo);
        if (s == S2) MFMA32(acc4b, w4, hi);
        d_hook(s);
      });
      swish_block<PK>(acc4a, e, f4, [&](int s) {
        if (s == S1) MFMA32(acc3, w3a, b0);
        if (s == S2) MFMA32(acc3, w3b, b1);
        d_hook(56 + s);
      });
      swish_block<PK>(acc4b, e, f4, [&](int s) { d_hook(112 + s); });
      if (P == 6) {
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dacc[0]) : "v"(e[0]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dacc[1]) : "v"(e[0]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dacc[2]) : "v"(e[0]));
        asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(dacc[3]) : "v"(e[0]));
      }
    } else if (P == 5) {
      i32x4 da[5], dw[5];
#pragma unroll
      for (int half = 0; half < 2; ++half) {
#pragma unroll
        for (int c = 0; c < 5; ++c) { da[c] = *reinterpret_cast<const i32x4*>(smem + 4096 * (1 + half) + dbase + 4096 * c); dw[c] = *reinterpret_cast<const i32x4*>(smem + 32768 + 16384 * half + 1024 * c + wbase); }
#pragma unroll
        for (int c = 0; c < 5; ++c) dacc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(__attribute__((ext_vector_type(8))) short, da[c]), __builtin_bit_cast(__attribute__((ext_vector_type(8))) short, dw[c]), dacc, 0, 0, 0);
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) dacc[i] = dacc[i] * 1.0001f + 0.5f;
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  uint32_t sink = f3[0] ^ f3[5] ^ f4[0] ^ f4[1] ^ f4[2] ^ f4[3] ^ f4[4] ^ f4[5] ^ f4[6] ^ f4[7] ^ __builtin_bit_cast(uint32_t, acc3[3]) ^ __builtin_bit_cast(uint32_t, acc4a[5]) ^ __builtin_bit_cast(uint32_t, acc4b[7]) ^ __builtin_bit_cast(uint32_t, dacc[0] + dacc[3]);
  out[blockIdx.x * 64 * NW + tid] = sink;
  if (lane == 0) { cyc[2 * (blockIdx.x * NW + (tid >> 6))] = t1 - t0; cyc[2 * (blockIdx.x * NW + (tid >> 6)) + 1] = r1 - r0; }
}

template <int P, bool PK, int S1, int S2, int NW>
static int run(uint32_t* out, unsigned long long* cyc, const char* name) {
  const int iters = 4000, lds = 100 * 1024;
  auto fn = k<P, PK, S1, S2, NW>;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL(fn, dim3(256), dim3(64 * NW), lds, 0, out, cyc, 200);
  CK(hipDeviceSynchronize());
  float best = 1e30f;
  for (int rep = 0; rep < 3; ++rep) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fn, dim3(256), dim3(64 * NW), lds, 0, out, cyc, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    best = ms < best ? ms : best;
  }
  static unsigned long long h[2 * 256 * 16];
  CK(hipMemcpy(h, cyc, sizeof(unsigned long long) * 2 * 256 * NW, hipMemcpyDeviceToHost));
  double sum = 0, rsum = 0; for (int i = 0; i < 256 * NW; ++i) { sum += (double)h[2 * i]; rsum += (double)h[2 * i + 1]; }
  const double ticks = sum / (256.0 * NW) / iters;            // s_memtime ticks per iteration of one wave
  const double ns_wave = rsum / (256.0 * NW) / iters * 10.0;  // s_memrealtime: 100 MHz
  const double ns_wall = best * 1e6 / iters;                   // event time per iteration
  // NW/4 waves share a SIMD and each runs one item per iteration: SIMD time per item = iteration time / (NW/4)
  printf("%-74s %d w/SIMD  %7.3f ms  per item and SIMD: %7.2f ns wall (%7.2f ns in-kernel) = %7.1f cycles at 2.4 GHz;  s_memtime %7.1f ticks/iteration (%.2f per ns)\n",
         name, NW / 4, best, ns_wall / (NW / 4), ns_wave / (NW / 4), ns_wall / (NW / 4) * 2.4, ticks, ticks / ns_wave);
  return 0;
}

#define ROW(P, PK, S1, S2, name) \
  if (run<P, PK, S1, S2, 4>(out, cyc, name)) return 1; \
  if (run<P, PK, S1, S2, 8>(out, cyc, name)) return 1; \
  if (run<P, PK, S1, S2, 16>(out, cyc, name)) return 1;

int main() {
  uint32_t* out; unsigned long long* cyc;
  CK(hipMalloc(&out, 4 * 256 * 1024)); CK(hipMalloc(&cyc, 2 * 8 * 256 * 16));
  printf("BC-like item = 4 x v_mfma_f32_32x32x16_bf16 + 48 wave-registers of swish (96 transcendentals); 'per item and SIMD' = SIMD time one item costs.\n"
         "Wall time is what counts: the chip lowers its clock under these loads and s_memtime does not tick at one rate across the rows.\n");
  ROW(0, true, 0, 0, "P0 4 MFMA 32x32x16 bf16 only");
  ROW(1, true, 0, 0, "P1 swish only, packed add/mul");
  ROW(1, false, 0, 0, "P1 swish only, plain add/mul");
  ROW(2, true, 0, 0, "P2 r2 order: MFMA -> wait -> swish block, packed");
  ROW(2, false, 0, 0, "P2 r2 order: MFMA -> wait -> swish block, plain");
  ROW(3, true, 3, 11, "P3 MFMAs inside the swish stream (after exp 3, exp 11), packed");
  ROW(3, false, 3, 11, "P3 MFMAs inside the swish stream (after exp 3, exp 11), plain");
  ROW(3, true, 3, 4, "P3 MFMAs back to back after exp 3, 4, packed");
  ROW(3, true, 27, 35, "P3 MFMAs among the rcp (slots 27, 35), packed");
  ROW(3, false, 35, 43, "P3 MFMAs among the rcp (slots 35, 43), plain");
  ROW(5, true, 0, 0, "P5 D-like item alone: 20 ds_read_b128 + 10 MFMA 16x16x32 + epilogue");
  ROW(6, true, 3, 11, "P6 BC-like (P3 placement) + D-like MFMAs spread through it, packed");
  ROW(6, false, 3, 11, "P6 BC-like (P3 placement) + D-like MFMAs spread through it, plain");
  return 0;
}
