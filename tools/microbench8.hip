// Microbench v8: do f32-input MFMAs (v_mfma_f32_16x16x4_f32 / 32x32x2) and f32 vector instructions of OTHER waves on the same
// SIMD overlap?  512-thread workgroups, one per CU: waves w and w + 4 share a SIMD.  Event-timed, every CU busy.
//   mode 0: every wave issues NM MFMAs per iteration              -> cycles per MFMA per SIMD (expected 32 / 64)
//   mode 1: every wave issues NV vector instructions per iteration -> cycles per instruction per SIMD
//   mode 2: waves 0-3 MFMAs only, waves 4-7 vector only            -> if the pipes are separate: max(mode 0, mode 1) work times
//   mode 3: every wave issues both, interleaved in its own stream
// Vector instruction types: 0 v_fma_f32, 1 v_pk_fma_f32, 2 v_exp_f32, 3 bf16 MFMA stands in for the f32 one (control).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int MODE, int VT, int MT, int NW = 8>   // MT: 0 f32 16x16x4, 1 f32 32x32x2, 2 bf16 16x16x32; NW waves per workgroup
__global__ void __launch_bounds__(64 * NW) k(float* out, int iters) {
  const int wave = threadIdx.x >> 6;
  const bool do_m = MODE == 0 || MODE == 3 || (MODE == 2 && wave < NW / 2);
  const bool do_v = MODE == 1 || MODE == 3 || (MODE == 2 && wave >= NW / 2);
  f32x4 acc[4]; f32x16 big[2];
  float v[16]; f32x2 pk[8];
  const float a = 1.0f + 1e-6f * threadIdx.x, b = 0.5f;
  const f32x2 c2 = {1.0001f, 0.9999f};
  s16x8 ab = {1, 2, 3, 4, 5, 6, 7, 8};
#pragma unroll
  for (int i = 0; i < 4; ++i) acc[i] = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int i = 0; i < 2; ++i) for (int j = 0; j < 16; ++j) big[i][j] = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) { pk[i].x = v[2 * i]; pk[i].y = v[2 * i + 1]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if (do_m) {
        if (MT == 0) {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
        } else if (MT == 1) {
#pragma unroll
          for (int i = 0; i < 2; ++i) big[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, big[i], 0, 0, 0);
        } else {
#pragma unroll
          for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, ab, acc[i], 0, 0, 0);
        }
      }
      if (do_v) {
        if (VT == 0) {
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(b));
        } else if (VT == 1) {
#pragma unroll
          for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[i]) : "v"(c2));
        } else {
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
        }
      }
    }
  }
  float s = 0;
  for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][3];
  for (int i = 0; i < 2; ++i) s += big[i][0] + big[i][7];
  for (int i = 0; i < 16; ++i) s += v[i];
  for (int i = 0; i < 8; ++i) s += pk[i].x + pk[i].y;
  if (s == 123.456f) out[threadIdx.x] = s;
}

template <int MODE, int VT, int MT, int NW = 8>
static int run(float* d, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  hipLaunchKernelGGL((k<MODE, VT, MT, NW>), dim3(256), dim3(64 * NW), 0, 0, d, 10);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL((k<MODE, VT, MT, NW>), dim3(256), dim3(64 * NW), 0, 0, d, iters);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  // per SIMD and iteration: MFMA waves x 16 (or 8) MFMAs, vector waves x 64 (or 32) instructions
  const double cyc = ms * 1e-3 * 2.4e9 / iters;
  printf("%-58s %8.3f ms  %9.1f SIMD-cycles per iteration (at 2.4 GHz)\n", name, ms, cyc);
  return 0;
}

int main() {
  float* d; CK(hipMalloc(&d, 4096));
  printf("per iteration and wave: 16 x v_mfma_f32_16x16x4_f32 (or 8 x 32x32x2, or 16 x bf16 16x16x32) and/or 64 vector instructions (32 packed)\n");
  run<0, 0, 0>(d, "MFMA f32 16x16x4 only, 2 waves/SIMD (32 MFMA/SIMD/iter)");
  run<0, 0, 1>(d, "MFMA f32 32x32x2 only, 2 waves/SIMD (16 MFMA/SIMD/iter)");
  run<0, 0, 2>(d, "MFMA bf16 16x16x32 only, 2 waves/SIMD (32 MFMA/SIMD/iter)");
  run<1, 0, 0>(d, "v_fma_f32 only, 2 waves/SIMD (128 instr/SIMD/iter)");
  run<1, 1, 0>(d, "v_pk_fma_f32 only, 2 waves/SIMD (64 instr/SIMD/iter)");
  run<1, 2, 0>(d, "v_exp_f32 only, 2 waves/SIMD (128 instr/SIMD/iter)");
  run<2, 0, 0>(d, "wave A: 16 MFMA f32 16x16x4 | wave B: 64 v_fma_f32");
  run<2, 1, 0>(d, "wave A: 16 MFMA f32 16x16x4 | wave B: 32 v_pk_fma_f32");
  run<2, 2, 0>(d, "wave A: 16 MFMA f32 16x16x4 | wave B: 64 v_exp_f32");
  run<2, 0, 1>(d, "wave A: 8 MFMA f32 32x32x2  | wave B: 64 v_fma_f32");
  run<2, 0, 2>(d, "wave A: 16 MFMA bf16 16x16x32 | wave B: 64 v_fma_f32");
  run<2, 2, 2>(d, "wave A: 16 MFMA bf16 16x16x32 | wave B: 64 v_exp_f32");
  run<3, 0, 0>(d, "every wave: 16 MFMA f32 16x16x4 + 64 v_fma_f32 interleaved");
  run<3, 2, 0>(d, "every wave: 16 MFMA f32 16x16x4 + 64 v_exp_f32 interleaved");
  run<3, 0, 2>(d, "every wave: 16 MFMA bf16 16x16x32 + 64 v_fma_f32 interleaved");
  printf("-- 4 waves per SIMD (1024-thread workgroups); 'A | B' = two MFMA-only waves beside two vector-only waves per SIMD --\n");
  run<0, 0, 0, 4>(d, "ONE wave/SIMD: MFMA f32 16x16x4 only (16 MFMA/SIMD/iter)");
  run<1, 0, 0, 4>(d, "ONE wave/SIMD: v_fma_f32 only (64 instr/SIMD/iter)");
  run<0, 0, 0, 16>(d, "4 waves/SIMD: MFMA f32 16x16x4 only (64 MFMA/SIMD/iter)");
  run<1, 0, 0, 16>(d, "4 waves/SIMD: v_fma_f32 only (256 instr/SIMD/iter)");
  run<1, 2, 0, 16>(d, "4 waves/SIMD: v_exp_f32 only (256 instr/SIMD/iter)");
  run<2, 0, 0, 16>(d, "2 waves x 16 MFMA f32 16x16x4 | 2 waves x 64 v_fma_f32");
  run<2, 1, 0, 16>(d, "2 waves x 16 MFMA f32 16x16x4 | 2 waves x 32 v_pk_fma_f32");
  run<2, 2, 0, 16>(d, "2 waves x 16 MFMA f32 16x16x4 | 2 waves x 64 v_exp_f32");
  run<2, 0, 1, 16>(d, "2 waves x 8 MFMA f32 32x32x2 | 2 waves x 64 v_fma_f32");
  run<2, 0, 2, 16>(d, "2 waves x 16 MFMA bf16 16x16x32 | 2 waves x 64 v_fma_f32");
  run<2, 2, 2, 16>(d, "2 waves x 16 MFMA bf16 16x16x32 | 2 waves x 64 v_exp_f32");
  run<3, 0, 0, 16>(d, "4 waves, each 16 MFMA f32 16x16x4 + 64 v_fma_f32 interleaved");
  run<3, 2, 0, 16>(d, "4 waves, each 16 MFMA f32 16x16x4 + 64 v_exp_f32 interleaved");
  return 0;
}
