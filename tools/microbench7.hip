// Microbench v7: are the 16-bit transcendentals any cheaper than the f32 ones?  (They would have to be for a packed-f16
// swish to beat the f32 sequence of microbench6.)  Same method: event-timed, every CU busy, 16 independent registers per
// wave, 4 and 8 waves per SIMD, cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

template <int V>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  unsigned v[16], e[16];
  const unsigned c = 0x3c003c01u;  // (1.0, 1.001) as packed f16
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = 0x3c003c00u + threadIdx.x + i; e[i] = v[i]; }
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(v[i]));
    } else if (V == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f16 %0, %0" : "+v"(v[i]));
    } else if (V == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    } else if (V == 3) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    } else if (V == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_fma_f16 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
    } else if (V == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pkrtz_f16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    } else if (V == 6) {  // packed-f16 swish of 32 activations held as 16 pairs: exp lo/hi, pk_add, rcp lo/hi, pk_mul (96 instructions)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(e[i]) : "v"(v[i]));
        asm volatile("v_exp_f16_sdwa %0, -%1 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(e[i]) : "v"(v[i]));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_add_f16 %0, %0, %1" : "+v"(e[i]) : "v"(c));
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        asm volatile("v_rcp_f16_sdwa %0, %0 dst_sel:WORD_0 dst_unused:UNUSED_PRESERVE src0_sel:WORD_0" : "+v"(e[i]));
        asm volatile("v_rcp_f16_sdwa %0, %0 dst_sel:WORD_1 dst_unused:UNUSED_PRESERVE src0_sel:WORD_1" : "+v"(e[i]));
      }
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(v[i]) : "v"(e[i]));
    }
  }
  unsigned s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + e[i];
  out[blockIdx.x * 256 + threadIdx.x] = (float)s;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); int ncu = pr.multiProcessorCount;
  float* out; CK(hipMalloc(&out, 4 * 256 * ncu * 8));
  const int iters = 4000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  void (*fns[])(float*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>};
  const char* names[] = {"v_exp_f16", "v_rcp_f16", "v_pk_add_f16", "v_pk_mul_f16", "v_pk_fma_f16", "v_cvt_pkrtz_f16_f32", "packed-f16 swish (6 instr per pair) per activation"};
  const int per_iter[] = {16, 16, 16, 16, 16, 16, 32};  // instruction slots (or activations) per iteration
  for (int v = 0; v < 7; ++v)
    for (int bpc : {4, 8}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(fns[v], dim3(ncu * bpc), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      }
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%-52s waves/SIMD %d: %6.2f cycles per wave-%s per SIMD\n", names[v], bpc, ms * 1e-3 * 2.4e9 / ((double)bpc * iters * per_iter[v]), v >= 6 ? "activation" : "instruction");
    }
  return 0;
}
