// Microbench v6: SIMD-level VALU throughput (event-timed, every CU busy) of single instruction types and of the swish sequence,
// 16 independent registers per wave, at 4 and 8 waves per SIMD.  Cycles per wave-instruction per SIMD at 2.4 GHz.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int V>
__global__ void __launch_bounds__(256) k(float* out, int iters) {
  float v[16], e[16]; f32x2 p[8];
  const float c = 1.0001f; const f32x2 c2 = {1.0001f, 0.9999f};
#pragma unroll
  for (int i = 0; i < 16; ++i) { v[i] = 1.0f + 0.001f * (threadIdx.x + i); e[i] = v[i]; }
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i].x = v[2 * i]; p[i].y = v[2 * i + 1]; }
  for (int it = 0; it < iters; ++it) {
    if (V == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    } else if (V == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if (V == 2) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[i]));
    } else if (V == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
    } else if (V == 4) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    } else if (V == 5) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[i]) : "v"(c));
    } else if (V == 6) {  // swish, batches of 16 (counts as 64 instructions)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, -%1" : "=v"(e[i]) : "v"(v[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(e[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(e[i]));
    } else {              // swish with packed add / mul (48 instructions)
#pragma unroll
      for (int i = 0; i < 8; ++i) { asm volatile("v_exp_f32 %0, -%1" : "=v"(e[2 * i]) : "v"(p[i].x)); asm volatile("v_exp_f32 %0, -%1" : "=v"(e[2 * i + 1]) : "v"(p[i].y)); }
      f32x2 q[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) { q[i].x = e[2 * i]; q[i].y = e[2 * i + 1]; asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(q[i]) : "v"(c2)); }
#pragma unroll
      for (int i = 0; i < 8; ++i) { asm volatile("v_rcp_f32 %0, %0" : "+v"(q[i].x)); asm volatile("v_rcp_f32 %0, %0" : "+v"(q[i].y)); }
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(q[i]));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + e[i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); int ncu = pr.multiProcessorCount;
  float* out; CK(hipMalloc(&out, 4 * 256 * ncu * 8));
  const int iters = 4000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  void (*fns[])(float*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>};
  const char* names[] = {"v_add_f32", "v_exp_f32", "v_rcp_f32", "v_pk_add/mul_f32", "v_cvt_pk_bf16_f32", "v_fma_f32", "swish (exp,add,rcp,mul) per activation", "swish with pk add/mul per activation"};
  const int per_iter[] = {16, 16, 16, 16, 16, 16, 16, 16};  // instruction slots (or activations) per iteration
  for (int v = 0; v < 8; ++v)
    for (int bpc : {4, 8}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(fns[v], dim3(ncu * bpc), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      }
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%-42s waves/SIMD %d: %6.2f cycles per wave-%s per SIMD\n", names[v], bpc, ms * 1e-3 * 2.4e9 / ((double)bpc * iters * per_iter[v]), v >= 6 ? "activation" : "instruction");
    }
  return 0;
}
