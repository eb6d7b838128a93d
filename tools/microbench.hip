// Microbenchmarks that size the fused bf16 tail (DESIGN.md "VALU ceiling"):
//  1. swish (exp2+rcp form) throughput per CU vs waves/SIMD
//  2. the same interleaved with bf16 MFMA 32x32x16 (1 MFMA per 16 swish)
//  3. MFMA-only issue rate
// Build: hipcc -O3 --offload-arch=gfx950 tools/microbench.hip -o gpurun_out/microbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

__device__ __forceinline__ float swish_fast(float u) {  // u = x*log2e
  float e = __builtin_amdgcn_exp2f(-u);
  return u * __builtin_amdgcn_rcpf(1.0f + e);
}
__device__ __forceinline__ float swish_std(float x) {
  float e = __builtin_amdgcn_exp2f(x * -1.4426950408889634f);
  return x * __builtin_amdgcn_rcpf(1.0f + e);
}

template <int MODE>
__global__ void __launch_bounds__(256) k_swish(float* out, int iters, float seed) {
  f32x16 v;
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = seed + 0.01f * r + 0.001f * threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = (MODE == 0 ? swish_fast(v[r]) : swish_std(v[r])) + 1.0f;
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += v[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// 1 MFMA (32x32x16 bf16) + 16 swish on its result per iteration
template <int NSW>
__global__ void __launch_bounds__(256) k_mfma_swish(float* out, int iters, float seed) {
  bf16x8 a, b;
#pragma unroll
  for (int j = 0; j < 8; ++j) { a[j] = (short)(0x3c00 + threadIdx.x + j); b[j] = (short)(0x3c10 + j); }
  f32x16 bias;
#pragma unroll
  for (int r = 0; r < 16; ++r) bias[r] = seed * r;
  f32x16 accum = bias;
  for (int it = 0; it < iters; ++it) {
    f32x16 d = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, bias, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < NSW; ++r) d[r] = swish_fast(d[r]);
#pragma unroll
    for (int r = 0; r < 16; ++r) accum[r] += d[r];
    a[0] = (short)(a[0] + 1);
  }
  float s = 0;
#pragma unroll
  for (int r = 0; r < 16; ++r) s += accum[r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename F> float time_ms(F f, int reps = 5) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  f();
  hipDeviceSynchronize();
  float best = 1e30f;
  for (int i = 0; i < reps; ++i) { hipEventRecord(a); f(); hipEventRecord(b); hipEventSynchronize(b); float t; hipEventElapsedTime(&t, a, b); if (t < best) best = t; }
  return best;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
  printf("device %s CUs %d clock %d kHz\n", p.name, p.multiProcessorCount, p.clockRate);
  int ncu = p.multiProcessorCount;
  float* out; CK(hipMalloc(&out, sizeof(float) * 256 * 8 * ncu * 4));
  const int iters = 4000;
  for (int wps = 1; wps <= 8; wps *= 2) {        // waves per SIMD: blocks of 256 thr = 1 wave/SIMD each
    int blocks = ncu * wps;
    for (int mode = 0; mode < 2; ++mode) {
      float t = time_ms([&] { if (mode == 0) hipLaunchKernelGGL(k_swish<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f); else hipLaunchKernelGGL(k_swish<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.5f); });
      double elems = (double)blocks * 256 * 16 * iters;
      // cycles per wave-register-swish per SIMD: time * clk / (swish wave-instr groups per SIMD)
      double per_simd = (double)wps * 16 * iters;  // swish-regs per SIMD
      printf("swish mode=%d waves/SIMD=%d: %.3f ms, %.2f Telem/s, %.2f cyc/swish-reg/SIMD (@2.4GHz)\n", mode, wps, t, elems / t * 1e-9, t * 1e-3 * 2.4e9 / per_simd);
    }
  }
  for (int wps = 1; wps <= 4; wps *= 2) {
    int blocks = ncu * wps;
    float t16 = time_ms([&] { hipLaunchKernelGGL(k_mfma_swish<16>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.01f); });
    float t8 = time_ms([&] { hipLaunchKernelGGL(k_mfma_swish<8>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.01f); });
    float t0 = time_ms([&] { hipLaunchKernelGGL(k_mfma_swish<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 0.01f); });
    double per_simd = (double)wps * iters;
    printf("mfma+swish waves/SIMD=%d: 16sw %.3f ms (%.1f cyc/iter/SIMD)  8sw %.3f ms (%.1f)  0sw %.3f ms (%.1f)\n", wps,
           t16, t16 * 1e-3 * 2.4e9 / per_simd, t8, t8 * 1e-3 * 2.4e9 / per_simd, t0, t0 * 1e-3 * 2.4e9 / per_simd);
  }
  return 0;
}
