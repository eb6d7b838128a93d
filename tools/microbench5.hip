// Microbench v5: throughput of a BC-like item (LDS read -> 2 MFMA -> swish x16 -> 2 x (MFMA -> swish x16 -> LDS write)) per SIMD
// as a function of waves per SIMD.  Answers: how much would the tail gain from more resident waves?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../sr-for-cfd_amd/csrc/dev16.h"
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)
using namespace srcfd;

template <int V>  // 0 full; 1 biases resident in registers; 2 no swish; 3 no MFMA; 4 no LDS writes; 5 biases resident + operands resident (no LDS reads)
__global__ void __launch_bounds__(256) bc_like(uint32_t* out, int iters) {
  __shared__ uint4 lds[256 * 6];
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 256 * 6; i += 256) lds[i] = make_uint4(0x3c003c00u + i, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u);
  __syncthreads();
  uint4 w4 = lds[lane];
  uint32_t sink = 0;
  const f32x16 rb3 = load_bias16(reinterpret_cast<const char*>(lds) + (lane >> 5) * 64), rb4 = load_bias16(reinterpret_cast<const char*>(lds) + 128 + (lane >> 5) * 64);
  const uint4 r0 = lds[256 + tid], r1 = lds[512 + tid], rw0 = lds[768 + tid], rw1 = lds[1024 + tid];
  for (int it = 0; it < iters; ++it) {
    uint4 b0 = V == 5 ? r0 : lds[256 + tid], b1 = V == 5 ? r1 : lds[512 + tid];
    if (V == 5) { b0.x ^= sink; }
    f32x16 acc3 = (V == 1 || V == 5) ? rb3 : load_bias16(reinterpret_cast<const char*>(lds) + (lane >> 5) * 64);
    if (V != 3) {
      acc3 = mfma32<false>(V == 5 ? rw0 : lds[768 + tid], b0, acc3);
      acc3 = mfma32<false>(V == 5 ? rw1 : lds[1024 + tid], b1, acc3);
    } else { acc3[0] += __builtin_bit_cast(float, b0.x); }
    uint32_t f3[8];
    swish_pack16<false>(acc3, f3, V == 2);
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
      uint4 bf = make_uint4(f3[4 * tt], f3[4 * tt + 1], f3[4 * tt + 2], f3[4 * tt + 3]);
      const f32x16 bias4 = (V == 1 || V == 5) ? rb4 : load_bias16(reinterpret_cast<const char*>(lds) + 128 + (lane >> 5) * 64);
      f32x16 acc4 = bias4;
      if (V != 3) acc4 = mfma32<false>(w4, bf, bias4); else acc4[0] += __builtin_bit_cast(float, bf.x);
      uint32_t f4[8];
      swish_pack16<false>(acc4, f4, V == 2);
#pragma unroll
      for (int q = 0; q < 4; ++q) if (V != 4) reinterpret_cast<uint2*>(lds + 1280)[(tid * 4 + q) & 511] = make_uint2(f4[2 * q], f4[2 * q + 1]); else sink ^= f4[2 * q + 1];
      sink ^= f4[0];
    }
  }
  out[blockIdx.x * 256 + tid] = sink;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); int ncu = pr.multiProcessorCount;
  uint32_t* out; CK(hipMalloc(&out, 4 * 256 * ncu * 8));
  const int iters = 2000;
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  void (*fns[])(uint32_t*, int) = {bc_like<0>, bc_like<1>, bc_like<2>, bc_like<3>, bc_like<4>, bc_like<5>};
  const char* names[] = {"full", "biases in registers", "no swish", "no MFMA", "no LDS writes", "no LDS reads at all"};
  for (int v = 0; v < 6; ++v)
    for (int bpc : {1, 4, 8}) {
      for (int rep = 0; rep < 2; ++rep) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(fns[v], dim3(ncu * bpc), dim3(256), 0, 0, out, iters);
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
      }
      float ms; CK(hipEventElapsedTime(&ms, a, b));
      printf("%-22s waves/SIMD %d: %6.0f cycles per BC-like item per SIMD\n", names[v], bpc, ms * 1e-3 * 2.4e9 / (bpc * iters));
    }
  return 0;
}
