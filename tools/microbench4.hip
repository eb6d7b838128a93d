// Microbench v4: per-wave issue cadence of VALU / packed-f32 / MFMA instructions vs waves per SIMD (s_memtime).
// 16 independent chains per wave; block sizes 256..1024 threads give 1..4 waves per SIMD (one block per CU), two
// 1024-thread blocks per CU give 8.  Prints elapsed cycles per instruction PER WAVE (not divided by the wave count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define SWISH_B16 \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[0]) : "v"(v[0])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[1]) : "v"(v[1])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[2]) : "v"(v[2])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[3]) : "v"(v[3])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[4]) : "v"(v[4])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[5]) : "v"(v[5])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[6]) : "v"(v[6])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[7]) : "v"(v[7])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[8]) : "v"(v[8])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[9]) : "v"(v[9])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[10]) : "v"(v[10])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[11]) : "v"(v[11])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[12]) : "v"(v[12])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[13]) : "v"(v[13])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[14]) : "v"(v[14])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[15]) : "v"(v[15])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[0])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[1])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[2])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[3])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[4])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[5])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[6])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[7])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[8])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[9])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[10])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[11])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[12])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[13])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[14])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[15])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[0])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[1])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[2])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[3])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[4])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[5])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[6])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[7])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[8])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[9])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[10])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[11])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[12])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[13])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[14])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[15])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[0]) : "v"(e[0])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[1]) : "v"(e[1])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[2]) : "v"(e[2])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[3]) : "v"(e[3])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[4]) : "v"(e[4])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[5]) : "v"(e[5])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[6]) : "v"(e[6])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[7]) : "v"(e[7])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[8]) : "v"(e[8])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[9]) : "v"(e[9])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[10]) : "v"(e[10])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[11]) : "v"(e[11])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[12]) : "v"(e[12])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[13]) : "v"(e[13])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[14]) : "v"(e[14])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[15]) : "v"(e[15]));

#define SWISH_B8 \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[0]) : "v"(v[0])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[1]) : "v"(v[1])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[2]) : "v"(v[2])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[3]) : "v"(v[3])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[4]) : "v"(v[4])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[5]) : "v"(v[5])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[6]) : "v"(v[6])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[7]) : "v"(v[7])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[0])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[1])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[2])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[3])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[4])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[5])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[6])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[7])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[0])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[1])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[2])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[3])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[4])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[5])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[6])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[7])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[0]) : "v"(e[0])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[1]) : "v"(e[1])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[2]) : "v"(e[2])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[3]) : "v"(e[3])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[4]) : "v"(e[4])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[5]) : "v"(e[5])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[6]) : "v"(e[6])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[7]) : "v"(e[7])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[8]) : "v"(v[8])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[9]) : "v"(v[9])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[10]) : "v"(v[10])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[11]) : "v"(v[11])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[12]) : "v"(v[12])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[13]) : "v"(v[13])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[14]) : "v"(v[14])); \
  asm volatile("v_exp_f32 %0, -%1" : "=v"(e[15]) : "v"(v[15])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[8])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[9])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[10])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[11])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[12])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[13])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[14])); \
  asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[15])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[8])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[9])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[10])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[11])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[12])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[13])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[14])); \
  asm volatile("v_rcp_f32 %0, %0" : "+v"(e[15])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[8]) : "v"(e[8])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[9]) : "v"(e[9])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[10]) : "v"(e[10])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[11]) : "v"(e[11])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[12]) : "v"(e[12])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[13]) : "v"(e[13])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[14]) : "v"(e[14])); \
  asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[15]) : "v"(e[15]));

#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s16x8 __attribute__((ext_vector_type(8)));

template <int WHICH>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* stamps, int iters) {
  float v[16]; f32x2 p[8]; float c = 1.0001f; f32x2 c2 = {1.0001f, 0.9999f};
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
#pragma unroll
  for (int i = 0; i < 8; ++i) { p[i].x = v[2 * i]; p[i].y = v[2 * i + 1]; }
  f32x16 acc[2]; f32x4 acc4[4];
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }
  for (int i = 0; i < 4; ++i) acc4[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  s16x8 a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {1, 1, 1, 1, 1, 1, 1, 1};
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (WHICH == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
    } else if (WHICH == 1) {
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    } else if (WHICH == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
    } else if (WHICH == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i]) : "v"(c2));
    } else if (WHICH == 4) {  // 16 dependent v_add on ONE chain
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(c));
    } else if (WHICH == 5) {  // 16 dependent v_exp on ONE chain
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[0]));
    } else if (WHICH == 6) {  // the swish sequence of dev16.h on 16 values (per 16 activations: count as 16 "instructions")
      float e[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_exp_f32 %0, -%1" : "=v"(e[i]) : "v"(v[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_add_f32 %0, 1.0, %0" : "+v"(e[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_rcp_f32 %0, %0" : "+v"(e[i]));
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(e[i]));
    } else if (WHICH == 7) {  // 8 independent-ish MFMA 32x32x16 alternating two accumulators (count 16 per iter: 2 loops of 8)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[i & 1], 0, 0, 0);
    } else if (WHICH == 8) {  // 16 dependent MFMA 16x16x32 on one accumulator
#pragma unroll
      for (int i = 0; i < 16; ++i) acc4[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[0], 0, 0, 0);
    } else if (WHICH == 9) {  // 16 MFMA 16x16x32 over four accumulators
#pragma unroll
      for (int i = 0; i < 16; ++i) acc4[i & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc4[i & 3], 0, 0, 0);
    } else if (WHICH == 11) { float e[16]; SWISH_B16
    } else if (WHICH == 12) { float e[16]; SWISH_B8
    } else {                  // cvt_pk
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[i]) : "v"(c));
    }
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0;
  for (int i = 0; i < 16; ++i) s += v[i] + acc[0][i] + acc[1][i];
  for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
  for (int i = 0; i < 4; ++i) s += acc4[i][0];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0)); int ncu = pr.multiProcessorCount;
  float* out; unsigned long long* st; CK(hipMalloc(&out, sizeof(float) * 1024 * ncu * 2)); CK(hipMalloc(&st, 8 * 16 * ncu * 2));
  const int iters = 1000; std::vector<unsigned long long> h(16 * ncu * 2);
  const char* names[] = {"v_exp_f32 x16 indep", "v_add_f32 x16 indep", "v_pk_add_f32 x16 (8 regs x2)", "v_pk_mul_f32 x16 (8 regs x2)", "v_add_f32 x16 dependent",
                         "v_exp_f32 x16 dependent", "swish x16 (exp,add,rcp,mul batches = 64 instr)", "mfma 32x32x16 bf16 x16 (2 acc)", "mfma 16x16x32 bf16 x16 dependent",
                         "mfma 16x16x32 bf16 x16 (4 acc)", "v_cvt_pk_bf16_f32 x16 indep", "swish16 order B16", "swish16 order B8"};
  void (*fns[])(float*, unsigned long long*, int) = {k<0>, k<1>, k<2>, k<3>, k<4>, k<5>, k<6>, k<7>, k<8>, k<9>, k<10>, k<11>, k<12>};
  const int cfg[][2] = {{256, 1}, {512, 1}, {1024, 1}, {1024, 2}};  // threads per block, blocks per CU
  for (int w = 0; w < 13; ++w) {
    printf("%-48s", names[w]);
    for (auto& c : cfg) {
      int blocks = ncu * c[1];
      for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(fns[w], dim3(blocks), dim3(c[0]), 0, 0, out, st, iters); CK(hipDeviceSynchronize()); }
      int nw = blocks * c[0] / 64;
      CK(hipMemcpy(h.data(), st, 8 * nw, hipMemcpyDeviceToHost));
      std::vector<double> v(h.begin(), h.begin() + nw); std::sort(v.begin(), v.end());
      printf("  %dw/SIMD: %6.2f", c[0] / 256 * c[1], v[v.size() / 2] / iters / 16.0);
    }
    printf("   (cycles per instruction slot, per wave)\n");
  }
  return 0;
}
