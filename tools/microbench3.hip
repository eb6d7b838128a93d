// Microbench v3: issue cost of single VALU instructions at 4 waves/SIMD (s_memtime), 16 independent chains.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#define CK(x) do{hipError_t e=(x); if(e!=hipSuccess){printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1;}}while(0)

#define BODY16(INS) \
  asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) INS(8) INS(9) INS(10) INS(11) INS(12) INS(13) INS(14) INS(15) \
    : "+v"(v[0]),"+v"(v[1]),"+v"(v[2]),"+v"(v[3]),"+v"(v[4]),"+v"(v[5]),"+v"(v[6]),"+v"(v[7]),"+v"(v[8]),"+v"(v[9]),"+v"(v[10]),"+v"(v[11]),"+v"(v[12]),"+v"(v[13]),"+v"(v[14]),"+v"(v[15]) : "v"(c))

#define I_EXP32(i) "v_exp_f32 %" #i ", %" #i "\n"
#define I_RCP32(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define I_EXP16(i) "v_exp_f16 %" #i ", %" #i "\n"
#define I_RCP16(i) "v_rcp_f16 %" #i ", %" #i "\n"
#define I_ADD32(i) "v_add_f32 %" #i ", %" #i ", %16\n"
#define I_MUL32(i) "v_mul_f32 %" #i ", %" #i ", %16\n"
#define I_FMA32(i) "v_fma_f32 %" #i ", %" #i ", %16, %16\n"
#define I_PKMUL16(i) "v_pk_mul_f16 %" #i ", %" #i ", %16\n"
#define I_PKFMA16(i) "v_pk_fma_f16 %" #i ", %" #i ", %16, %16\n"
#define I_PKADD32(i) "v_pk_add_f32 %" #i ", %" #i ", %16\n"
#define I_CVTPK(i) "v_cvt_pk_bf16_f32 %" #i ", %" #i ", %16\n"
#define I_SQRT(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define I_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define I_MOV(i) "v_mov_b32 %" #i ", %16\n"

template <int WHICH>
__global__ void __launch_bounds__(1024) k(float* out, unsigned long long* stamps, int iters) {
  float v[16]; float c = 1.0001f;
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = 1.0f + 0.001f * (threadIdx.x + i);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (WHICH == 0) BODY16(I_EXP32); else if (WHICH == 1) BODY16(I_RCP32); else if (WHICH == 2) BODY16(I_EXP16); else if (WHICH == 3) BODY16(I_RCP16);
    else if (WHICH == 4) BODY16(I_ADD32); else if (WHICH == 5) BODY16(I_MUL32); else if (WHICH == 6) BODY16(I_FMA32); else if (WHICH == 7) BODY16(I_PKMUL16);
    else if (WHICH == 8) BODY16(I_PKFMA16); else if (WHICH == 9) BODY16(I_CVTPK); else if (WHICH == 10) BODY16(I_SQRT); else if (WHICH == 11) BODY16(I_LOG); else BODY16(I_MOV);
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0; for (int i = 0; i < 16; ++i) s += v[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) stamps[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

int main() {
  hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0)); int ncu = p.multiProcessorCount;
  float* out; unsigned long long* st; CK(hipMalloc(&out, sizeof(float) * 1024 * ncu * 2)); CK(hipMalloc(&st, 8 * 16 * ncu * 2));
  const int iters = 2000; std::vector<unsigned long long> h(16 * ncu * 2);
  const char* names[] = {"v_exp_f32","v_rcp_f32","v_exp_f16","v_rcp_f16","v_add_f32","v_mul_f32","v_fma_f32","v_pk_mul_f16","v_pk_fma_f16","v_cvt_pk_bf16_f32","v_sqrt_f32","v_log_f32","v_mov_b32"};
  for (int blocks_per_cu = 1; blocks_per_cu <= 2; ++blocks_per_cu)
  for (int w = 0; w < 13; ++w) {
    int blocks = ncu * blocks_per_cu;
    void (*fn)(float*, unsigned long long*, int) = nullptr;
    switch (w) { case 0: fn = k<0>; break; case 1: fn = k<1>; break; case 2: fn = k<2>; break; case 3: fn = k<3>; break; case 4: fn = k<4>; break; case 5: fn = k<5>; break;
      case 6: fn = k<6>; break; case 7: fn = k<7>; break; case 8: fn = k<8>; break; case 9: fn = k<9>; break; case 10: fn = k<10>; break; case 11: fn = k<11>; break; default: fn = k<12>; }
    for (int rep = 0; rep < 2; ++rep) { hipLaunchKernelGGL(fn, dim3(blocks), dim3(1024), 0, 0, out, st, iters); CK(hipDeviceSynchronize()); }
    CK(hipMemcpy(h.data(), st, 8 * 16 * blocks, hipMemcpyDeviceToHost));
    std::vector<double> c(h.begin(), h.begin() + 16 * blocks); std::sort(c.begin(), c.end());
    double wave_cyc = c[c.size() / 2];
    int wps = 4 * blocks_per_cu;
    printf("%-20s waves/SIMD=%d: %.2f SIMD cycles per wave-instruction\n", names[w], wps, wave_cyc / iters / 16.0 / wps);
  }
  return 0;
}
