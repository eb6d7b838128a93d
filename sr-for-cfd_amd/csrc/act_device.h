// Device-side activation helpers shared by the f32 GEMM epilogues (kernels_fp32.hip) and the training kernels (train.hip),
// so that a fused epilogue and the stand-alone element-wise kernel it replaces produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace srcfd {

// sigmoid(z) from the hardware exp2 and reciprocal (<= 2 ulp each) with one Newton step on the reciprocal: ~8 instructions
// against ~40 for libm expf + an IEEE divide, a few 1e-7 relative apart (kernels_fp32.hip, act_apply_precise).
__device__ __forceinline__ float sigmoid_fast(float z) {
  const float den = 1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
  float r = __builtin_amdgcn_rcpf(den);
  return fmaf(fmaf(-den, r, 1.0f), r, r);
}

__device__ __forceinline__ float swish_train(float v) { return v * sigmoid_fast(v); }

// swish'(z) = s + z s (1 - s), s = sigmoid(z)
__device__ __forceinline__ float swish_grad_train(float z) {
  const float s = sigmoid_fast(z);
  return s + z * s * (1.0f - s);
}

}  // namespace srcfd
