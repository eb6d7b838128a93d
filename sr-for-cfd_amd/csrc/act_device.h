// Device-side activation helpers shared by the f32 GEMM epilogues (kernels_fp32.hip) and the training kernels (train.hip),
// so that a fused epilogue and the stand-alone element-wise kernel it replaces produce the same bits.
#pragma once
#include <hip/hip_runtime.h>

namespace srcfd {

// sigmoid(z) from the hardware exp2 and reciprocal (<= 2 ulp each) with one Newton step on the reciprocal: ~8 instructions
// against ~40 for libm expf + an IEEE divide, a few 1e-7 relative apart (kernels_fp32.hip, act_apply_precise).
// For z < -88 exp2 overflows to inf and rcp(inf) = 0, which is the right sigmoid; but the Newton step's residual
// fma(-inf, 0, 1) is then NaN.  fminf(residual, 1) returns the non-NaN operand (the residual is ~1e-7 otherwise), so the
// refined value stays 0 and z * sigmoid(z) = -0.  A NaN input still gives NaN.
__device__ __forceinline__ float sigmoid_fast(float z) {
  const float den = 1.0f + __builtin_amdgcn_exp2f(z * -1.4426950408889634f);
  const float r = __builtin_amdgcn_rcpf(den);
  return fmaf(fminf(fmaf(-den, r, 1.0f), 1.0f), r, r);
}

__device__ __forceinline__ float swish_train(float v) { return v * sigmoid_fast(v); }

// swish'(z) = s + z s (1 - s), s = sigmoid(z)
__device__ __forceinline__ float swish_grad_train(float z) {
  const float s = sigmoid_fast(z);
  return s + z * s * (1.0f - s);
}

}  // namespace srcfd
