// Training step, forward pass of encoder_10's four layers as two launches (train_enc.hip).  Internal to the library.
#pragma once
#include <hip/hip_runtime.h>

#include "kernels.h"

namespace srcfd {

struct TrainEncParams {
  const float* x;                        // (n, 10, 10, 1)
  int n;
  const float* w0; const float* b0;      // conv2d: B[9][64], bias[64]          (the f32 engine's operands, gathered from the flat parameters)
  const float* w1; const float* b1;      // conv2d_1: B[576][128] (k = tap * 64 + ci), bias[128]
  const float* wd; const float* bd;      // dense: B[3200][128] (k = flattened NHWC index), bias[128]
  const float* wl; const float* bl;      // latent_vector: B[128][nl_pad], bias[nl]
  int nl, nl_pad;
  int swish[4];                          // per layer: swish (1) or linear (0)
  // pre-activation / activation of every layer, kept for the backward pass (y == z for a linear layer)
  float* z0; float* y0;                  // (n, 5, 5, 64)
  float* z1; float* y1;                  // (n, 3200) flattened NHWC
  float* z2; float* y2;                  // (n, 128)
  float* z3; float* y3;                  // (n, nl)
  float* partial;                        // workspace [50 slices][n][128]
};
// the four forward descriptors are encoder_10's (conv2d 3x3 s2 SAME 1 -> 64 on 10x10, conv2d_1 3x3 s1 SAME 64 -> 128 on 5x5, dense 3200 -> 128, dense 128 -> nl)
bool train_enc_qualifies(const GemmDesc& conv0, const GemmDesc& conv1, const GemmDesc& dense, const GemmDesc& latent);
hipError_t launch_train_enc(const TrainEncParams& q, hipStream_t s);

}  // namespace srcfd
