// How exactly does v_mfma_f32_32x32x16_bf16 accumulate?  (round 4: the split-bf16 GEMM, kernels_x3.hip, relies on the answer)
//   D = sum_k A[i][k] B[k][j] over K = 16 * STEPS with random bf16 operands, one wave, chained MFMAs (C in = D out);
//   reference in double on the host.  Three data sets: all products positive (a rounding BIAS shows as a drift of the
//   mean error), signed products, and "one big + many small" (are small addends lost against a big accumulator?).
//   mode 3 is the case that matters for kernels_x3.hip: the SAME accumulator first takes 64 MFMAs of O(1) products, then 64 MFMAs
//   of products 2^-8 smaller (a "hi x mid" plane).  Finding (MI355X, ROCm 7.2; profiles/r04/o_...): added to the big accumulator
//   the small plane's contribution is off by up to 2.5e-4 of ITS sum (each of an MFMA's 16 products is cut at the big accumulator's
//   last bit on the way into the adder); accumulated alone (C starts at 0) the same products are good to 6e-7.  In the split-bf16
//   GEMM that was 3e-5 on a layer whose arithmetic is good to 1.5e-7 -- hence one accumulator set per magnitude class there.
// build: hipcc --offload-arch=gfx950 -O2 tools/microbench11.hip -o tools/_build/microbench11
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <random>
#include <vector>
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
// A: [32][K] bf16 row-major, B: [K][32] stored as Bt[32][K]; lane (r = lane & 31, h = lane >> 5) holds A[r][16 s + 8 h + j], B[16 s + 8 h + j][r]
__global__ void k(const uint16_t* A, const uint16_t* Bt, float* D, int steps, int K) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc;
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int s = 0; s < steps; ++s) {
    s16x8 a = *reinterpret_cast<const s16x8*>(A + r * K + 16 * s + 8 * h);
    s16x8 b = *reinterpret_cast<const s16x8*>(Bt + r * K + 16 * s + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];   // row = (reg&3) + 8 (reg>>2) + 4 h, col = lane & 31
}
static uint16_t bf(float f) { uint32_t u; memcpy(&u, &f, 4); u = (u + 0x7fff + ((u >> 16) & 1)) >> 16; return (uint16_t)u; }
static float fb(uint16_t b) { uint32_t u = (uint32_t)b << 16; float f; memcpy(&f, &u, 4); return f; }
// mode 3: acc = sum_k A B  (steps MFMAs), then acc += sum_k A B2 with B2 = B * 2^-8 rounded to bf16 (steps more); separately S = sum_k A B2 from zero
__global__ void k2(const uint16_t* A, const uint16_t* Bt, const uint16_t* Bt2, float* D, float* S, int steps, int K) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 acc, small;
  for (int i = 0; i < 16; ++i) { acc[i] = 0.f; small[i] = 0.f; }
  for (int s = 0; s < steps; ++s) {
    s16x8 a = *reinterpret_cast<const s16x8*>(A + r * K + 16 * s + 8 * h);
    s16x8 b = *reinterpret_cast<const s16x8*>(Bt + r * K + 16 * s + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
  }
  for (int s = 0; s < steps; ++s) {
    s16x8 a = *reinterpret_cast<const s16x8*>(A + r * K + 16 * s + 8 * h);
    s16x8 b = *reinterpret_cast<const s16x8*>(Bt2 + r * K + 16 * s + 8 * h);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
    small = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, small, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = acc[i];
    S[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = small[i];
  }
}
int main() {
  const int steps = 64, K = 16 * steps;
  std::mt19937 g(1);
  std::uniform_real_distribution<float> U(0.5f, 1.5f), S(-1.f, 1.f);
  for (int mode = 0; mode < 3; ++mode) {
    std::vector<uint16_t> A(32 * K), Bt(32 * K);
    for (int i = 0; i < 32 * K; ++i) {
      float a = mode == 0 ? U(g) : S(g), b = mode == 0 ? U(g) : S(g);
      if (mode == 2) { a = (i % K) == 0 ? 1024.f : S(g) * 1e-3f; b = (i % K) == 0 ? 1024.f : S(g); }
      A[i] = bf(a); Bt[i] = bf(b);
    }
    uint16_t *dA, *dB; float* dD;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, Bt.size() * 2); hipMalloc(&dD, 32 * 32 * 4);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), Bt.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, steps, K);
    std::vector<float> D(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
    double maxrel = 0, meanrel = 0, f32rel = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double ref = 0, mag = 0; float f = 0.f;
        for (int kk = 0; kk < K; ++kk) { double p = (double)fb(A[i * K + kk]) * fb(Bt[j * K + kk]); ref += p; mag += std::fabs(p); f = fmaf(fb(A[i * K + kk]), fb(Bt[j * K + kk]), f); }
        double e = (D[i * 32 + j] - ref) / mag;
        maxrel = std::fmax(maxrel, std::fabs(e)); meanrel += e / 1024; f32rel = std::fmax(f32rel, std::fabs((f - ref) / mag));
      }
    printf("mode %d (%s): K = %d  max |err| / sum|products| = %.3e  mean signed = %+.3e   (sequential f32 fma chain: max %.3e; 2^-24 = 5.96e-8)\n", mode,
           mode == 0 ? "positive products" : mode == 1 ? "signed products" : "one 2^20 product + small ones", K, maxrel, meanrel, f32rel);
  }
  {  // mode 3
    std::uniform_real_distribution<float> P(0.5f, 1.5f);
    std::vector<uint16_t> A(32 * K), Bt(32 * K), B2(32 * K);
    for (int i = 0; i < 32 * K; ++i) { A[i] = bf(P(g)); Bt[i] = bf(P(g)); B2[i] = bf(P(g) * (1.f / 256.f) * 1.37f); }
    uint16_t *dA, *dB, *dB2; float *dD, *dS;
    hipMalloc(&dA, A.size() * 2); hipMalloc(&dB, A.size() * 2); hipMalloc(&dB2, A.size() * 2); hipMalloc(&dD, 4096); hipMalloc(&dS, 4096);
    hipMemcpy(dA, A.data(), A.size() * 2, hipMemcpyHostToDevice); hipMemcpy(dB, Bt.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipMemcpy(dB2, B2.data(), A.size() * 2, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k2, dim3(1), dim3(64), 0, 0, dA, dB, dB2, dD, dS, steps, K);
    std::vector<float> D(1024), S(1024);
    hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost); hipMemcpy(S.data(), dS, 4096, hipMemcpyDeviceToHost);
    double m_joint = 0, m_sep = 0, mean_joint = 0;
    for (int i = 0; i < 32; ++i)
      for (int j = 0; j < 32; ++j) {
        double big = 0, sm = 0;
        for (int kk = 0; kk < K; ++kk) { big += (double)fb(A[i * K + kk]) * fb(Bt[j * K + kk]); sm += (double)fb(A[i * K + kk]) * fb(B2[j * K + kk]); }
        const double ej = (D[i * 32 + j] - (big + sm)) / sm, es = ((double)S[i * 32 + j] - sm) / sm;   // relative to the SMALL plane's sum
        m_joint = std::fmax(m_joint, std::fabs(ej)); mean_joint += ej / 1024; m_sep = std::fmax(m_sep, std::fabs(es));
      }
    printf("mode 3 (O(1) plane, then a plane 2^-8 smaller): error relative to the small plane's own sum: into the SAME accumulator max %.3e mean %+.3e; "
           "in an accumulator of its own max %.3e\n", m_joint, mean_joint, m_sep);
  }
  return 0;
}
