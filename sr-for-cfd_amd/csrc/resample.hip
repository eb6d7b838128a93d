// Separable float64 resampling of the super-resolved fields on the device (gfx950).
//
// Reference: `reshape_square_to_rectangular` (bfs_ml_accelerated.py:104-145) evaluates, per component,
// scipy's interpolating bicubic `RectBivariateSpline(y_sq, x_sq, field, kx=3, ky=3)(y_rect, x_rect)`.
// That operator is linear in the data and a tensor product, so with the two 1-D spline
// interpolation matrices Ry [OH][H] and Rx [OW][W] (built once on the host, float64)
//     out = Ry * field * Rx^T
// exactly (to rounding).  Two float64 GEMMs per component replace a FITPACK fit + evaluation of a
// 400x400 field on the host, which is ~97 % of the reference BFS call once the network runs on the GPU.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "engine.h"

namespace srcfd {

#define HIPCHECK(expr)                                                               \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));           \
      return SRCFD_EHIP;                                                             \
    }                                                                                \
  } while (0)

// C[z] (M x N) = A[z] (M x K) * B[z] (K x N), row-major, float64 accumulation on the matrix cores.
// One wave per 32 x 32 tile of C (2 x 2 v_mfma_f64_16x16x4_f64 tiles), operands straight from global memory (they are
// L2-resident: <= 1.3 MB per matrix at 400 x 400), the next 16-deep k chunk prefetched into registers while the current
// one is multiplied.  Lane (r = lane % 16, q = lane / 16) holds A[row r][k0 + 4q + j] and B[k0 + 4q + j][col r], j = 0..3;
// MFMA step j contracts the four k = k0 + 4q + j, so every k is covered once (summation order: by j, then q; fixed).
// 400^3 per component (the BFS call): 507 waves; measured 88 us -> see DESIGN.md for the VALU tile kernel it replaces.
typedef double f64x4 __attribute__((ext_vector_type(4)));

// Edge handling without branches: the address is clamped into the matrix and the loaded bits are ANDed with a lane mask, so
// the sixteen loads of a chunk issue back to back behind one wait.  (`ok ? load : 0`, and `ok ? v : 0` after an
// unconditional load alike, compile to a branch + s_waitcnt per load: 2.7 us per chunk.  The mask goes through an empty
// asm so that the AND is not folded back into that select.)
__device__ __forceinline__ double masked(double v, bool ok) {
  unsigned long long m = ok ? ~0ull : 0ull;
  asm("" : "+v"(m));
  return __builtin_bit_cast(double, __builtin_bit_cast(unsigned long long, v) & m);
}
__device__ __forceinline__ double masked(float v, bool ok) {
  unsigned m = ok ? ~0u : 0u;
  asm("" : "+v"(m));
  return (double)__builtin_bit_cast(float, __builtin_bit_cast(unsigned, v) & m);
}

// raw (unmasked) operand fragments of one 16-deep k chunk: 4 consecutive k per lane
template <typename T>
__device__ __forceinline__ void load_a_raw(const T* __restrict__ A, int row, int M, int K, int k, T (&a)[4]) {
  const T* p = A + (int64_t)min(row, M - 1) * K;
#pragma unroll
  for (int j = 0; j < 4; ++j) a[j] = p[min(k + j, K - 1)];
}
template <typename T>
__device__ __forceinline__ void load_b_raw(const T* __restrict__ B, int col, int N, int K, int k, T (&b)[4]) {
  const T* p = B + min(col, N - 1);
#pragma unroll
  for (int j = 0; j < 4; ++j) b[j] = p[(int64_t)min(k + j, K - 1) * N];
}
template <typename T>
__device__ __forceinline__ void mask_frag(const T (&raw)[4], bool in_range, int k, int K, double (&out)[4]) {
#pragma unroll
  for (int j = 0; j < 4; ++j) out[j] = masked(raw[j], in_range && k + j < K);
}

template <typename TA, typename TB>
__global__ void __launch_bounds__(64) gemm_f64(const TA* __restrict__ A, int64_t strideA, const TB* __restrict__ B, int64_t strideB,
                                                double* __restrict__ C, int64_t strideC, int M, int N, int K) {
  const int lane = threadIdx.x, r = lane & 15, q = lane >> 4;
  const int m0 = blockIdx.y * 32, n0 = blockIdx.x * 32;
  A += (int64_t)blockIdx.z * strideA;
  B += (int64_t)blockIdx.z * strideB;
  C += (int64_t)blockIdx.z * strideC;
  f64x4 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f64x4{0.0, 0.0, 0.0, 0.0};
  TA ar[2][4];
  TB br[2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    load_a_raw(A, m0 + 16 * t + r, M, K, 4 * q, ar[t]);
    load_b_raw(B, n0 + 16 * t + r, N, K, 4 * q, br[t]);
  }
  for (int k0 = 0; k0 < K; k0 += 16) {
    double a[2][4], b[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      mask_frag(ar[t], m0 + 16 * t + r < M, k0 + 4 * q, K, a[t]);
      mask_frag(br[t], n0 + 16 * t + r < N, k0 + 4 * q, K, b[t]);
    }
    // next chunk, unconditionally (past the end the clamped loads are masked to zero and never used): one basic block,
    // so the loads stay in flight under the sixteen MFMAs below
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      load_a_raw(A, m0 + 16 * t + r, M, K, k0 + 16 + 4 * q, ar[t]);
      load_b_raw(B, n0 + 16 * t + r, N, K, k0 + 16 + 4 * q, br[t]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) acc[mt][nt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[mt][j], b[nt][j], acc[mt][nt], 0, 0, 0);
  }
  // D layout of v_mfma_f64_16x16x4_f64: register v of lane (r, q) is C[4v + q][r] (rows interleave over the lane groups)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int v = 0; v < 4; ++v) {
        int m = m0 + 16 * mt + 4 * v + q, n = n0 + 16 * nt + r;
        if (m < M && n < N) C[(int64_t)m * N + n] = acc[mt][nt][v];
      }
}

__global__ void __launch_bounds__(256) widen_f64(const float* __restrict__ x, double* __restrict__ y, int64_t n) {
  int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i < n) y[i] = (double)x[i];
}

struct Resampler {
  int device = 0;
  int H = 0, W = 0, OH = 0, OW = 0;
  double* d_Ry = nullptr;   // [OH][H]
  double* d_RxT = nullptr;  // [W][OW]
  double* d_T = nullptr;    // [cap][H][OW]
  double* d_out = nullptr;  // [cap][OH][OW]
  int cap = 0;
  ~Resampler() {
    (void)hipSetDevice(device);
    for (void* p : {(void*)d_Ry, (void*)d_RxT, (void*)d_T, (void*)d_out}) if (p) (void)hipFree(p);
  }
  int reserve(int n) {
    if (n <= cap) return SRCFD_OK;
    for (void* p : {(void*)d_T, (void*)d_out}) if (p) HIPCHECK(hipFree(p));
    d_T = d_out = nullptr; cap = 0;
    HIPCHECK(hipMalloc(&d_T, (size_t)n * H * OW * sizeof(double)));
    HIPCHECK(hipMalloc(&d_out, (size_t)n * OH * OW * sizeof(double)));
    cap = n;
    return SRCFD_OK;
  }
  // out[z] = Ry * (in[z] * Rx^T) on `s`.  A factor that is the identity (the BFS case lx = max(lx, ly): the square and
  // the rectangle share their x nodes) is skipped: one GEMM instead of two.
  bool rx_identity = false, ry_identity = false;
  int run(const float* in_dev, int n, double* out_dev, hipStream_t s) {
    int rc = reserve(n);
    if (rc) return rc;
    if (rx_identity && ry_identity) {
      hipLaunchKernelGGL(widen_f64, dim3((unsigned)(((int64_t)n * H * W + 255) / 256)), dim3(256), 0, s, in_dev, out_dev, (int64_t)n * H * W);
    } else if (rx_identity) {
      hipLaunchKernelGGL((gemm_f64<double, float>), dim3((W + 31) / 32, (OH + 31) / 32, n), dim3(64), 0, s, d_Ry, (int64_t)0, in_dev, (int64_t)H * W, out_dev,
                         (int64_t)OH * W, OH, W, H);
    } else if (ry_identity) {
      hipLaunchKernelGGL((gemm_f64<float, double>), dim3((OW + 31) / 32, (H + 31) / 32, n), dim3(64), 0, s, in_dev, (int64_t)H * W, d_RxT, (int64_t)0, out_dev,
                         (int64_t)H * OW, H, OW, W);
    } else {
      hipLaunchKernelGGL((gemm_f64<float, double>), dim3((OW + 31) / 32, (H + 31) / 32, n), dim3(64), 0, s, in_dev, (int64_t)H * W, d_RxT, (int64_t)0, d_T,
                         (int64_t)H * OW, H, OW, W);
      hipLaunchKernelGGL((gemm_f64<double, double>), dim3((OW + 31) / 32, (OH + 31) / 32, n), dim3(64), 0, s, d_Ry, (int64_t)0, d_T, (int64_t)H * OW, out_dev,
                         (int64_t)OH * OW, OH, OW, H);
    }
    HIPCHECK(hipGetLastError());
    return SRCFD_OK;
  }
};

// ---------------------------------------------------------------------------
// Hand-off into the solver state (PyCFD_ML_accelerated.py:936-943, bfs_ml_accelerated.py:1211-1218):
//   Var[k, 1+i, 1+j] = field_k[j, i]                       (transposed injection, float64)
//   ghost cells from apply_bc_configured (PyCFD...:118-146): Dirichlet 2*value - inner, Neumann = inner,
//   left/right for j in 1..ny, then top/bottom for i in 1..nx; corners are never written (stay 0);
//   optional per-row Dirichlet profile on the left boundary (BFS inlet/wall mix, bfs...:524-562).
// One thread per Var element; same f64 expressions as the reference, so the result is bit-identical.
// ---------------------------------------------------------------------------
struct BcDev {
  int type[3][4];
  double value[3][4];
  int has_profile[3];
};

template <typename T>
__global__ void __launch_bounds__(256) solver_state_f64(const T* __restrict__ fields, int ny, int nx, BcDev bc, const double* __restrict__ left_profile,
                                                         double* __restrict__ Var) {
  const int64_t per = (int64_t)(nx + 2) * (ny + 2);
  const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (idx >= 3 * per) return;
  const int k = (int)(idx / per);
  const int r = (int)(idx - (int64_t)k * per);
  const int i = r / (ny + 2), j = r - i * (ny + 2);
  const T* f = fields + (int64_t)k * ny * nx;
  auto inner = [&](int ii, int jj) { return (double)f[(int64_t)(jj - 1) * nx + (ii - 1)]; };
  const bool ib = i == 0 || i == nx + 1, jb = j == 0 || j == ny + 1;
  double v = 0.0;
  if (!ib && !jb) {
    v = inner(i, j);
  } else if (ib && !jb) {  // left / right
    const int side = i == 0 ? 0 : 1;
    const double in = inner(i == 0 ? 1 : nx, j);
    if (side == 0 && bc.has_profile[k]) v = 2.0 * left_profile[(int64_t)k * ny + (j - 1)] - in;
    else v = bc.type[k][side] == 0 ? 2 * bc.value[k][side] - in : in;
  } else if (jb && !ib) {  // top / bottom
    const int side = j == ny + 1 ? 2 : 3;
    const double in = inner(i, j == 0 ? 1 : ny);
    v = bc.type[k][side] == 0 ? 2 * bc.value[k][side] - in : in;
  }
  Var[idx] = v;
}

}  // namespace srcfd

using srcfd::Resampler;
using srcfd::set_error;

namespace srcfd {
// ---------------------------------------------------------------------------
// Input preparation of the BFS call on the device: for every sample (one component of one
// coarse field, float64 (h, w)):  [optional] aspect-ratio resampling  X = Ry * F * Rx^T  (bfs_ml_accelerated.py:59-101),
// the float32 cast, and the adaptive-normalisation blend of bfs_ml_accelerated.py:1091-1097
//     input_mean = np.mean(X32), input_std = np.std(X32)
//     mean = (1 - b) * mean_train + b * input_mean;   std = (1 - b) * std_train + b * max(input_std, 1e-8)
// One workgroup per sample.  The statistics follow numpy's arithmetic step by step -- float32 pairwise sums with its
// eight-accumulator block for n <= 128, float32 division and sqrt, and NumPy-2 scalar promotion of the blend (a Python
// float next to a float32 scalar is rounded to float32 first) -- so that the (mean, std) pair equals the host recipe's.
// ---------------------------------------------------------------------------
__device__ float np_pairwise_block_f32(const float* a, int n) {   // numpy/_core/src/umath/loops_utils.h.src, pairwise_sum: the n <= 128 block
  if (n < 8) {
    float res = 0.f;
    for (int i = 0; i < n; ++i) res = __fadd_rn(res, a[i]);
    return res;
  }
  float r[8];
  for (int j = 0; j < 8; ++j) r[j] = a[j];
  int i = 8;
  for (; i < n - (n % 8); i += 8)
    for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], a[i + j]);
  float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])), __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
  for (; i < n; ++i) res = __fadd_rn(res, a[i]);
  return res;
}
// ... and its recursion above 128 elements (n2 = n / 2, rounded down to a multiple of 8; left sum + right sum): sides up to
// PREP_MAX = 32 give n <= 1024 = 128 * 8, three levels
template <int DEPTH>
__device__ float np_pairwise_sum_f32_rec(const float* a, int n) {
  if (n <= 128) return np_pairwise_block_f32(a, n);
  int n2 = n / 2;
  n2 -= n2 % 8;
  return __fadd_rn(np_pairwise_sum_f32_rec<DEPTH - 1>(a, n2), np_pairwise_sum_f32_rec<DEPTH - 1>(a + n2, n - n2));
}
template <>
__device__ float np_pairwise_sum_f32_rec<0>(const float* a, int n) { return np_pairwise_block_f32(a, n); }
__device__ float np_pairwise_sum_f32(const float* a, int n) { return np_pairwise_sum_f32_rec<4>(a, n); }   // n <= 2048

constexpr int PREP_MAX = 32;   // source and target sides up to 32 (the path's coarse fields are 10 x 10)

__global__ void __launch_bounds__(128) prepare_inputs_f64(const double* __restrict__ fields, int h, int w, const double* __restrict__ Ry,
                                                           const double* __restrict__ Rx, int lr, const double* __restrict__ train_stats,
                                                           int adaptive, double blend, float* __restrict__ x_out, float* __restrict__ aff_out) {
  __shared__ double F[PREP_MAX * PREP_MAX], T[PREP_MAX * PREP_MAX];
  __shared__ float X[PREP_MAX * PREP_MAX], D[PREP_MAX * PREP_MAX];
  const int s = blockIdx.x, tid = threadIdx.x;
  const double* src = fields + (size_t)s * h * w;
  for (int i = tid; i < h * w; i += 128) F[i] = src[i];
  __syncthreads();
  const int oh = Ry ? lr : h, ow = Rx ? lr : w;
  if (Ry) {                       // T = Ry * F   (oh x w), k ascending
    for (int e = tid; e < oh * w; e += 128) {
      const int r = e / w, c = e - r * w;
      double acc = 0.0;
      for (int k = 0; k < h; ++k) acc = __fma_rn(Ry[r * h + k], F[k * w + c], acc);
      T[e] = acc;
    }
  } else {
    for (int e = tid; e < h * w; e += 128) T[e] = F[e];
  }
  __syncthreads();
  for (int e = tid; e < oh * ow; e += 128) {   // X = T * Rx^T  (oh x ow)
    const int r = e / ow, c = e - r * ow;
    double acc;
    if (Rx) {
      acc = 0.0;
      for (int k = 0; k < w; ++k) acc = __fma_rn(T[r * w + k], Rx[c * w + k], acc);
    } else acc = T[r * w + c];
    const float v = (float)acc;
    X[e] = v;
    x_out[(size_t)s * oh * ow + e] = v;
  }
  __syncthreads();
  const int n = oh * ow;
  __shared__ float s_mean;
  if (tid == 0) s_mean = __fdiv_rn(np_pairwise_sum_f32(X, n), (float)n);
  __syncthreads();
  for (int e = tid; e < n; e += 128) { const float d = __fsub_rn(X[e], s_mean); D[e] = __fmul_rn(d, d); }
  __syncthreads();
  if (tid == 0) {
    const double mean_tr = train_stats[2 * s], std_tr = train_stats[2 * s + 1];
    float mean_o, std_o;
    if (adaptive) {
      const float in_mean = s_mean;
      const float in_std = sqrtf(__fdiv_rn(np_pairwise_sum_f32(D, n), (float)n));   // correctly rounded (hipcc's default for sqrt and divide); __fsqrt_rn is the native approximation
      // (1 - b) * mean_tr: Python floats (float64); b * np.float32: float32; their sum: the float64 term rounded to float32 first
      mean_o = __fadd_rn((float)((1.0 - blend) * mean_tr), __fmul_rn((float)blend, in_mean));
      if (in_std >= (float)1e-8) std_o = __fadd_rn((float)((1.0 - blend) * std_tr), __fmul_rn((float)blend, in_std));
      else std_o = (float)((1.0 - blend) * std_tr + blend * 1e-8);   // max() picked the Python float: float64 throughout
    } else {
      mean_o = (float)mean_tr; std_o = (float)std_tr;
    }
    aff_out[2 * s] = mean_o; aff_out[2 * s + 1] = std_o;
  }
}
}  // namespace srcfd

extern "C" {

int srcfd_resampler_create(int device, const double* Ry, const double* Rx, int in_h, int in_w, int out_h, int out_w, srcfd_resampler** out) {
  return srcfd::abi_guard("srcfd_resampler_create", [&]() -> int {
    if (!out || !Ry || !Rx || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) { set_error("srcfd_resampler_create: bad arguments"); return SRCFD_EINVAL; }
    *out = nullptr;
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || device < 0 || device >= ndev) {
      (void)hipGetLastError();
      set_error("srcfd_resampler_create: no such HIP device");
      return SRCFD_ENODEV;
    }
    std::unique_ptr<Resampler> r(new Resampler());
    r->device = device; r->H = in_h; r->W = in_w; r->OH = out_h; r->OW = out_w;
    HIPCHECK(hipSetDevice(device));
    std::vector<double> rxt((size_t)in_w * out_w);
    for (int o = 0; o < out_w; ++o)
      for (int w = 0; w < in_w; ++w) rxt[(size_t)w * out_w + o] = Rx[(size_t)o * in_w + w];
    auto is_identity = [](const double* R, int rows, int cols) {
      if (rows != cols) return false;
      for (int i = 0; i < rows; ++i)
        for (int j = 0; j < cols; ++j)
          if (std::fabs(R[(size_t)i * cols + j] - (i == j ? 1.0 : 0.0)) > 1e-13) return false;
      return true;
    };
    r->ry_identity = is_identity(Ry, out_h, in_h);
    r->rx_identity = is_identity(Rx, out_w, in_w);
    HIPCHECK(hipMalloc(&r->d_Ry, (size_t)out_h * in_h * sizeof(double)));
    HIPCHECK(hipMalloc(&r->d_RxT, rxt.size() * sizeof(double)));
    HIPCHECK(hipMemcpy(r->d_Ry, Ry, (size_t)out_h * in_h * sizeof(double), hipMemcpyHostToDevice));
    HIPCHECK(hipMemcpy(r->d_RxT, rxt.data(), rxt.size() * sizeof(double), hipMemcpyHostToDevice));
    *out = reinterpret_cast<srcfd_resampler*>(r.release());
    return SRCFD_OK;
  });
}

void srcfd_resampler_destroy(srcfd_resampler* r) { delete reinterpret_cast<Resampler*>(r); }

int srcfd_resample_device(srcfd_resampler* r, const float* in_dev, int n, double* out_dev, void* hip_stream) {
  return srcfd::abi_guard("srcfd_resample_device", [&]() -> int {
    if (!r || n < 0 || (n > 0 && (!in_dev || !out_dev))) { set_error("srcfd_resample_device: bad arguments"); return SRCFD_EINVAL; }
    if (n == 0) return SRCFD_OK;
    Resampler* rr = reinterpret_cast<Resampler*>(r);
    HIPCHECK(hipSetDevice(rr->device));
    return rr->run(in_dev, n, out_dev, reinterpret_cast<hipStream_t>(hip_stream));
  });
}

int srcfd_predict_resampled(srcfd_model* m, srcfd_resampler* r, const float* x, int n, const float* in_affine, const float* out_affine, double* y,
                            int flags, int64_t* n_nonfinite) {
  return srcfd::abi_guard("srcfd_predict_resampled", [&]() -> int {
    if (!m || !r || n < 0 || (n > 0 && (!x || !y))) { set_error("srcfd_predict_resampled: bad arguments"); return SRCFD_EINVAL; }
    srcfd::Model* mm = reinterpret_cast<srcfd::Model*>(m);
    Resampler* rr = reinterpret_cast<Resampler*>(r);
    const int* os = mm->desc.out_shape();
    if (os[0] != rr->H || os[1] != rr->W || os[2] != 1) { set_error("srcfd_predict_resampled: resampler input size != model output size"); return SRCFD_EINVAL; }
    if (mm->device != rr->device) { set_error("srcfd_predict_resampled: model and resampler live on different devices"); return SRCFD_EINVAL; }
    return mm->predict_host(x, n, in_affine, out_affine, nullptr, flags, n_nonfinite, [&](const float* y_dev, int first, int count) -> int {
      int rc = rr->reserve(count);
      if (rc) return rc;
      rc = rr->run(y_dev, count, rr->d_out, nullptr);
      if (rc) return rc;
      HIPCHECK(hipMemcpyAsync(y + (size_t)first * rr->OH * rr->OW, rr->d_out, (size_t)count * rr->OH * rr->OW * sizeof(double), hipMemcpyDeviceToHost,
                              nullptr));
      return SRCFD_OK;
    });
  });
}

int srcfd_predict_into_solver_state(srcfd_model* m, srcfd_resampler* r, const float* x, const float* in_affine, const float* out_affine,
                                    const srcfd_solver_bc bc[3], double* Var, int flags, int64_t* n_nonfinite) {
  return srcfd::abi_guard("srcfd_predict_into_solver_state", [&]() -> int {
    if (!m || !x || !bc || !Var) { set_error("srcfd_predict_into_solver_state: bad arguments"); return SRCFD_EINVAL; }
    srcfd::Model* mm = reinterpret_cast<srcfd::Model*>(m);
    Resampler* rr = reinterpret_cast<Resampler*>(r);
    const int* os = mm->desc.out_shape();
    if (os[2] != 1) { set_error("srcfd_predict_into_solver_state: single-channel models only"); return SRCFD_EINVAL; }
    if (rr && (os[0] != rr->H || os[1] != rr->W || mm->device != rr->device)) {
      set_error("srcfd_predict_into_solver_state: resampler does not match the model");
      return SRCFD_EINVAL;
    }
    const int ny = rr ? rr->OH : os[0], nx = rr ? rr->OW : os[1];
    srcfd::BcDev b{};
    std::vector<double> prof((size_t)3 * ny, 0.0);
    bool any_profile = false;
    for (int k = 0; k < 3; ++k) {
      for (int s = 0; s < 4; ++s) { b.type[k][s] = bc[k].type[s]; b.value[k][s] = bc[k].value[s]; }
      b.has_profile[k] = bc[k].left_profile != nullptr;
      if (bc[k].left_profile) { std::memcpy(&prof[(size_t)k * ny], bc[k].left_profile, sizeof(double) * ny); any_profile = true; }
    }
    const size_t var_elems = (size_t)3 * (nx + 2) * (ny + 2);
    return mm->predict_host(x, 3, in_affine, out_affine, nullptr, flags, n_nonfinite, [&](const float* y_dev, int first, int count) -> int {
      if (first != 0 || count != 3) { set_error("srcfd_predict_into_solver_state: internal chunking error"); return SRCFD_EINVAL; }
      const size_t need = var_elems + prof.size();
      if (need > mm->solver_state_elems) {
        if (mm->d_solver_state) { HIPCHECK(hipFree(mm->d_solver_state)); mm->d_solver_state = nullptr; mm->solver_state_elems = 0; }
        HIPCHECK(hipMalloc(&mm->d_solver_state, need * sizeof(double)));
        mm->solver_state_elems = need;
      }
      double* d_var = mm->d_solver_state;
      double* d_prof = any_profile ? d_var + var_elems : nullptr;
      if (any_profile) HIPCHECK(hipMemcpyAsync(d_prof, prof.data(), prof.size() * sizeof(double), hipMemcpyHostToDevice, nullptr));
      const unsigned blocks = (unsigned)((var_elems + 255) / 256);
      int rc = SRCFD_OK;
      if (rr) {
        rc = rr->reserve(3);
        if (rc) return rc;
        rc = rr->run(y_dev, 3, rr->d_out, nullptr);
        if (rc) return rc;
        hipLaunchKernelGGL((srcfd::solver_state_f64<double>), dim3(blocks), dim3(256), 0, nullptr, rr->d_out, ny, nx, b, d_prof, d_var);
      } else {
        hipLaunchKernelGGL((srcfd::solver_state_f64<float>), dim3(blocks), dim3(256), 0, nullptr, y_dev, ny, nx, b, d_prof, d_var);
      }
      HIPCHECK(hipGetLastError());
      HIPCHECK(hipMemcpyAsync(Var, d_var, var_elems * sizeof(double), hipMemcpyDeviceToHost, nullptr));
      HIPCHECK(hipStreamSynchronize(nullptr));
      return rc;
    });
  });
}

int srcfd_prepare_inputs_device(const double* fields_dev, int n, int h, int w, const double* Ry_dev, const double* Rx_dev, int lr,
                                const double* train_stats_dev, int adaptive, double blend, float* x_dev, float* in_affine_dev,
                                void* hip_stream) {
  return srcfd::abi_guard("srcfd_prepare_inputs_device", [&]() -> int {
    if (n < 0 || (n > 0 && (!fields_dev || !train_stats_dev || !x_dev || !in_affine_dev)) || h < 1 || w < 1 || h > srcfd::PREP_MAX ||
        w > srcfd::PREP_MAX || ((Ry_dev || Rx_dev) && (lr < 1 || lr > srcfd::PREP_MAX)) || (!Ry_dev) != (!Rx_dev)) {
      srcfd::set_error("srcfd_prepare_inputs_device: bad arguments (sides up to 32; Ry and Rx together or not at all)");
      return SRCFD_EINVAL;
    }
    if (n == 0) return SRCFD_OK;
    hipLaunchKernelGGL(srcfd::prepare_inputs_f64, dim3(n), dim3(128), 0, reinterpret_cast<hipStream_t>(hip_stream), fields_dev, h, w, Ry_dev,
                       Rx_dev, lr, train_stats_dev, adaptive, blend, x_dev, in_affine_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { srcfd::set_error(std::string("prepare_inputs launch failed: ") + hipGetErrorString(e)); return SRCFD_EHIP; }
    return SRCFD_OK;
  });
}

}  // extern "C"
