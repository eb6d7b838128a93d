/* srcfd.h -- C ABI of libsrcfd.so, the MI355X-native super-resolution engine.
 *
 * Drop-in boundary for ONE hot path of bitseal02/SR-for-CFD: the 10x10 ->
 * 400x400 convolutional-autoencoder SR call.  The reference has no FFI layer;
 * its boundary is the Python surface the solver scripts call, so every entry
 * point below cites the reference call it replaces (paths relative to the
 * reference checkout).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions: plain C, no torch / STL types.  Every function returns an int
 * status (0 = SRCFD_OK, negative = error) unless noted, never throws, and
 * leaves a thread-local message readable through srcfd_last_error().
 * A handle is not thread-safe; distinct handles may be used from distinct
 * threads.  Compute entry points fail with SRCFD_ENODEV when no HIP device is
 * present -- there is no CPU fallback.
 */
#ifndef SRCFD_H
#define SRCFD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRCFD_OK 0
#define SRCFD_ENOENT (-2)   /* file missing            -> Python FileNotFoundError (PyCFD_ML_accelerated.py:1080-1087) */
#define SRCFD_EIO (-5)      /* unreadable / malformed  -> Python OSError           (PyCFD_ML_accelerated.py:835-837)   */
#define SRCFD_ENOMEM (-12)
#define SRCFD_ENODEV (-19)  /* no HIP device / extension cannot run */
#define SRCFD_EINVAL (-22)
#define SRCFD_EKEY (-126)   /* missing key             -> Python KeyError          (PyCFD_ML_accelerated.py:822-825)   */
#define SRCFD_EHIP (-1000)  /* HIP runtime error, text in srcfd_last_error() */

typedef enum {
  SRCFD_F32 = 0, SRCFD_F64 = 1, SRCFD_I32 = 2, SRCFD_I64 = 3, SRCFD_U8 = 4, SRCFD_STR = 5,
  SRCFD_BF16 = 16, SRCFD_F16 = 17
} srcfd_dtype;

/* Arithmetic the network is evaluated in. */
typedef enum {
  SRCFD_PREC_FP32 = 0,       /* f32 MFMA (v_mfma_f32_32x32x2_f32), exact f32 products: parity path     */
  SRCFD_PREC_BF16 = 1,       /* bf16 operands, f32 accumulate, fused tail: throughput path             */
  SRCFD_PREC_FP32_NAIVE = 2, /* one thread per output, f32 FMA chain: bring-up / cross-check           */
  SRCFD_PREC_F16 = 3,        /* f16 operands, f32 accumulate (BASELINE config 5)                        */
  SRCFD_PREC_FP32X3 = 4      /* f32 storage and vector math; the wide decoder GEMMs (ConvT#0, ConvT#1) and the
                                streaming tail's first layer as six bf16 MFMAs on operands split exactly into
                                three bf16 terms (kernels_x3.hip, tail32<X3>): f32-grade results (<= 1e-5 vs the
                                float64 oracle, measured 5e-7) at 6/16 of the f32 matrix time, from 64 samples per
                                call on; smaller calls and everything else run the SRCFD_PREC_FP32 kernels.   */
} srcfd_precision;

typedef enum {
  SRCFD_LAYER_CONV2D = 1, SRCFD_LAYER_CONV2D_TRANSPOSE = 2, SRCFD_LAYER_DENSE = 3,
  SRCFD_LAYER_FLATTEN = 4, SRCFD_LAYER_RESHAPE = 5
} srcfd_layer_kind;

typedef enum { SRCFD_ACT_LINEAR = 0, SRCFD_ACT_SWISH = 1, SRCFD_ACT_RELU = 2, SRCFD_ACT_SIGMOID = 3, SRCFD_ACT_TANH = 4 } srcfd_activation;

/* One Keras layer (sr-ae-conv.ipynb:c162-169, c277-287).  Kernel layouts are
 * Keras': Conv2D (kh,kw,Cin,Cout); Conv2DTranspose (kh,kw,Cout,Cin); Dense
 * (in,out).  Pointers are host memory, copied at create. */
typedef struct {
  int kind;        /* srcfd_layer_kind */
  int activation;  /* srcfd_activation */
  int kh, kw, stride, same_padding; /* conv / conv-transpose */
  int cin, cout;   /* channels (Dense: in/out features) */
  int reshape[3];  /* RESHAPE target (h,w,c) */
  const float* kernel;
  const float* bias;
  const char* name; /* optional Keras layer name (used by srcfd_model_save_h5); NULL -> layer_<i> */
} srcfd_layer;

typedef struct srcfd_model srcfd_model;
typedef struct srcfd_h5 srcfd_h5;
typedef struct srcfd_h5w srcfd_h5w;
typedef struct srcfd_trainer srcfd_trainer;
typedef struct srcfd_resampler srcfd_resampler;

const char* srcfd_last_error(void);
const char* srcfd_version(void);
/* Number of HIP devices (0 on a CPU-only box; never an error). */
int srcfd_device_count(void);

/* ---- model lifetime ------------------------------------------------------
 * srcfd_model_load_h5 replaces the pair of
 *   tf.keras.models.load_model(encoder_file, compile=False)
 *   tf.keras.models.load_model(decoder_file, compile=False)
 * + SuperResolutionAE(encoder_lr, decoder_hr)   (PyCFD_ML_accelerated.py:831-833,
 * bfs_ml_accelerated.py:1069-1071): parses `model_config` and
 * `model_weights/<layer>/<layer>/{kernel,bias}` from legacy Keras-H5 files and
 * chains the sub-models.  Either path may be NULL to load one half.
 * device < 0 builds a host-only handle (shape queries / weight access only). */
int srcfd_model_load_h5(const char* encoder_h5, const char* decoder_h5, int device, srcfd_model** out);
/* Same from in-memory layers (tests, training hand-off). in_shape = (h,w,c). */
int srcfd_model_create(const srcfd_layer* layers, int n_layers, const int in_shape[3], int device, srcfd_model** out);
void srcfd_model_destroy(srcfd_model* m);

int srcfd_model_input_shape(const srcfd_model* m, int shape[3]);
int srcfd_model_output_shape(const srcfd_model* m, int shape[3]);
int srcfd_model_num_layers(const srcfd_model* m);
/* Layer i: fills *layer with pointers into the handle's host copy of the weights. */
int srcfd_model_get_layer(const srcfd_model* m, int i, srcfd_layer* layer, char* name, size_t name_len);
/* Multiply-accumulates per sample (140 024 128 for encoder_10+decoder_400). */
int64_t srcfd_model_macs_per_sample(const srcfd_model* m);
int srcfd_model_set_precision(srcfd_model* m, int precision);
int srcfd_model_get_precision(const srcfd_model* m);
/* Allocates now what the first n-sample forward at the CURRENT precision would allocate or pack lazily (activation
 * workspaces, 16-bit operand packs), so that a latency- or throughput-critical first call does no set-up work.  Optional:
 * every forward reserves what it needs itself.  The reference pays the equivalent cost inside its first
 * `predict` (graph tracing, PyCFD_ML_accelerated.py:858). */
int srcfd_model_reserve(srcfd_model* m, int n);
/* What an n-sample forward at `precision` keeps allocated, WITHOUT allocating anything (works on a host-only handle): bytes[0] device
 * activation workspace (what srcfd_model_reserve reserves), bytes[1] device weights and operand packs (upper bound), bytes[2] device
 * staging of the host-buffer entry (srcfd_predict), bytes[3] host bytes of one result of the call (page-locked when it comes from
 * srcfd_host_alloc).  For capacity planning of several ranks on one host (bench.py's dry run); the reference has no counterpart
 * (TensorFlow grows its arena on demand, PyCFD_ML_accelerated.py:858). */
int srcfd_model_footprint(const srcfd_model* m, int n, int precision, size_t bytes[4]);
/* 1 when the bf16/f16 fused decoder_400 kernels apply to this layer graph. */
int srcfd_model_has_fused_path(const srcfd_model* m);

/* ---- forward -------------------------------------------------------------
 * srcfd_predict replaces `inference_model.predict(x, verbose=0)`
 * (PyCFD_ML_accelerated.py:858, bfs_ml_accelerated.py:1109, sr-ae-conv.ipynb:c349):
 * x is float32 NHWC (n,h,w,c) C-contiguous host memory, y a caller-allocated
 * float32 (n,oh,ow,oc) host buffer.  Blocks until y is complete.
 *
 * Optional fused pre/post-processing, per sample i (both may be NULL):
 *   in_affine[2i..2i+1]  = (mean, std): x := (x - mean) / std in float32
 *       (standardize_with_stats, PyCFD_ML_accelerated.py:665-668; std==0 -> 1e-8)
 *   out_affine[2i..2i+1] = (mean, std): y := y * std + mean in float32
 *       (inverse_standardize, PyCFD_ML_accelerated.py:671-673).  SRCFD_PREC_FP32 (the parity path): two roundings,
 *       bit for bit numpy's float32 result; the 16-bit precisions fuse the two into ONE fma (<= 1 ulp from it)
 * flags: SRCFD_FLAG_NAN_GUARD zero-fills NaN/Inf in y and counts them in
 * *n_nonfinite (PyCFD_ML_accelerated.py:869-876). */
#define SRCFD_FLAG_NAN_GUARD 1
int srcfd_predict(srcfd_model* m, const float* x, int n, const float* in_affine, const float* out_affine,
                  float* y, int flags, int64_t* n_nonfinite);

/* Page-locked host memory for the arrays handed to srcfd_predict.  A result (y) that lives in such memory -- from here, or any
 * hipHostMalloc / hipHostRegister'ed range -- is filled at the PCIe rate with the copy of one chunk overlapping the kernels of the
 * next; a pageable result is staged by the runtime (about 45 GB/s, nothing overlaps) and, when freshly allocated, also pays its
 * first-touch page faults.  The Python surface keeps a recycling pool of these behind predict() (engine.py). */
int srcfd_host_alloc(size_t bytes, void** out);
void srcfd_host_free(void* p);

/* Device-resident variant for the batched path: x_dev float32 (n,h,w,c),
 * affines float32 device arrays or NULL, y_dev of dtype out_dtype (SRCFD_F32,
 * SRCFD_BF16 or SRCFD_F16), nonfinite_dev an optional device int64 counter
 * that is ADDED to.  Enqueues on hip_stream (a hipStream_t, NULL = default
 * stream) and returns without synchronising. */
int srcfd_predict_device(srcfd_model* m, const void* x_dev, int n, const float* in_affine_dev,
                         const float* out_affine_dev, void* y_dev, int out_dtype, int flags,
                         int64_t* nonfinite_dev, void* hip_stream);
/* Largest n one srcfd_predict_device call processes without internal chunking
 * at the current precision, and the workspace bytes it holds for that. */
int srcfd_model_workspace(srcfd_model* m, int n, size_t* bytes);

/* Per-kernel timing of the last srcfd_predict_device call made with profiling
 * enabled: HIP events on the launch stream around every kernel.  names is a
 * '\n'-joined list; ms has one entry per name. */
int srcfd_model_set_profiling(srcfd_model* m, int enable);
int srcfd_model_get_profile(srcfd_model* m, char* names, size_t names_len, float* ms, int* count, int max_count);
/* Test hook: copies the first `bytes` of inter-kernel activation buffer `index`
 * (0 or 1) of the bf16/f16 pipeline to host memory after synchronising.  After a
 * forward, index 0 is ConvT#1's output (n,50,50,64), 16-bit, scaled by log2(e); index 1
 * is ConvT#0's (n,25,25,128) only when the generic GEMM path ran (env SRCFD_MID=0) --
 * the default fused ConvT#0->ConvT#1 kernel never materialises it. */
int srcfd_model_debug_activation(srcfd_model* m, int index, void* dst, size_t bytes);
/* Test hook: which implementation of each stage the handle's LAST forward ran, as "key=value" words separated by blanks, e.g.
 * "precision=bf16 encoder=enc16 dense_1=dense1_16 middle=mid16 tail=tail16 tail_seg=10 graph=replay".  The A/B switches SRCFD_ENC,
 * SRCFD_DENSE1, SRCFD_MID, SRCFD_TAIL (=s), SRCFD_NO_ENC32, SRCFD_NO_DENSE_SKINNY and SRCFD_TAIL_SEG are read from the environment on every
 * srcfd_predict* call; each selects a complete second implementation (the parity tests compare the two), none skips work, and a
 * captured hipGraph is replayed only under the switches it was captured with.  graph: eager | capture | replay. */
int srcfd_model_last_plan(const srcfd_model* m, char* buf, size_t buf_len);

/* ---- stats file ----------------------------------------------------------
 * `key value` lines, '#' comments (PyCFD_ML_accelerated.py:787-797).  Looks up
 * mean{lr}_{c}, std{lr}_{c}, mean{hr}_{c}, std{hr}_{c} for c in u,v,p
 * (PyCFD_ML_accelerated.py:800-809) into out[12] = lr(u,v,p)x(mean,std) then
 * hr(u,v,p)x(mean,std).  SRCFD_ENOENT / SRCFD_EKEY as the reference raises. */
int srcfd_stats_load(const char* path, int lr_dim, int hr_dim, double out[12]);
/* Writer of the same format (sr-ae-conv.ipynb:c589-603). */
int srcfd_stats_save(const char* path, int lr_dim, int hr_dim, const double in[12]);

/* ---- HDF5 subset (legacy Keras-H5, solver field dumps) -------------------
 * Replaces h5py for this path: reads what `load_model` reads
 * (PyCFD_ML_accelerated.py:831-832) and the coarse-field files the solvers
 * write (PyCFD_ML_accelerated.py:517-544); writes the layout
 * `encoder.save(...h5)` produces (sr-ae-conv.ipynb:c584-586). */
int srcfd_h5_open(const char* path, srcfd_h5** out);
void srcfd_h5_close(srcfd_h5* f);
/* '\n'-joined child names of a group; *needed = bytes incl. terminator. */
int srcfd_h5_list(srcfd_h5* f, const char* group, char* buf, size_t buf_len, size_t* needed);
/* kind: 0 absent, 1 group, 2 dataset. */
int srcfd_h5_kind(srcfd_h5* f, const char* path);
int srcfd_h5_dataset_info(srcfd_h5* f, const char* path, int* dtype, int* rank, uint64_t dims[8]);
int srcfd_h5_read(srcfd_h5* f, const char* path, void* dst, size_t dst_bytes, int as_dtype);
/* String attribute ('\n'-joined when an array); numeric attribute as doubles. */
int srcfd_h5_attr_string(srcfd_h5* f, const char* obj, const char* name, char* buf, size_t buf_len, size_t* needed);
int srcfd_h5_attr_numeric(srcfd_h5* f, const char* obj, const char* name, double* out, int max_count, int* count);
int srcfd_h5_attr_names(srcfd_h5* f, const char* obj, char* buf, size_t buf_len, size_t* needed);

int srcfd_h5w_create(srcfd_h5w** out);
void srcfd_h5w_free(srcfd_h5w* w);
int srcfd_h5w_group(srcfd_h5w* w, const char* path);
int srcfd_h5w_dataset(srcfd_h5w* w, const char* path, int dtype, int rank, const uint64_t* dims, const void* data);
/* n_strings == 0 writes a scalar string attribute from strings[0]. */
int srcfd_h5w_attr_strings(srcfd_h5w* w, const char* obj, const char* name, const char* const* strings,
                           int n_strings, int is_scalar, int utf8);
int srcfd_h5w_attr_numeric(srcfd_h5w* w, const char* obj, const char* name, int dtype, const void* value, int count, int is_scalar);
int srcfd_h5w_save(srcfd_h5w* w, const char* path);
/* Saves the handle's weights as legacy Keras-H5 sub-model files
 * (sr-ae-conv.ipynb:c584-585); split_at = index of the first decoder layer. */
int srcfd_model_save_h5(const srcfd_model* m, const char* encoder_h5, const char* decoder_h5);
/* The whole-model file `superres_model.save("superres_{lr}to{hr}_vanilla_ae_{suffix}.h5")` (sr-ae-conv.ipynb:c586):
 * encoder and decoder in ONE legacy Keras-H5 file (groups model_weights/<sub-model>/<layer>/{kernel,bias}, the layout
 * Keras 3.8 writes for a subclassed Model; unpinned -- the reference's superres files are absent, .MISSING_LARGE_BLOBS:29-31).
 * Loading takes the architecture from the nested sub-model configs when the file carries them and otherwise assumes the
 * reference's encoder_10 / decoder_400 definition (sr-ae-conv.ipynb:c162-169, c277-287). */
int srcfd_model_save_superres_h5(const srcfd_model* m, const char* superres_h5);
int srcfd_model_load_superres_h5(const char* superres_h5, int device, srcfd_model** out);

/* ---- device-side resampling of the SR output (BFS aspect-ratio correction) -------------------
 * Replaces `reshape_square_to_rectangular` (bfs_ml_accelerated.py:104-145): per component
 * RectBivariateSpline(y_sq, x_sq, field, kx=3, ky=3)(y_rect, x_rect) is linear in `field` and
 * separable, i.e. out = Ry * field * Rx^T with the 1-D interpolating-spline matrices
 * Ry [out_h][in_h], Rx [out_w][in_w] (row-major float64, host; built once by the caller, see
 * sr-for-cfd_amd/resample.py).  The handle keeps them on `device`. */
int srcfd_resampler_create(int device, const double* Ry, const double* Rx, int in_h, int in_w, int out_h, int out_w,
                           srcfd_resampler** out);
void srcfd_resampler_destroy(srcfd_resampler* r);
/* in_dev float32 (n,in_h,in_w) -> out_dev float64 (n,out_h,out_w), enqueued on hip_stream. */
int srcfd_resample_device(srcfd_resampler* r, const float* in_dev, int n, double* out_dev, void* hip_stream);
/* srcfd_predict followed by the resampling, without the float32 result leaving the device:
 * `predict` + inverse standardise + NaN guard (bfs_ml_accelerated.py:1109-1127) + resample back
 * (:1130-1135).  y is float64 host (n,out_h,out_w), like scipy returns it. */
int srcfd_predict_resampled(srcfd_model* m, srcfd_resampler* r, const float* x, int n, const float* in_affine,
                            const float* out_affine, double* y, int flags, int64_t* n_nonfinite);

/* ---- input preparation on the device (batched BFS calls) ------------------------------------------
 * Replaces, per sample (one component of one coarse field), what `ml_super_resolution` does before `predict` in the BFS
 * solver: `reshape_rectangular_to_square` (bfs_ml_accelerated.py:59-101; X = Ry F Rx^T with the 1-D spline matrices
 * Ry [lr][h], Rx [lr][w], float64, DEVICE; both NULL = no resampling), `.astype(np.float32)` (:1086), and the adaptive
 * blend of the training statistics with np.mean / np.std of the float32 field (:1091-1097; adaptive = 0: the training
 * statistics as they are).  fields_dev float64 (n,h,w); train_stats_dev float64 (n,2) = (mean, std) of each sample's
 * component; writes x_dev float32 (n,lr,lr) [or (n,h,w)] and in_affine_dev float32 (n,2), the arguments
 * srcfd_predict_device takes.  The statistics reproduce numpy's float32 arithmetic (pairwise sums, NumPy-2 scalar
 * promotion) bit for bit.  Sides up to 32. */
int srcfd_prepare_inputs_device(const double* fields_dev, int n, int h, int w, const double* Ry_dev, const double* Rx_dev, int lr,
                                const double* train_stats_dev, int adaptive, double blend, float* x_dev, float* in_affine_dev,
                                void* hip_stream);

/* ---- hand-off into the solver state ---------------------------------------------------------
 * Replaces the three transposed assignments `solver.Var[k, 1:-1, 1:-1] = ml_initial_fields[c].T` and the
 * ghost-cell pass `_apply_bc_wrapper(k)` that follow the SR call (PyCFD_ML_accelerated.py:936-943,
 * bfs_ml_accelerated.py:1211-1218): x holds the u, v, p samples of ONE field; Var is the solver's
 * float64 (3, nx+2, ny+2) host array and is written completely (corners 0, as in a fresh solver).
 * bc[k]: apply_bc_configured's arrays (PyCFD_ML_accelerated.py:118-146), order left, right, top,
 * bottom; type 0 = Dirichlet (ghost = 2*value - inner), 1 = Neumann (ghost = inner).  left_profile:
 * optional [ny] Dirichlet values that override the left boundary row by row (the BFS inlet/wall mix,
 * bfs_ml_accelerated.py:524-562).  r: optional resampler applied first (BFS), or NULL. */
typedef struct srcfd_solver_bc {
  int type[4];
  double value[4];
  const double* left_profile;
} srcfd_solver_bc;
int srcfd_predict_into_solver_state(srcfd_model* m, srcfd_resampler* r, const float* x, const float* in_affine,
                                    const float* out_affine, const srcfd_solver_bc bc[3], double* Var, int flags,
                                    int64_t* n_nonfinite);

/* ---- coarse-mesh solver (the producer of the SR call's input) ------------------------------------
 * Replaces, for the lid-driven cavity, `CFDSolver(mesh, fluid, settings, bc).solve()` as `run_coarse_simulation` uses it
 * (PyCFD_ML_accelerated.py:694-761; kernels :110-328, time loop :396-505): float64, host.  bc_type / bc_value are
 * apply_bc_configured's arrays per variable (u, v, p), order left, right, top, bottom; 0 = Dirichlet, 1 = Neumann
 * (`_get_bc_arrays`, :352-375).  var_out receives the solver state Var (3, nx+2, ny+2) including ghost cells; the
 * coarse fields the SR call takes are Var[k, 1:-1, 1:-1].T (:755-759).  Returns SRCFD_EINVAL with a message when the
 * residuals turn NaN / Inf (the reference raises ValueError, :487-492). */
#define SRCFD_SCHEME_QUICK 0
#define SRCFD_SCHEME_UPWIND 1
#define SRCFD_CASE_LDC 0
#define SRCFD_CASE_BFS 1
typedef struct srcfd_coarse_problem {
  int nx, ny;
  double lx, ly;
  double reynolds, rho, dt;
  int scheme;                /* SRCFD_SCHEME_* */
  int max_iterations;
  double tolerance[3];       /* convergence_criteria u, v, p on rms / dt */
  int bc_type[3][4];
  double bc_value[3][4];
  /* backward-facing step (bfs_ml_accelerated.py:471-673): SRCFD_CASE_BFS overrides the left boundary with a wall below
   * step_height and a parabolic inlet (bulk velocity Ub over channel_height) above it, and under-relaxes u, v, p with
   * relax[] after each solve.  SRCFD_CASE_LDC ignores these five fields. */
  int case_type;             /* SRCFD_CASE_* */
  double relax[3];
  double step_height, channel_height, bulk_velocity;
} srcfd_coarse_problem;
int srcfd_coarse_solve(const srcfd_coarse_problem* problem, double* var_out, int* iterations, double rms[3]);

/* ---- training -----------------------------------------------------------
 * One optimisation step of SuperResolutionAE, split so that a data-parallel driver can put its
 * gradient all-reduce between the two halves (one flat f32 buffer per step).
 * Replaces `SuperResolutionAE.train_step` + `Adam()` (sr-ae-conv.ipynb:c306-320, c556).
 * Parameters, gradients and Adam moments are caller-owned flat f32 DEVICE arrays in Keras'
 * trainable_weights order (per layer: kernel then bias, Keras layouts); srcfd_trainer_get_params
 * returns the model's current values in that order (host). */
int srcfd_trainer_create(const srcfd_model* m, int max_batch, srcfd_trainer** out);
void srcfd_trainer_destroy(srcfd_trainer* t);
int64_t srcfd_trainer_num_params(const srcfd_trainer* t);
int srcfd_trainer_get_params(const srcfd_trainer* t, float* params_host);
/* Forward + backward on n <= max_batch samples: x_dev (n,h,w,c) inputs, y_dev targets of the model's
 * output shape.  ADDS d/dparams of loss_scale * sum((pred - y)^2) into grads_dev (zero it first; pass
 * loss_scale = 1 / (global batch * output elements) for Keras' reduce_mean(mse)) and adds the local sum of
 * squared errors to *sse_dev (device double, may be NULL).  Enqueues on hip_stream, no synchronisation: part of the weight
 * gradients runs on a stream of the trainer's own that forks from and joins hip_stream inside the call.  From the second
 * call with the same params_dev / grads_dev / sse_dev / n / loss_scale the step is replayed as one hipGraph (x_dev and
 * y_dev are first copied to staging buffers, so they need not stay at one address). */
int srcfd_trainer_forward_backward(srcfd_trainer* t, const float* params_dev, const float* x_dev, const float* y_dev, int n,
                                   float loss_scale, float* grads_dev, double* sse_dev, void* hip_stream);
/* The same with flags.  SRCFD_TRAIN_OVERWRITE: grads_dev and *sse_dev are WRITTEN, not added into (every parameter's gradient is
 * stored by exactly one thread of the step's last launch): a step needs no zero-fill launches in front of it.  Without the flag
 * this is srcfd_trainer_forward_backward (accumulation over several calls, e.g. micro-batches of one optimiser step). */
#define SRCFD_TRAIN_OVERWRITE 1
/* SRCFD_TRAIN_SAME_PARAMS: params_dev holds exactly what it held at this trainer's previous forward_backward call (the second and
 * later micro-batches of one optimiser step): the per-call re-packing of the parameters into the kernels' operand layouts (one
 * 18 us launch) is skipped.  Wrong gradients, silently, if the parameters did change -- the caller's contract. */
#define SRCFD_TRAIN_SAME_PARAMS 2
int srcfd_trainer_forward_backward_ex(srcfd_trainer* t, const float* params_dev, const float* x_dev, const float* y_dev, int n,
                                      float loss_scale, float* grads_dev, double* sse_dev, int flags, void* hip_stream);
/* Keras Adam update, step counted from 1: alpha_t = lr*sqrt(1-beta2^t)/(1-beta1^t); p -= alpha_t*m/(sqrt(v)+eps). */
int srcfd_adam_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, int64_t n, int step, float lr,
                    float beta1, float beta2, float eps, void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* SRCFD_H */
