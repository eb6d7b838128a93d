// Internal engine types (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <string>
#include <vector>

#include "kernels.h"
#include "model.h"

namespace srcfd {

extern thread_local std::string g_last_error;
void set_error(const std::string& m);

struct Op {
  GemmDesc d;           // M filled per call (rows per image * batch)
  size_t w_off = 0;     // into the packed float buffer: B[K][Npad]
  size_t b_off = 0;     // bias[Npad]
  int layer = 0;        // index into ModelDesc::layers
  std::string name;
};

void build_plan(const ModelDesc& desc, std::vector<Op>& ops, std::vector<float>& pack);

struct ProfEvent {
  hipEvent_t a, b;
  std::string name;
};

struct FusedState;  // bf16/f16 path (fused_bf16.hip)

struct Model {
  ModelDesc desc;
  int device = -1;
  int precision = SRCFD_PREC_FP32;
  bool has_fused = false;

  std::vector<Op> ops;
  std::vector<float> pack;  // host copy of packed weights
  // ops[pair_op], ops[pair_op + 1]: two kernel == stride == 2 transposed convolutions (32 -> 16 -> 8 channels) that run as
  // one kernel (kernels.h, PairDesc); -1: none.  Offsets into `pack`.
  int pair_op = -1;
  size_t pair_wa = 0, pair_ba = 0, pair_wb = 0, pair_bb = 0;
  // ops[triple_op .. +2]: the same with a 64 -> 32 layer in front (kernels.h, TripleDesc); takes precedence over the pair
  int triple_op = -1;
  size_t tri_w1 = 0, tri_b1 = 0, tri_w2 = 0, tri_b2 = 0, tri_w3 = 0, tri_b3 = 0;
  // ops[tail32_op .. +3]: that chain followed by the network's last layer, a 3x3 SAME conv 8 -> 1 (kernels.h, Tail32Params):
  // one streaming kernel; takes precedence over the triple
  int tail32_op = -1;
  // ops[0..3] = conv2d (1->64, s2) -> conv2d_1 (64->128 on 5x5) -> dense (3200->128) -> latent (128->nl) run as one launch (kernels.h, Enc32Params)
  bool enc32_ok = false;
  size_t enc32_w2 = 0;     // conv2d_1 A fragments in `pack`
  size_t t32_w1 = 0, t32_b1 = 0, t32_w2 = 0, t32_b2 = 0, t32_w3 = 0, t32_b3 = 0, t32_wc = 0;
  int num_cus = 256;
  float* d_pack = nullptr;

  float* buf[2] = {nullptr, nullptr};
  int ws_chunk = 0;
  size_t ws_per_sample = 0;   // elements per sample the two buffers were sized for (depends on the precision)
  float* d_splitk = nullptr;  // split-K slabs of the skinny f32 GEMMs
  double* d_solver_state = nullptr;  // (3, nx+2, ny+2) + row profiles of the solver hand-off
  size_t solver_state_elems = 0;
  size_t splitk_floats = 0;
  float* d_x_stage = nullptr;
  float* d_y_stage = nullptr;
  float* d_aff = nullptr;
  int stage_chunk = 0;
  unsigned long long* d_nonfinite = nullptr;

  bool profiling = false;
  std::vector<ProfEvent> prof_events;
  size_t prof_used = 0;

  FusedState* fused = nullptr;

  // hipGraph replay of the fused 16-bit pipeline: a call whose arguments equal the previous call's is
  // captured once on a private stream and replayed afterwards (removes ~8 launch gaps per batch)
  struct GraphKey {
    const void* x = nullptr; const void* y = nullptr; const float* ain = nullptr; const float* aout = nullptr;
    unsigned long long* nf = nullptr; int n = -1, out_dtype = 0, flags = 0, precision = 0;
    bool operator==(const GraphKey& o) const {
      return x == o.x && y == o.y && ain == o.ain && aout == o.aout && nf == o.nf && n == o.n && out_dtype == o.out_dtype &&
             flags == o.flags && precision == o.precision;
    }
  };
  GraphKey graph_key, last_key;
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t graph_stream = nullptr;
  void drop_graph();

  ~Model();
  int init_device();
  void free_workspace();
  size_t max_act_elems() const;
  int chunk_cap() const;
  int ensure_workspace(int n);
  int launch(const char* name, hipStream_t s, const std::function<hipError_t()>& fn);
  int forward_generic(const float* x_dev, int n, const float* aff_in, const float* aff_out, void* y_dev, int out_dtype, int flags,
                      unsigned long long* nonfinite, hipStream_t s);
  int predict_device(const void* x_dev, int n, const float* aff_in, const float* aff_out, void* y_dev, int out_dtype, int flags,
                     unsigned long long* nonfinite, hipStream_t s);
  // sink (optional): consumes each chunk's device result [first, first+count) on the default stream instead of the copy into y
  int predict_host(const float* x, int n, const float* aff_in, const float* aff_out, float* y, int flags, int64_t* n_nonfinite,
                   const std::function<int(const float* y_dev, int first, int count)>& sink = nullptr);
};

// bf16 / f16 fused path for the encoder_10 + decoder_400 graph.
int fused_init(Model& m);
void fused_free(Model& m);
int fused_debug_read(Model& m, int index, void* dst, size_t bytes);
int fused_reserve(Model& m, int n);
int fused_forward(Model& m, const float* x_dev, int n, const float* aff_in, const float* aff_out, void* y_dev, int out_dtype,
                  int flags, unsigned long long* nonfinite, hipStream_t s);

}  // namespace srcfd
