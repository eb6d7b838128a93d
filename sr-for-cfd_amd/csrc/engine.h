// Internal engine types (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <new>
#include <string>
#include <vector>

#include "abi_guard.h"
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

// Functional A/B switches: every one selects a COMPLETE alternative implementation of a stage (no work is skipped), for the parity
// tests and the tools.  They are read from the environment once per srcfd_predict* call (Switches::from_env), are part of the hipGraph
// key (a captured forward is replayed only under the switches it was captured with) and are reported by srcfd_model_last_plan.
struct Switches {
  bool enc16 = true;     // SRCFD_ENC=0: layer-by-layer 16-bit encoder instead of the one-launch enc16
  bool mid16 = true;     // SRCFD_MID=0: generic GEMMs instead of the fused ConvT#0 -> ConvT#1 kernel
  int mid_shape = 3;     // workgroup shape of mid16 (SRCFD_MID=1, 2, 3): 1 = 8 waves x 32 pixels, two workgroups per CU (128 registers; rounds 1-2);
                         // 2 = 8 waves x 64 pixels, one per CU; 3 = 4 waves x 64 pixels, two per CU (256 registers; default)
  bool dense1_16 = true; // SRCFD_DENSE1=0: dense_1 on the generic 16-bit GEMM
  bool enc32 = true;     // SRCFD_NO_ENC32=1: layer-by-layer f32 encoder
  bool skinny32 = true;  // SRCFD_NO_DENSE_SKINNY=1: dense_1 on the generic f32 GEMM
  bool tail16s = false;  // SRCFD_TAIL=s: the software-pipelined 8-wave tail kernel (kernels_tail16s.hip) instead of the 16-wave, stage-by-stage one (measured slower)
  int mid_waves = 0;     // SRCFD_MID_WAVES: other workgroup shapes of mid16 (4, 16: waves per workgroup at 32 pixels per wave); 0 = the shape mid_shape selects
  int mid_order = 0;     // SRCFD_MID_ORDER=1: mid16's workgroups dispatched with output phases 0 / 3 and 1 / 2 alternating instead of phase by phase
  int tail_seg = 0;      // SRCFD_TAIL_SEG: segments per sample of the 16-bit tail (1, 2, 5, 10, 25); 0 = chosen per batch
  unsigned bits() const {
    return (enc16 ? 1u : 0u) | (mid16 ? 2u : 0u) | (dense1_16 ? 4u : 0u) | (enc32 ? 8u : 0u) | (skinny32 ? 16u : 0u) | (tail16s ? 32u : 0u) |
           ((unsigned)mid_shape << 6) | ((unsigned)tail_seg << 8) | ((unsigned)mid_waves << 16) | ((unsigned)mid_order << 24);
  }
  bool all_default() const { return enc16 && mid16 && mid_shape == Switches().mid_shape && dense1_16 && enc32 && skinny32 && !tail16s && tail_seg == 0 && mid_waves == 0 && mid_order == Switches().mid_order; }
  static Switches from_env();
};

// What the last forward of a handle ran (srcfd_model_last_plan): the switches it saw, the tail segmentation that was launched and how
// the launches were issued.
struct Plan {
  Switches sw;
  int precision = 0;
  int tail_seg = 0;   // 16-bit tail: segments per sample actually launched (last chunk)
  int graph = 0;      // 0 plain launches, 1 captured now and launched as a graph, 2 replay of an earlier capture
  bool fused = false;
};

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
  // SRCFD_PREC_FP32X3: ops the split-bf16 GEMM takes (kernels_x3.hip): x3_off[i] = offset of op i's three weight planes in pack_x3
  // (elements), or -1.  Built at create, uploaded with the f32 pack.
  std::vector<int64_t> x3_off;
  int64_t t32_w1x = -1, t32_w2x = -1;   // the streaming tail's first- and second-layer fragments in pack_x3 (kernels.h, Tail32Params::w1x / w2x), or -1
  std::vector<uint16_t> pack_x3;
  uint16_t* d_pack_x3 = nullptr;

  float* buf[2] = {nullptr, nullptr};
  int ws_chunk = 0;
  size_t ws_per_sample = 0;   // elements per sample the two buffers were sized for (depends on the precision)
  float* d_splitk = nullptr;  // split-K slabs of the skinny f32 GEMMs
  double* d_solver_state = nullptr;  // (3, nx+2, ny+2) + row profiles of the solver hand-off
  size_t solver_state_elems = 0;
  size_t splitk_floats = 0;
  float* d_x_stage = nullptr;
  float* d_y_stage = nullptr;    // host-buffer entry: result of the chunk being computed ...
  float* d_y_stage2 = nullptr;   // ... and of the chunk being copied out (page-locked destinations: copy and compute overlap)
  float* d_aff = nullptr;
  int stage_chunk = 0;
  hipStream_t copy_stream = nullptr;
  hipEvent_t ev_computed[2] = {nullptr, nullptr}, ev_copied[2] = {nullptr, nullptr};
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
    unsigned switches = 0;   // Switches::bits(): the A/B switches select different kernels, so they belong to the key
    bool operator==(const GraphKey& o) const {
      return x == o.x && y == o.y && ain == o.ain && aout == o.aout && nf == o.nf && n == o.n && out_dtype == o.out_dtype &&
             flags == o.flags && precision == o.precision && switches == o.switches;
    }
  };
  GraphKey graph_key, last_key;
  Switches sw;            // of the call in progress
  Plan plan, graph_plan;  // of the last forward / of the captured graph
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
