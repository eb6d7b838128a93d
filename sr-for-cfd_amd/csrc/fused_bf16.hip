// Host side of the bf16 / f16 throughput path for the encoder_10 + decoder_400
// graph: weight repacking (16-bit, log2e folding, MFMA fragment order) and the
// four-launch pipeline  enc16 (conv2d .. latent_vector) -> dense1_16 -> mid16
// (ConvT#0 -> ConvT#1) -> tail16; the layer-by-layer launches they replaced
// stay reachable (SRCFD_ENC=0, SRCFD_DENSE1=0, SRCFD_MID=0) for the A/B tests.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "engine.h"
#include "kernels16.h"

namespace srcfd {

#define HIPCHECK(expr)                                                               \
  do {                                                                               \
    hipError_t _e = (expr);                                                          \
    if (_e != hipSuccess) {                                                          \
      set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));           \
      return SRCFD_EHIP;                                                             \
    }                                                                                \
  } while (0)

static const double LOG2E = 1.4426950408889634;

static uint16_t to_bf16(float f) {
  uint32_t u;
  std::memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40);  // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);                                           // round to nearest even
  return (uint16_t)(u >> 16);
}
static uint16_t to_f16(float f) {
  _Float16 hv = (_Float16)f;
  uint16_t r;
  std::memcpy(&r, &hv, 2);
  return r;
}
static uint16_t to16(float f, bool f16) { return f16 ? to_f16(f) : to_bf16(f); }

struct Op16 {
  GemmDesc d;
  size_t w_off = 0;  // elements into Pack16::d_w
  size_t b_off = 0;  // floats into FusedState::d_f32
  int Kpad = 0;
  std::string name;
  int layer = 0;
};

struct Pack16 {  // one per operand type (bf16, f16)
  uint16_t* d_w = nullptr;
  void* d_consts = nullptr;
  void* d_w2f = nullptr;
  void* d_w1f = nullptr;   // mid16: ConvT#1 operands
  uint16_t* d_w0t = nullptr;   // mid16: ConvT#0 weights as the 16 KB LDS images of its stages, per output phase (offsets in w0t_off)
  size_t w0t_off[4] = {0, 0, 0, 0};
  void* d_encf = nullptr;  // enc16: conv2d_1, dense, latent_vector operand fragments (one blob)
  size_t enc_wd_off = 0, enc_wl_off = 0;  // byte offsets of the dense / latent fragments in d_encf
  float* d_encb = nullptr; // enc16: conv2d_1 bias fragments (128 floats)
  float* d_midb = nullptr; // mid16: bias fragments (b0f 128 floats, then b1f 64 floats)
  bool built = false;
};

struct FusedState {
  std::vector<int> cl;  // indices of the 11 compute layers in ModelDesc::layers
  std::vector<Op16> ops;
  std::vector<float> f32;  // conv1 weights [9][64], conv1 bias [64], per-op biases
  size_t c1w_off = 0, c1b_off = 0;
  float* d_f32 = nullptr;
  Pack16 packs[2];
  uint16_t* act[2] = {nullptr, nullptr};
  float* d_part = nullptr;  // split-K partial-sum slabs
  size_t part_elems = 0;
  int cap = 0;
  int t1_buf = 0;  // which act[] holds ConvT#1's output after the last forward
  int num_cus = 256;
  bool enc_ok = false;     // the encoder has the shape enc16 is written for
};

static double scale_in(const ModelDesc& md, const std::vector<int>& cl, int i) {
  return (i > 0 && md.layers[cl[i - 1]].act == SRCFD_ACT_SWISH) ? 1.0 / LOG2E : 1.0;
}
static double scale_out(const ModelDesc& md, const std::vector<int>& cl, int i) {
  return md.layers[cl[i]].act == SRCFD_ACT_SWISH ? LOG2E : 1.0;
}
static double scale_w(const ModelDesc& md, const std::vector<int>& cl, int i) {
  double si = scale_in(md, cl, i), so = scale_out(md, cl, i);
  return (si != 1.0 && so != 1.0) ? 1.0 : si * so;  // swish -> swish: the factors cancel exactly
}

int fused_init(Model& m) {
  FusedState* fs = new FusedState();
  m.fused = fs;
  const ModelDesc& md = m.desc;
  for (size_t i = 0; i < md.layers.size(); ++i)
    if (md.layers[i].kind != SRCFD_LAYER_FLATTEN && md.layers[i].kind != SRCFD_LAYER_RESHAPE) fs->cl.push_back((int)i);
  hipDeviceProp_t prop;
  HIPCHECK(hipGetDeviceProperties(&prop, m.device));
  fs->num_cus = prop.multiProcessorCount;

  // conv1 (VALU kernel): f32 weights, scaled
  {
    const Layer& L = md.layers[fs->cl[0]];
    double sw = scale_w(md, fs->cl, 0), so = scale_out(md, fs->cl, 0);
    fs->c1w_off = fs->f32.size();
    for (float v : L.kernel) fs->f32.push_back((float)(v * sw));
    fs->c1b_off = fs->f32.size();
    for (float v : L.bias) fs->f32.push_back((float)(v * so));
  }
  // GEMM ops: compute layers 1..6 (conv2d_1, dense, latent_vector, dense_1, conv2d_transpose, conv2d_transpose_1)
  for (const Op& op : m.ops) {
    int ci = -1;
    for (size_t k = 0; k < fs->cl.size(); ++k) if (fs->cl[k] == op.layer) ci = (int)k;
    if (ci < 1 || ci > 6) continue;
    Op16 o;
    o.d = op.d;
    o.name = op.name;
    o.layer = ci;
    if (ci == 3) { o.d.N = 64; o.d.CO = 64; o.d.OC = 64; }   // latent 50 -> 64 zero-padded channels
    if (ci == 4) { o.d.CI = 64; o.d.K = 64; }                // dense_1 reads the padded latent
    o.d.Npad = (o.d.N + 63) / 64 * 64;
    o.Kpad = (o.d.K + 63) / 64 * 64;
    if (o.d.CI % 64 != 0 || o.d.K % 64 != 0 || o.d.N % 4 != 0 || o.d.CO % 4 != 0) { set_error("fused path: unsupported channel count in " + op.name); return SRCFD_EINVAL; }
    const double so = scale_out(md, fs->cl, ci);
    while (fs->f32.size() % 4) fs->f32.push_back(0.f);
    o.b_off = fs->f32.size();
    for (int n = 0; n < o.d.Npad; ++n) fs->f32.push_back(n < op.d.N ? (float)(m.pack[op.b_off + n] * so) : 0.f);
    fs->ops.push_back(o);
  }
  if (fs->ops.size() != 9) { set_error("fused path: unexpected plan shape"); return SRCFD_EINVAL; }
  {  // enc16 is written for encoder_10 exactly: 3x3 s1 pad-1 conv 64->128 on 5x5, dense 3200->128, latent 128->64 (padded)
    const GemmDesc& c2 = fs->ops[0].d; const GemmDesc& de = fs->ops[1].d; const GemmDesc& la = fs->ops[2].d;
    const Layer& l0 = md.layers[fs->cl[0]];
    fs->enc_ok = fs->ops[0].layer == 1 && fs->ops[1].layer == 2 && fs->ops[2].layer == 3 && l0.act == SRCFD_ACT_SWISH &&
                 l0.kernel.size() == 9 * 64 && c2.act == SRCFD_ACT_SWISH &&
                 c2.TY == 3 && c2.TX == 3 && c2.CI == 64 && c2.N == 128 && c2.IH == 5 && c2.IW == 5 && c2.MH == 5 && c2.MW == 5 && c2.ay == 1 &&
                 c2.ax == 1 && c2.by == 1 && c2.bx == 1 && c2.cy == -1 && c2.cx == -1 && fs->ops[0].Kpad == 576 &&
                 de.MH == 1 && de.MW == 1 && de.K == 3200 && de.N == 128 && fs->ops[1].Kpad == 3200 &&
                 la.MH == 1 && la.MW == 1 && la.K == 128 && la.N == 64 && fs->ops[2].Kpad == 128;
  }
  HIPCHECK(hipMalloc(&fs->d_f32, fs->f32.size() * sizeof(float)));
  HIPCHECK(hipMemcpy(fs->d_f32, fs->f32.data(), fs->f32.size() * sizeof(float), hipMemcpyHostToDevice));
  return SRCFD_OK;
}

static int build_pack(Model& m, FusedState* fs, bool f16) {
  Pack16& P = fs->packs[f16 ? 1 : 0];
  if (P.built) return SRCFD_OK;
  const ModelDesc& md = m.desc;
  // ---- GEMM weights, transposed: Wt[Npad][Kpad] ----
  std::vector<uint16_t> w;
  size_t oi = 0;
  for (const Op& op : m.ops) {
    int ci = -1;
    for (size_t k = 0; k < fs->cl.size(); ++k) if (fs->cl[k] == op.layer) ci = (int)k;
    if (ci < 1 || ci > 6) continue;
    Op16& o = fs->ops[oi++];
    const double sw = scale_w(md, fs->cl, ci);
    while (w.size() % 8) w.push_back(0);
    o.w_off = w.size();
    w.resize(w.size() + (size_t)o.d.Npad * o.Kpad, 0);
    for (int n = 0; n < op.d.N; ++n)
      for (int k = 0; k < op.d.K; ++k)
        w[o.w_off + (size_t)n * o.Kpad + k] = to16((float)(m.pack[op.w_off + (size_t)k * op.d.Npad + n] * sw), f16);
  }
  HIPCHECK(hipMalloc(&P.d_w, w.size() * sizeof(uint16_t)));
  HIPCHECK(hipMemcpy(P.d_w, w.data(), w.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  if (fs->enc_ok) {
    // ---- enc16 operands: the same 16-bit values, re-ordered so that one lane's MFMA A operand is one 16-byte load ----
    const uint16_t* W2 = w.data() + fs->ops[0].w_off;   // [128][576]
    const uint16_t* WD = w.data() + fs->ops[1].w_off;   // [128][3200]
    const uint16_t* WL = w.data() + fs->ops[2].w_off;   // [64][128]
    std::vector<uint16_t> ef;
    ef.reserve((size_t)(4 * 36 + 8 * 100 + 4 * 4) * 512);
    for (int mt = 0; mt < 4; ++mt)         // 32x32x16: lane l = row l % 32, k = 16 ks + 8 (l / 32) + j
      for (int ks = 0; ks < 36; ++ks)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) ef.push_back(W2[(size_t)(mt * 32 + (l & 31)) * 576 + ks * 16 + (l >> 5) * 8 + j]);
    P.enc_wd_off = ef.size() * 2;
    for (int ft = 0; ft < 8; ++ft)         // 16x16x32: lane l = row l % 16, k = 32 ks + 8 (l / 16) + j
      for (int ks = 0; ks < 100; ++ks)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) ef.push_back(WD[(size_t)(ft * 16 + (l & 15)) * 3200 + ks * 32 + (l >> 4) * 8 + j]);
    P.enc_wl_off = ef.size() * 2;
    for (int ft = 0; ft < 4; ++ft)
      for (int ks = 0; ks < 4; ++ks)
        for (int l = 0; l < 64; ++l)
          for (int j = 0; j < 8; ++j) ef.push_back(WL[(size_t)(ft * 16 + (l & 15)) * 128 + ks * 32 + (l >> 4) * 8 + j]);
    HIPCHECK(hipMalloc(&P.d_encf, ef.size() * sizeof(uint16_t)));
    HIPCHECK(hipMemcpy(P.d_encf, ef.data(), ef.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    std::vector<float> eb(128);
    for (int mt = 0; mt < 4; ++mt)
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) eb[(mt * 2 + hh) * 16 + r] = fs->f32[fs->ops[0].b_off + 32 * mt + (r & 3) + 8 * (r >> 2) + 4 * hh];
    HIPCHECK(hipMalloc(&P.d_encb, eb.size() * sizeof(float)));
    HIPCHECK(hipMemcpy(P.d_encb, eb.data(), eb.size() * sizeof(float), hipMemcpyHostToDevice));
  }

  // ---- tail constants ----
  const Layer& L2 = md.layers[fs->cl[7]];   // ConvT 64->32, kernel (2,2,32,64)
  const Layer& L3 = md.layers[fs->cl[8]];   // ConvT 32->16, kernel (2,2,16,32)
  const Layer& L4 = md.layers[fs->cl[9]];   // ConvT 16->8,  kernel (2,2,8,16)
  const Layer& LO = md.layers[fs->cl[10]];  // Conv 8->1,    kernel (3,3,8,1)
  auto rowof = [](int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; };  // 32x32 accumulator row of register r
  std::vector<uint8_t> cst(TAIL_CONST_BYTES, 0);
  uint16_t* wc = reinterpret_cast<uint16_t*>(cst.data() + TC_OFF_WC);
  for (int kk = 0; kk < 10; ++kk)
    for (int l = 0; l < 64; ++l) {
      int n = l & 15, kg = l >> 4, oy = n >> 3, ox = n & 7;
      int wy = 2 * (kk / 5) + (kg & 1), wx = 2 * (kk % 5) + (kg >> 1);
      int ky = wy - oy, kx = wx - ox;
      for (int c = 0; c < 8; ++c) {
        float v = 0.f;
        if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) v = (float)(LO.kernel[(size_t)(ky * 3 + kx) * 8 + c] / LOG2E);
        wc[((size_t)kk * 64 + l) * 8 + c] = to16(v, f16);
      }
    }
  uint16_t* w3 = reinterpret_cast<uint16_t*>(cst.data() + TC_OFF_W3);
  for (int m3 = 0; m3 < 2; ++m3)
    for (int kk = 0; kk < 2; ++kk)
      for (int l = 0; l < 64; ++l) {
        int i = l & 31, hh = l >> 5, tap = 2 * m3 + (i >> 4), co = i & 15;
        for (int j = 0; j < 8; ++j) {
          int ci = 16 * kk + 8 * hh + j;
          w3[(((size_t)m3 * 2 + kk) * 64 + l) * 8 + j] = to16(L3.kernel[((size_t)tap * 16 + co) * 32 + ci], f16);
        }
      }
  uint16_t* w4 = reinterpret_cast<uint16_t*>(cst.data() + TC_OFF_W4);
  for (int l = 0; l < 64; ++l) {
    int i = l & 31, hh = l >> 5, tap = i >> 3, co = i & 7;
    for (int j = 0; j < 8; ++j) {
      int ci = (j & 3) + 8 * (j >> 2) + 4 * hh;  // k order of an accumulator used as the next B operand
      w4[(size_t)l * 8 + j] = to16(L4.kernel[((size_t)tap * 8 + co) * 16 + ci], f16);
    }
  }
  float* b2 = reinterpret_cast<float*>(cst.data() + TC_OFF_B2);
  float* b3 = reinterpret_cast<float*>(cst.data() + TC_OFF_B3);
  float* b4 = reinterpret_cast<float*>(cst.data() + TC_OFF_B4);
  for (int hh = 0; hh < 2; ++hh)
    for (int r = 0; r < 16; ++r) {
      int row = rowof(r, hh);
      b2[hh * 16 + r] = (float)(L2.bias[row] * LOG2E);
      b3[hh * 16 + r] = (float)(L3.bias[row & 15] * LOG2E);
      b4[hh * 16 + r] = (float)(L4.bias[row & 7] * LOG2E);
    }
  *reinterpret_cast<float*>(cst.data() + TC_OFF_BC) = LO.bias[0];
  HIPCHECK(hipMalloc(&P.d_consts, cst.size()));
  HIPCHECK(hipMemcpy(P.d_consts, cst.data(), cst.size(), hipMemcpyHostToDevice));

  std::vector<uint16_t> w2((size_t)4 * 4 * 64 * 8);
  for (int mt = 0; mt < 4; ++mt)
    for (int kk = 0; kk < 4; ++kk)
      for (int l = 0; l < 64; ++l) {
        int co = l & 31, hh = l >> 5;
        for (int j = 0; j < 8; ++j) {
          int ci = 16 * kk + 8 * hh + j;
          w2[(((size_t)mt * 4 + kk) * 64 + l) * 8 + j] = to16(L2.kernel[((size_t)mt * 32 + co) * 64 + ci], f16);
        }
      }
  HIPCHECK(hipMalloc(&P.d_w2f, w2.size() * sizeof(uint16_t)));
  HIPCHECK(hipMemcpy(P.d_w2f, w2.data(), w2.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  // ---- mid16 (ConvT#0 -> ConvT#1) operands ----
  {
    const Layer& L0 = md.layers[fs->cl[5]];  // ConvT 256->128 (bias only; weights reuse the per-phase GEMM packing)
    const Layer& L1 = md.layers[fs->cl[6]];  // ConvT 128->64, kernel (2,2,64,128)
    std::vector<uint16_t> w1((size_t)8 * 8 * 64 * 8);
    for (int j8 = 0; j8 < 8; ++j8)
      for (int st = 0; st < 8; ++st)
        for (int l = 0; l < 64; ++l) {
          int i = l & 31, hh = l >> 5, tap = j8 >> 1, co = 32 * (j8 & 1) + i;
          for (int j = 0; j < 8; ++j) {
            // k-step st consumes accumulator tile st>>1, registers 8*(st&1)+j: channel of that register
            int ci = 32 * (st >> 1) + 16 * (st & 1) + 8 * (j >> 2) + 4 * hh + (j & 3);
            w1[(((size_t)j8 * 8 + st) * 64 + l) * 8 + j] = to16(L1.kernel[((size_t)tap * 64 + co) * 128 + ci], f16);
          }
        }
    // ConvT#0 weights, stage by stage, in the order the kernel's LDS tile holds them: stage (chunk c, tap t) of a phase = [128 rows][64 k]
    // = 1024 sixteen-byte pieces, piece row * 8 + slot holding k-piece slot ^ ((row >> 1) & 7) (the bank swizzle of the fragment reads).
    // A tile is then 16 KB of CONSECUTIVE memory.  Read from the GEMM layout Wt[128][Kpad] instead, its 128 row segments lie Kpad * 2 =
    // 512 / 1024 / 2048 bytes apart -- powers of two: every workgroup of a phase asks the same one or two L2 channels for the same tile
    // at the same time (round 3: the tile loads' cost did not hide behind anything, whatever the prefetch depth).
    {
      std::vector<uint16_t> wt;
      int ph = 0;
      for (const Op16& o : fs->ops) {
        if (o.layer != 5 || ph >= 4) continue;
        const int NT = o.d.K / 256;
        P.w0t_off[ph++] = wt.size();
        const uint16_t* W = w.data() + o.w_off;
        for (int st = 0; st < 4 * NT; ++st) {
          const int c = st / NT, t = st - c * NT;
          for (int i = 0; i < 1024; ++i) {
            const int row = i >> 3, kc = (i & 7) ^ ((row >> 1) & 7);
            for (int j = 0; j < 8; ++j) wt.push_back(W[(size_t)row * o.Kpad + t * 256 + c * 64 + kc * 8 + j]);
          }
        }
      }
      if (!wt.empty()) {
        HIPCHECK(hipMalloc(&P.d_w0t, wt.size() * sizeof(uint16_t)));
        HIPCHECK(hipMemcpy(P.d_w0t, wt.data(), wt.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
      }
    }
    HIPCHECK(hipMalloc(&P.d_w1f, w1.size() * sizeof(uint16_t)));
    HIPCHECK(hipMemcpy(P.d_w1f, w1.data(), w1.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
    std::vector<float> mb(128 + 64);
    for (int mt = 0; mt < 4; ++mt)
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) mb[(mt * 2 + hh) * 16 + r] = (float)(L0.bias[32 * mt + rowof(r, hh)] * LOG2E);
    for (int jj = 0; jj < 2; ++jj)
      for (int hh = 0; hh < 2; ++hh)
        for (int r = 0; r < 16; ++r) mb[128 + (jj * 2 + hh) * 16 + r] = (float)(L1.bias[32 * jj + rowof(r, hh)] * LOG2E);
    HIPCHECK(hipMalloc(&P.d_midb, mb.size() * sizeof(float)));
    HIPCHECK(hipMemcpy(P.d_midb, mb.data(), mb.size() * sizeof(float), hipMemcpyHostToDevice));
  }
  P.built = true;
  return SRCFD_OK;
}

void fused_free(Model& m) {
  FusedState* fs = m.fused;
  if (!fs) return;
  for (auto& P : fs->packs) {
    if (P.d_w) (void)hipFree(P.d_w);
    if (P.d_consts) (void)hipFree(P.d_consts);
    if (P.d_w2f) (void)hipFree(P.d_w2f);
    if (P.d_w1f) (void)hipFree(P.d_w1f);
    if (P.d_w0t) (void)hipFree(P.d_w0t);
    if (P.d_midb) (void)hipFree(P.d_midb);
    if (P.d_encf) (void)hipFree(P.d_encf);
    if (P.d_encb) (void)hipFree(P.d_encb);
  }
  if (fs->d_f32) (void)hipFree(fs->d_f32);
  for (auto* b : fs->act) if (b) (void)hipFree(b);
  if (fs->d_part) (void)hipFree(fs->d_part);
  delete fs;
  m.fused = nullptr;
}

int fused_debug_read(Model& m, int index, void* dst, size_t bytes) {
  FusedState* fs = m.fused;
  if (!fs || !fs->act[index]) { set_error("no fused activations yet"); return SRCFD_EINVAL; }
  if (bytes > (size_t)fs->cap * 160000 * sizeof(uint16_t)) { set_error("read past the activation buffer"); return SRCFD_EINVAL; }
  HIPCHECK(hipSetDevice(m.device));
  HIPCHECK(hipDeviceSynchronize());
  HIPCHECK(hipMemcpy(dst, fs->act[index == 0 ? fs->t1_buf : fs->t1_buf ^ 1], bytes, hipMemcpyDeviceToHost));
  return SRCFD_OK;
}

static const size_t ACT_ELEMS = 160000;  // largest inter-kernel activation per sample: (50,50,64)

// Everything the 16-bit forward of an n-sample batch allocates or packs lazily: the operand packs of the current operand
// type and the two activation buffers (+ split-K slabs).  Called by the forward itself and by srcfd_model_reserve.
int fused_reserve(Model& m, int n) {
  FusedState* fs = m.fused;
  if (!fs) { set_error("fused path not initialised"); return SRCFD_EINVAL; }
  const bool f16 = m.precision == SRCFD_PREC_F16;
  int rc = build_pack(m, fs, f16);
  if (rc) return rc;
  const int want = std::min(n, 1024);
  if (want > fs->cap) {
    m.drop_graph();  // a captured forward holds the old buffers' addresses
    for (auto*& b : fs->act) if (b) { HIPCHECK(hipFree(b)); b = nullptr; }
    fs->cap = 0;
    for (auto*& b : fs->act) HIPCHECK(hipMalloc(&b, (size_t)want * ACT_ELEMS * sizeof(uint16_t)));
    if (fs->d_part) { HIPCHECK(hipFree(fs->d_part)); fs->d_part = nullptr; }
    fs->part_elems = (size_t)16 * want * 128;  // dense(3200->128): up to 16 K-slice slabs of (rows x 128) f32
    HIPCHECK(hipMalloc(&fs->d_part, fs->part_elems * sizeof(float)));
    fs->cap = want;
  }
  return SRCFD_OK;
}

int fused_forward(Model& m, const float* x_dev, int n, const float* aff_in, const float* aff_out, void* y_dev, int out_dtype, int flags,
                  unsigned long long* nonfinite, hipStream_t s) {
  FusedState* fs = m.fused;
  if (!fs) { set_error("fused path not initialised"); return SRCFD_EINVAL; }
  const bool f16 = m.precision == SRCFD_PREC_F16;
  int rc = fused_reserve(m, n);
  if (rc) return rc;
  const Pack16& P = fs->packs[f16 ? 1 : 0];
  const size_t osz = out_dtype == SRCFD_F32 ? 4 : 2;
  for (int i0 = 0; i0 < n; i0 += fs->cap) {
    const int c = std::min(fs->cap, n - i0);
    const float* xin = x_dev + (size_t)i0 * 100;
    const float* ain = aff_in ? aff_in + 2 * (size_t)i0 : nullptr;
    const float* aout = aff_out ? aff_out + 2 * (size_t)i0 : nullptr;
    int cur = 0;
    // functional A/B switches of the tests (both implementations of the encoder, of dense_1 and of the network's middle are complete):
    // read once per call by Model::predict_device (Switches, engine.h), part of the hipGraph key
    const bool use_enc = fs->enc_ok && m.sw.enc16;   // false: layer-by-layer encoder
    const bool use_mid = m.sw.mid16;                 // false: generic GEMMs
    const bool use_d1 = m.sw.dense1_16;              // false: dense_1 on the generic GEMM
    if (use_enc) {
      EncParams ep;
      ep.x = xin; ep.affine = ain; ep.n = c;
      ep.w1 = fs->d_f32 + fs->c1w_off; ep.b1 = fs->d_f32 + fs->c1b_off;
      ep.w2f = P.d_encf; ep.b2f = P.d_encb;
      ep.wdf = (const char*)P.d_encf + P.enc_wd_off; ep.bd = fs->d_f32 + fs->ops[1].b_off;
      ep.wlf = (const char*)P.d_encf + P.enc_wl_off; ep.bl = fs->d_f32 + fs->ops[2].b_off;
      ep.z = fs->act[1];
      ep.act_dense = fs->ops[1].d.act; ep.act_latent = fs->ops[2].d.act;
      ep.prof = nullptr;
#ifdef SRCFD_DIAG
      static unsigned long long* d_eprof = nullptr;
      static int eprof_calls = 0;
      static const bool eprof = getenv("SRCFD_ENC_PROF") != nullptr;
      if (eprof && !d_eprof) HIPCHECK(hipMalloc(&d_eprof, 64 * sizeof(unsigned long long)));
      ep.prof = eprof ? d_eprof : nullptr;
#endif
      rc = m.launch("encoder(conv2d..latent_vector)", s, [&] { return launch_enc16(f16, ep, s); });
      if (rc) return rc;
#ifdef SRCFD_DIAG
      if (eprof && ++eprof_calls == 20) {
        unsigned long long hbuf[64];
        HIPCHECK(hipStreamSynchronize(s));
        HIPCHECK(hipMemcpy(hbuf, d_eprof, sizeof(hbuf), hipMemcpyDeviceToHost));
        fprintf(stderr, "enc16 workgroup 7, s_memtime ticks since entry: staged, conv2d, conv2d_1 MFMA, A2 ready, dense, end\n");
        for (int w = 0; w < 8; ++w)
          fprintf(stderr, "  wave %d: %6llu %6llu %6llu %6llu %6llu %6llu\n", w, hbuf[w * 8 + 1], hbuf[w * 8 + 2], hbuf[w * 8 + 3], hbuf[w * 8 + 4], hbuf[w * 8 + 5], hbuf[w * 8 + 6]);
      }
#endif
      cur = 1;   // where the layer-by-layer chain leaves the latent vectors, too
    } else {
      rc = m.launch("conv2d", s, [&] { return launch_enc_conv1_16(f16, xin, ain, fs->d_f32 + fs->c1w_off, fs->d_f32 + fs->c1b_off, fs->act[0], c, s); });
      if (rc) return rc;
    }
    int prev_layer = -1;
    for (const Op16& o : fs->ops) {
      if (use_enc && o.layer < 4) continue;  // conv2d_1, dense, latent_vector ran inside enc16
      if (use_mid && o.layer >= 5) break;  // ConvT#0 / ConvT#1 run in the fused mid kernel below
      if (o.layer != prev_layer && prev_layer >= 0) cur ^= 1;
      prev_layer = o.layer;
      GemmDesc d = o.d;
      d.M = c * d.MH * d.MW;
      const uint16_t* X = fs->act[cur];
      uint16_t* Y = fs->act[cur ^ 1];
      // dense layers with few rows and a long K: split K over workgroups (f32 slabs + finish kernel)
      int splits = 1;
      if (d.MH == 1 && d.MW == 1 && d.K >= 1024) splits = std::max(1, std::min(16, d.K / 256));
      if (splits > 1 && (size_t)splits * d.M * d.Npad > fs->part_elems) splits = 1;
      if (use_d1 && o.layer == 4 && dense1_16_qualifies(d, o.Kpad))
        rc = m.launch(o.name.c_str(), s, [&] { return launch_dense1_16(f16, d, X, P.d_w + o.w_off, fs->d_f32 + o.b_off, Y, s); });
      else
        rc = m.launch(o.name.c_str(), s, [&] { return launch_gemm16(f16, d, X, P.d_w + o.w_off, o.Kpad, fs->d_f32 + o.b_off, Y, fs->d_part, splits, s); });
      if (rc) return rc;
    }
    if (use_mid) {
      cur ^= 1;  // dense_1 output
      MidParams mp;
      mp.in = fs->act[cur];
      mp.out = fs->act[cur ^ 1];
      mp.n = c;
      int ph = 0;
      for (const Op16& o : fs->ops)
        if (o.layer == 5) { mp.w0[ph] = P.d_w + o.w_off; mp.kpad[ph] = o.Kpad; mp.w0t[ph] = P.d_w0t + P.w0t_off[ph]; ++ph; }
      mp.b0f = P.d_midb;
      mp.w1f = P.d_w1f;
      mp.b1f = P.d_midb + 128;
      mp.ablate = 0;
      mp.order = m.sw.mid_order;
      mp.prof = nullptr;
#ifdef SRCFD_DIAG
      { static const int abl = [] { const char* e = getenv("SRCFD_MID_ABLATE"); return e ? atoi(e) : 0; }(); mp.ablate = abl; }
      static unsigned long long* d_mprof = nullptr;   // SRCFD_MID_PROF=1: section cycle sums of wave 0 of every workgroup, per output phase
      static int mprof_calls = 0;
      static const bool mprof = getenv("SRCFD_MID_PROF") != nullptr;
      if (mprof && !d_mprof) { HIPCHECK(hipMalloc(&d_mprof, 60 * sizeof(unsigned long long))); }
      if (mprof) { HIPCHECK(hipMemsetAsync(d_mprof, 0, 60 * sizeof(unsigned long long), s)); mp.prof = d_mprof; }
#endif
      const int mid_waves = m.sw.mid_waves ? m.sw.mid_waves : (m.sw.mid_shape == 1 ? 8 : m.sw.mid_shape == 2 ? 82 : 42);
      rc = m.launch("mid(convT0+convT1)", s, [&] { return launch_mid16(f16, mp, mid_waves, s); });
      if (rc) return rc;
#ifdef SRCFD_DIAG
      if (mprof && ++mprof_calls == 20) {
        unsigned long long hb[60];
        HIPCHECK(hipStreamSynchronize(s));
        HIPCHECK(hipMemcpy(hb, d_mprof, sizeof(hb), hipMemcpyDeviceToHost));
        fprintf(stderr, "mid16, wave 0 of every workgroup, mean cycles per workgroup by output phase: workgroups | entry->tables | ->first stage ready | main loop | ConvT#0 swish | ConvT#1 stage\n");
        for (int ph = 0; ph < 4; ++ph) {
          const double nwg = (double)std::max<unsigned long long>(hb[ph * 6], 1);
          fprintf(stderr, "  phase %d: %6llu | %7.0f | %7.0f | %7.0f | %7.0f | %7.0f    main loop = sync %7.0f + issue %7.0f + fragments/MFMA %7.0f\n", ph, hb[ph * 6], hb[ph * 6 + 1] / nwg, hb[ph * 6 + 2] / nwg, hb[ph * 6 + 3] / nwg,
                  hb[ph * 6 + 4] / nwg, hb[ph * 6 + 5] / nwg, hb[24 + ph * 3] / nwg, hb[24 + ph * 3 + 1] / nwg, hb[24 + ph * 3 + 2] / nwg);
        }
        fprintf(stderr, "  workgroup 3 of phase 0, per wave, main loop: sync | issue | fragments/MFMA\n");
        for (int w = 0; w < 8; ++w) fprintf(stderr, "    wave %d: %7llu | %7llu | %7llu\n", w, hb[36 + w * 3], hb[36 + w * 3 + 1], hb[36 + w * 3 + 2]);
      }
#endif
    }
    cur ^= 1;
    fs->t1_buf = cur;
    TailParams tp;
    tp.in = fs->act[cur];
    tp.out = (char*)y_dev + (size_t)i0 * 160000 * osz;
    tp.n = c;
    tp.consts = P.d_consts;
    tp.w2frags = P.d_w2f;
    tp.aff_out = aout;
    tp.nan_guard = flags & SRCFD_FLAG_NAN_GUARD;
    tp.nonfinite = nonfinite;
    tp.out_dtype = out_dtype;
    tp.ablate = 0;
    tp.prof = nullptr;
#ifdef SRCFD_DIAG
    { static const int abl = [] { const char* e = getenv("SRCFD_TAIL_ABLATE"); return e ? atoi(e) : 0; }(); tp.ablate = abl; }
    static unsigned long long* d_prof = nullptr;   // SRCFD_TAIL_PROF=1: per-wave section timers of workgroup 0 (synchronises: never under graph capture)
    static int prof_calls = 0;
    static const bool prof = getenv("SRCFD_TAIL_PROF") != nullptr;
    if (prof && !d_prof) HIPCHECK(hipMalloc(&d_prof, 128 * sizeof(unsigned long long)));
    tp.prof = prof ? d_prof : nullptr;
#endif
    // Batches that do not fill the chip evenly (fewer samples than CUs, or a few more than a multiple of them): cut each
    // sample into S segments so that the longest workgroup walks fewer strips.  Cost of a choice = strips walked by the
    // busiest workgroup: ceil(n S / CUs) virtual samples of 50/S (+1 warm-up) strips, + 2 rounds of pipeline depth.
    int seg = 1;
    if (m.sw.tail_seg) seg = m.sw.tail_seg;   // SRCFD_TAIL_SEG (tests, tools): read per call, reported by srcfd_model_last_plan
    else {
      long best = ((long)(c + fs->num_cus - 1) / fs->num_cus) * 50 + 2;
      for (int cand : {2, 5, 10, 25}) {
        long cost = ((long)((long)c * cand + fs->num_cus - 1) / fs->num_cus) * (50 / cand + 1) + 2;
        if (cost * 115 < best * 100) { best = cost; seg = cand; }  // warm-up strips and extra workgroups are not free: ask for 15 %
      }
    }
    if (seg != 1 && seg != 2 && seg != 5 && seg != 10 && seg != 25) seg = 1;
    tp.seg = seg;
    m.plan.tail_seg = seg;
    const int blocks = std::min(c * seg, fs->num_cus);
    const bool tail_s = m.sw.tail16s;
    rc = m.launch("tail(convT2-4+out)", s, [&] { return tail_s ? launch_tail16s(f16, tp, blocks, s) : launch_tail16(f16, tp, blocks, s); });
    if (rc) return rc;
#ifdef SRCFD_DIAG
    if (prof && ++prof_calls == 20) {
      unsigned long long h[128];
      HIPCHECK(hipStreamSynchronize(s));
      HIPCHECK(hipMemcpy(h, d_prof, sizeof(h), hipMemcpyDeviceToHost));
      if (tail_s) {
        fprintf(stderr, "tail16s workgroup 0, per wave: fast rounds (work cycles, barrier wait cycles, count) | other rounds (work, wait, count)\n");
        for (int w = 0; w < 8; ++w)
          fprintf(stderr, "  wave %d: %9llu %9llu %5llu | %9llu %9llu %5llu   per fast round: work %6.0f wait %6.0f\n", w, h[w * 6], h[w * 6 + 1], h[w * 6 + 2],
                  h[w * 6 + 3], h[w * 6 + 4], h[w * 6 + 5], h[w * 6 + 2] ? (double)h[w * 6] / h[w * 6 + 2] : 0.0, h[w * 6 + 2] ? (double)h[w * 6 + 1] / h[w * 6 + 2] : 0.0);
        fprintf(stderr, "  cycles per fast round and section (top, then the blocks in issue order):\n");
        for (int w = 0; w < 8; ++w) {
          fprintf(stderr, "  wave %d:", w);
          for (int i = 0; i < 10; ++i) fprintf(stderr, " %6.0f", h[w * 6 + 2] ? (double)h[48 + w * 10 + i] / h[w * 6 + 2] : 0.0);
          fprintf(stderr, "\n");
        }
      } else {
        fprintf(stderr, "tail16 workgroup 0, cycles per wave: D, BC, A, barrier wait, total\n");
        for (int w = 0; w < 16; ++w)
          fprintf(stderr, "  wave %2d: %9llu %9llu %9llu %9llu %9llu\n", w, h[w * 5], h[w * 5 + 1], h[w * 5 + 2], h[w * 5 + 3], h[w * 5 + 4]);
      }
    }
#endif
  }
  return SRCFD_OK;
}

}  // namespace srcfd
