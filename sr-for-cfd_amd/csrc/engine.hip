// srcfd engine: builds the kernel plan for a layer graph, owns device weights
// and workspaces, and exports the model part of the C ABI (include/srcfd.h).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "engine.h"

namespace srcfd {

thread_local std::string g_last_error;
void set_error(const std::string& m) { g_last_error = m; }

#define HIPCHECK(expr)                                                                              \
  do {                                                                                              \
    hipError_t _e = (expr);                                                                         \
    if (_e != hipSuccess) {                                                                         \
      set_error(std::string(#expr) + " failed: " + hipGetErrorString(_e));                          \
      return SRCFD_EHIP;                                                                            \
    }                                                                                               \
  } while (0)

static int round_up(int v, int m) { return (v + m - 1) / m * m; }

// ---------------------------------------------------------------------------
// plan
// ---------------------------------------------------------------------------
static void pack_B(std::vector<float>& pack, Op& op, const std::vector<float>& Bmat /*[K][N]*/, const std::vector<float>& bias_n) {
  const GemmDesc& d = op.d;
  while (pack.size() % 64) pack.push_back(0.f);  // 256-byte aligned sub-buffers
  op.w_off = pack.size();
  pack.resize(pack.size() + (size_t)std::max(d.K, 1) * d.Npad, 0.f);
  for (int k = 0; k < d.K; ++k)
    std::memcpy(&pack[op.w_off + (size_t)k * d.Npad], &Bmat[(size_t)k * d.N], sizeof(float) * d.N);
  while (pack.size() % 64) pack.push_back(0.f);
  op.b_off = pack.size();
  pack.resize(pack.size() + d.Npad, 0.f);
  std::memcpy(&pack[op.b_off], bias_n.data(), sizeof(float) * d.N);
}

void build_plan(const ModelDesc& desc, std::vector<Op>& ops, std::vector<float>& pack) {
  ops.clear();
  pack.clear();
  for (size_t li = 0; li < desc.layers.size(); ++li) {
    const Layer& L = desc.layers[li];
    if (L.kind == SRCFD_LAYER_FLATTEN || L.kind == SRCFD_LAYER_RESHAPE) continue;  // views of NHWC buffers
    const int IH = L.in_shape[0], IW = L.in_shape[1], OH = L.out_shape[0], OW = L.out_shape[1];
    GemmDesc d{};
    d.act = L.act;
    d.OH = OH; d.OW = OW; d.OC = L.cout; d.CO = L.cout;
    d.nphx = 1; d.os = 1;
    d.IH = IH; d.IW = IW; d.CI = L.cin;
    if (L.kind == SRCFD_LAYER_DENSE) {
      d.IH = d.IW = 1; d.CI = L.cin; d.OH = d.OW = 1;
      d.MH = d.MW = 1; d.TY = d.TX = 1;
      d.K = L.cin; d.N = L.cout; d.Npad = round_up(d.N, 32);
      Op op; op.d = d; op.layer = (int)li; op.name = L.name;
      pack_B(pack, op, L.kernel, L.bias);
      ops.push_back(op);
    } else if (L.kind == SRCFD_LAYER_CONV2D) {
      int pt = 0, pl = 0;
      if (L.same) {
        int th = std::max((OH - 1) * L.stride + L.kh - IH, 0), tw = std::max((OW - 1) * L.stride + L.kw - IW, 0);
        pt = th / 2; pl = tw / 2;  // TF SAME: the extra pixel goes after
      }
      d.MH = OH; d.MW = OW; d.TY = L.kh; d.TX = L.kw;
      d.ay = d.ax = L.stride; d.by = d.bx = 1; d.cy = -pt; d.cx = -pl;
      d.K = L.kh * L.kw * L.cin; d.N = L.cout; d.Npad = round_up(d.N, 32);
      Op op; op.d = d; op.layer = (int)li; op.name = L.name;
      pack_B(pack, op, L.kernel, L.bias);  // (kh,kw,Cin,Cout) is already [K][N]
      ops.push_back(op);
    } else {  // Conv2DTranspose, VALID, kernel (kh,kw,Cout,Cin)
      const int s = L.stride;
      auto W = [&](int a, int b, int co, int ci) { return L.kernel[(((size_t)a * L.kw + b) * L.cout + co) * L.cin + ci]; };
      if (L.kh == s && L.kw == s) {
        d.MH = IH; d.MW = IW; d.TY = d.TX = 1;
        d.ay = d.ax = 1; d.by = d.bx = 0; d.cy = d.cx = 0;
        d.K = L.cin; d.N = s * s * L.cout; d.Npad = round_up(d.N, 32);
        d.nphx = s; d.os = s;
        std::vector<float> B((size_t)d.K * d.N), bn(d.N);
        for (int ci = 0; ci < L.cin; ++ci)
          for (int py = 0; py < s; ++py)
            for (int px = 0; px < s; ++px)
              for (int co = 0; co < L.cout; ++co) B[(size_t)ci * d.N + (py * s + px) * L.cout + co] = W(py, px, co, ci);
        for (int n = 0; n < d.N; ++n) bn[n] = L.bias[n % L.cout];
        Op op; op.d = d; op.layer = (int)li; op.name = L.name;
        pack_B(pack, op, B, bn);
        ops.push_back(op);
      } else {
        for (int py = 0; py < s; ++py)
          for (int px = 0; px < s; ++px) {
            GemmDesc p = d;
            p.TY = py < L.kh ? (L.kh - py + s - 1) / s : 0;
            p.TX = px < L.kw ? (L.kw - px + s - 1) / s : 0;
            p.MH = py < OH ? (OH - py + s - 1) / s : 0;
            p.MW = px < OW ? (OW - px + s - 1) / s : 0;
            if (p.MH == 0 || p.MW == 0) continue;
            p.ay = p.ax = 1; p.by = p.bx = -1; p.cy = p.cx = 0;
            p.K = p.TY * p.TX * L.cin; p.N = L.cout; p.Npad = round_up(p.N, 32);
            p.os = s; p.oy0 = py; p.ox0 = px;
            std::vector<float> B((size_t)std::max(p.K, 1) * p.N, 0.f);
            for (int ty = 0; ty < p.TY; ++ty)
              for (int tx = 0; tx < p.TX; ++tx)
                for (int ci = 0; ci < L.cin; ++ci)
                  for (int co = 0; co < L.cout; ++co)
                    B[((size_t)(ty * p.TX + tx) * L.cin + ci) * p.N + co] = W(py + s * ty, px + s * tx, co, ci);
            Op op; op.d = p; op.layer = (int)li;
            op.name = L.name + ".ph" + std::to_string(py) + std::to_string(px);
            pack_B(pack, op, B, L.bias);
            ops.push_back(op);
          }
      }
    }
  }
}

// Operands of the fused ConvT pair (kernels_fp32.hip, convt_pair_f32), appended to `pack`:
//   wa[(T*16 + s)*64 + lane]: first layer, row i = lane & 31 of tile T -> tap 2T + (i >> 4), channel i & 15; k = 16 (lane >> 5) + s
//   wb[u*64 + lane]:          second layer, row j = lane & 31 -> tap j >> 3, channel j & 7; k = (u & 3) + 8 (u >> 2) + 4 (lane >> 5)
// Conv2DTranspose kernels are (kh, kw, Cout, Cin).
static void plan_convt_pair(Model& m) {
  m.pair_op = -1;
  for (size_t i = 0; i + 1 < m.ops.size(); ++i) {
    const Op &a = m.ops[i], &b = m.ops[i + 1];
    const Layer &La = m.desc.layers[a.layer], &Lb = m.desc.layers[b.layer];
    auto is_k2s2 = [](const Layer& L) { return L.kind == SRCFD_LAYER_CONV2D_TRANSPOSE && L.kh == 2 && L.kw == 2 && L.stride == 2; };
    if (a.layer + 1 != b.layer || !is_k2s2(La) || !is_k2s2(Lb)) continue;
    if (a.d.nphx != 2 || b.d.nphx != 2) continue;  // one op per layer (the merged-phase form)
    if (La.cin != 32 || La.cout != 16 || Lb.cin != 16 || Lb.cout != 8) continue;
    auto& pk = m.pack;
    auto align = [&]() { while (pk.size() % 64) pk.push_back(0.f); };
    align(); m.pair_wa = pk.size(); pk.resize(pk.size() + 2 * 16 * 64);
    for (int T = 0; T < 2; ++T)
      for (int s = 0; s < 16; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = lane & 31, tap = 2 * T + (row >> 4), ch = row & 15, k = 16 * (lane >> 5) + s;
          pk[m.pair_wa + (size_t)(T * 16 + s) * 64 + lane] = La.kernel[((size_t)tap * 16 + ch) * 32 + k];
        }
    align(); m.pair_ba = pk.size(); pk.insert(pk.end(), La.bias.begin(), La.bias.end());
    align(); m.pair_wb = pk.size(); pk.resize(pk.size() + 8 * 64);
    for (int u = 0; u < 8; ++u)
      for (int lane = 0; lane < 64; ++lane) {
        const int row = lane & 31, tap = row >> 3, ch = row & 7, k = (u & 3) + 8 * (u >> 2) + 4 * (lane >> 5);
        pk[m.pair_wb + (size_t)u * 64 + lane] = Lb.kernel[((size_t)tap * 8 + ch) * 16 + k];
      }
    align(); m.pair_bb = pk.size(); pk.insert(pk.end(), Lb.bias.begin(), Lb.bias.end());
    align();
    m.pair_op = (int)i;
    return;
  }
}

// Operands of the three-layer chain (convt_triple_f32): w1[(T*32 + s)*64 + lane]: tap T, channel lane & 31, k = 32 (lane >> 5) + s;
// w2[(T*16 + u)*64 + lane]: tap 2T + (row >> 4), channel row & 15, k = (u & 3) + 8 (u >> 2) + 4 (lane >> 5); w3 as the pair's wb.
static void plan_convt_triple(Model& m) {
  m.triple_op = -1;
  auto is_k2s2 = [](const Layer& L) { return L.kind == SRCFD_LAYER_CONV2D_TRANSPOSE && L.kh == 2 && L.kw == 2 && L.stride == 2; };
  for (size_t i = 0; i + 2 < m.ops.size(); ++i) {
    const Op &a = m.ops[i], &b = m.ops[i + 1], &c = m.ops[i + 2];
    if (a.layer + 1 != b.layer || b.layer + 1 != c.layer) continue;
    const Layer &L1 = m.desc.layers[a.layer], &L2 = m.desc.layers[b.layer], &L3 = m.desc.layers[c.layer];
    if (!is_k2s2(L1) || !is_k2s2(L2) || !is_k2s2(L3) || a.d.nphx != 2 || b.d.nphx != 2 || c.d.nphx != 2) continue;
    if (L1.cin != 64 || L1.cout != 32 || L2.cin != 32 || L2.cout != 16 || L3.cin != 16 || L3.cout != 8) continue;
    auto& pk = m.pack;
    auto align = [&]() { while (pk.size() % 64) pk.push_back(0.f); };
    align(); m.tri_w1 = pk.size(); pk.resize(pk.size() + 4 * 32 * 64);
    for (int T = 0; T < 4; ++T)
      for (int s = 0; s < 32; ++s)
        for (int lane = 0; lane < 64; ++lane)
          pk[m.tri_w1 + (size_t)(T * 32 + s) * 64 + lane] = L1.kernel[((size_t)T * 32 + (lane & 31)) * 64 + 32 * (lane >> 5) + s];
    align(); m.tri_b1 = pk.size(); pk.insert(pk.end(), L1.bias.begin(), L1.bias.end());
    align(); m.tri_w2 = pk.size(); pk.resize(pk.size() + 2 * 16 * 64);
    for (int T = 0; T < 2; ++T)
      for (int u = 0; u < 16; ++u)
        for (int lane = 0; lane < 64; ++lane) {
          const int row = lane & 31, tap = 2 * T + (row >> 4), ch = row & 15, k = (u & 3) + 8 * (u >> 2) + 4 * (lane >> 5);
          pk[m.tri_w2 + (size_t)(T * 16 + u) * 64 + lane] = L2.kernel[((size_t)tap * 16 + ch) * 32 + k];
        }
    align(); m.tri_b2 = pk.size(); pk.insert(pk.end(), L2.bias.begin(), L2.bias.end());
    align(); m.tri_w3 = pk.size(); pk.resize(pk.size() + 8 * 64);
    for (int u = 0; u < 8; ++u)
      for (int lane = 0; lane < 64; ++lane) {
        const int row = lane & 31, tap = row >> 3, ch = row & 7, k = (u & 3) + 8 * (u >> 2) + 4 * (lane >> 5);
        pk[m.tri_w3 + (size_t)u * 64 + lane] = L3.kernel[((size_t)tap * 8 + ch) * 16 + k];
      }
    align(); m.tri_b3 = pk.size(); pk.insert(pk.end(), L3.bias.begin(), L3.bias.end());
    align();
    m.triple_op = (int)i;
    return;
  }
}

// Operands of the fused f32 tail (kernels_tail32.hip): the ConvT 64 -> 32 -> 16 -> 8 chain + the 3x3 SAME conv 8 -> 1 that
// ends the network, v_mfma_f32_16x16x4_f32 fragments (lane = (m = lane & 15, kg = lane >> 4)):
//   w1[((tap1*2 + t)*16 + s)*64 + lane] = W1[tap1][co 16t + m][ci 16kg + s]
//   w2[(tap2*8 + 4t + i)*64 + lane]     = W2[tap2][co m][ci 16t + 4kg + i]      (k order = the first layer's accumulator order)
//   w3[(u*4 + i)*64 + lane]             = W3[tap3 2u + (m >> 3)][co m & 7][ci 4kg + i]
// Conv2DTranspose kernels are (kh, kw, Cout, Cin); the Conv2D kernel (3, 3, 8, 1) is already (ty, tx, ci).
static void plan_tail32(Model& m) {
  m.tail32_op = -1;
  if (m.triple_op < 0 || (size_t)m.triple_op + 4 != m.ops.size()) return;
  const size_t i = (size_t)m.triple_op;
  const Op& last = m.ops[i + 3];
  if (last.layer != m.ops[i + 2].layer + 1) return;
  const Layer &L1 = m.desc.layers[m.ops[i].layer], &L2 = m.desc.layers[m.ops[i + 1].layer], &L3 = m.desc.layers[m.ops[i + 2].layer];
  const Layer& LO = m.desc.layers[last.layer];
  if (LO.kind != SRCFD_LAYER_CONV2D || LO.kh != 3 || LO.kw != 3 || LO.stride != 1 || !LO.same || LO.cin != 8 || LO.cout != 1) return;
  if (m.ops[i].d.MW > 50) return;  // the LDS ring holds rows of up to 400 pixels
  if (L1.act != SRCFD_ACT_SWISH || L2.act != SRCFD_ACT_SWISH || L3.act != SRCFD_ACT_SWISH || LO.act != SRCFD_ACT_LINEAR) return;
  const double LOG2E = 1.4426950408889634;   // swish layers produce log2(e) x (kernels_tail32.hip, swish_l2e)
  auto& pk = m.pack;
  auto align = [&]() { while (pk.size() % 64) pk.push_back(0.f); };
  align(); m.t32_w1 = pk.size(); pk.resize(pk.size() + 4 * 2 * 16 * 64);
  for (int tap = 0; tap < 4; ++tap)
    for (int t = 0; t < 2; ++t)
      for (int s = 0; s < 16; ++s)
        for (int lane = 0; lane < 64; ++lane) {
          const int mm = lane & 15, kg = lane >> 4;
          pk[m.t32_w1 + (size_t)((tap * 2 + t) * 16 + s) * 64 + lane] = (float)(L1.kernel[((size_t)tap * 32 + 16 * t + mm) * 64 + 16 * kg + s] * LOG2E);
        }
  align(); m.t32_b1 = pk.size(); for (float v : L1.bias) pk.push_back((float)(v * LOG2E));
  align(); m.t32_w2 = pk.size(); pk.resize(pk.size() + 4 * 8 * 64);
  for (int tap = 0; tap < 4; ++tap)
    for (int t = 0; t < 2; ++t)
      for (int ii = 0; ii < 4; ++ii)
        for (int lane = 0; lane < 64; ++lane) {
          const int mm = lane & 15, kg = lane >> 4;
          pk[m.t32_w2 + (size_t)(tap * 8 + 4 * t + ii) * 64 + lane] = L2.kernel[((size_t)tap * 16 + mm) * 32 + 16 * t + 4 * kg + ii];
        }
  align(); m.t32_b2 = pk.size(); for (float v : L2.bias) pk.push_back((float)(v * LOG2E));
  align(); m.t32_w3 = pk.size(); pk.resize(pk.size() + 2 * 4 * 64);
  for (int u = 0; u < 2; ++u)
    for (int ii = 0; ii < 4; ++ii)
      for (int lane = 0; lane < 64; ++lane) {
        const int mm = lane & 15, kg = lane >> 4;
        pk[m.t32_w3 + (size_t)(u * 4 + ii) * 64 + lane] = L3.kernel[((size_t)(2 * u + (mm >> 3)) * 8 + (mm & 7)) * 16 + 4 * kg + ii];
      }
  align(); m.t32_b3 = pk.size(); for (float v : L3.bias) pk.push_back((float)(v * LOG2E));
  align(); m.t32_wc = pk.size(); for (float v : LO.kernel) pk.push_back((float)(v / LOG2E)); pk.push_back(LO.bias[0]);
  align();
  m.tail32_op = (int)i;
}

// SRCFD_PREC_FP32X3: the ops gemm_x3 takes get their weights split into three bf16 planes (kernels_x3.hip).
static void plan_x3(Model& m) {
  m.x3_off.assign(m.ops.size(), -1);
  m.pack_x3.clear();
  for (size_t i = 0; i < m.ops.size(); ++i) {
    GemmDesc d = m.ops[i].d;
    d.M = 0;
    if (!gemm_x3_qualifies(d)) continue;
    while (m.pack_x3.size() % 64) m.pack_x3.push_back(0);
    m.x3_off[i] = (int64_t)m.pack_x3.size();
    m.pack_x3.resize(m.pack_x3.size() + (size_t)3 * d.N * gemm_x3_kpad(d));
    gemm_x3_split_weights(d, m.pack.data() + m.ops[i].w_off, m.pack_x3.data() + m.x3_off[i]);
  }
  // the streaming tail's first layer (kernels_tail32.hip, X3): w1 x log2(e), rounded to f32 as in plan_tail32, then split
  //   w1x[((((ty1*2 + tx1)*2 + t)*2 + c)*3 + plane)*64 + lane][j] = plane(W1[2 ty1 + tx1][co 16t + m][ci 32c + 8 kg + j]),  lane = (m, kg)
  m.t32_w1x = m.t32_w2x = -1;
  auto split3 = [](float w, uint16_t (&o)[3]) {
    uint32_t b0, b1, b2;
    std::memcpy(&b0, &w, 4); b0 &= 0xffff0000u;
    float hi; std::memcpy(&hi, &b0, 4);
    const float r1 = w - hi;
    std::memcpy(&b1, &r1, 4); b1 &= 0xffff0000u;
    float mid; std::memcpy(&mid, &b1, 4);
    const float r2 = r1 - mid;
    std::memcpy(&b2, &r2, 4);
    o[0] = (uint16_t)(b0 >> 16); o[1] = (uint16_t)(b1 >> 16); o[2] = (uint16_t)(b2 >> 16);
  };
  if (m.tail32_op >= 0) {
    const Layer& L2 = m.desc.layers[m.ops[m.tail32_op + 1].layer];
    while (m.pack_x3.size() % 64) m.pack_x3.push_back(0);
    m.t32_w2x = (int64_t)m.pack_x3.size();
    m.pack_x3.resize(m.pack_x3.size() + (size_t)4 * 3 * 64 * 8);
    for (int tap = 0; tap < 4; ++tap)
      for (int lane = 0; lane < 64; ++lane)
        for (int j = 0; j < 8; ++j) {
          const int mm = lane & 15, kg = lane >> 4, ci = j < 4 ? 4 * kg + j : 16 + 4 * kg + (j - 4);
          uint16_t o[3];
          split3(L2.kernel[((size_t)tap * 16 + mm) * 32 + ci], o);
          for (int pl = 0; pl < 3; ++pl) m.pack_x3[m.t32_w2x + ((size_t)(tap * 3 + pl) * 64 + lane) * 8 + j] = o[pl];
        }
    const Layer& L1 = m.desc.layers[m.ops[m.tail32_op].layer];
    const double LOG2E = 1.4426950408889634;
    while (m.pack_x3.size() % 64) m.pack_x3.push_back(0);
    m.t32_w1x = (int64_t)m.pack_x3.size();
    m.pack_x3.resize(m.pack_x3.size() + (size_t)2 * 2 * 2 * 2 * 3 * 64 * 8);
    uint16_t* out = m.pack_x3.data() + m.t32_w1x;
    for (int ty1 = 0; ty1 < 2; ++ty1)
      for (int tx1 = 0; tx1 < 2; ++tx1)
        for (int t = 0; t < 2; ++t)
          for (int c = 0; c < 2; ++c)
            for (int lane = 0; lane < 64; ++lane)
              for (int j = 0; j < 8; ++j) {
                const int mm = lane & 15, kg = lane >> 4, tap = 2 * ty1 + tx1;
                const float w = (float)(L1.kernel[((size_t)tap * 32 + 16 * t + mm) * 64 + 32 * c + 8 * kg + j] * LOG2E);
                uint32_t b0, b1, b2;
                std::memcpy(&b0, &w, 4); b0 &= 0xffff0000u;
                float hi; std::memcpy(&hi, &b0, 4);
                const float r1 = w - hi;
                std::memcpy(&b1, &r1, 4); b1 &= 0xffff0000u;
                float mid; std::memcpy(&mid, &b1, 4);
                const float r2 = r1 - mid;
                std::memcpy(&b2, &r2, 4);
                const size_t frag = (size_t)((((ty1 * 2 + tx1) * 2 + t) * 2 + c) * 3);
                out[((frag + 0) * 64 + lane) * 8 + j] = (uint16_t)(b0 >> 16);
                out[((frag + 1) * 64 + lane) * 8 + j] = (uint16_t)(b1 >> 16);
                out[((frag + 2) * 64 + lane) * 8 + j] = (uint16_t)(b2 >> 16);
              }
  }
}

// enc32 (kernels_enc32.hip): the encoder's four compute layers as one launch.  conv2d_1's weights are re-ordered into
// v_mfma_f32_16x16x4_f32 A fragments: frag[((w*36 + tap*4 + q)*64 + lane)*4 + j] = W[tap][ci = 16 q + 4 (lane / 16) + j][co = 16 w + lane % 16]
// (Keras Conv2D kernel (kh, kw, cin, cout)); the other three layers use their ordinary B[K][Npad] operands.
static void plan_enc32(Model& m) {
  m.enc32_ok = false;
  if (m.ops.size() < 4 || m.desc.in_shape[0] != 10 || m.desc.in_shape[1] != 10 || m.desc.in_shape[2] != 1) return;
  for (int i = 0; i < 3; ++i) if (m.ops[i + 1].layer == m.ops[i].layer) return;   // one op per layer
  if (m.ops.size() > 4 && m.ops[4].layer == m.ops[3].layer) return;
  const GemmDesc &c1 = m.ops[0].d, &c2 = m.ops[1].d, &de = m.ops[2].d, &la = m.ops[3].d;
  const Layer &L0 = m.desc.layers[m.ops[0].layer], &L1 = m.desc.layers[m.ops[1].layer];
  auto act_ok = [](int a) { return a == SRCFD_ACT_SWISH || a == SRCFD_ACT_LINEAR; };
  if (L0.kind != SRCFD_LAYER_CONV2D || L0.kh != 3 || L0.kw != 3 || L0.stride != 2 || !L0.same || L0.cin != 1 || L0.cout != 64 || c1.Npad != 64) return;
  if (L1.kind != SRCFD_LAYER_CONV2D || L1.kh != 3 || L1.kw != 3 || L1.stride != 1 || !L1.same || L1.cin != 64 || L1.cout != 128 || c2.MH != 5 || c2.MW != 5) return;
  if (de.MH != 1 || de.MW != 1 || de.K != 3200 || de.N != 128 || de.Npad != 128) return;
  if (la.MH != 1 || la.MW != 1 || la.K != 128 || la.N > 128 || la.OC != la.N) return;
  if (!act_ok(c1.act) || !act_ok(c2.act) || !act_ok(de.act) || !act_ok(la.act)) return;
  auto& pk = m.pack;
  while (pk.size() % 64) pk.push_back(0.f);
  m.enc32_w2 = pk.size();
  pk.resize(pk.size() + (size_t)8 * 36 * 64 * 4);
  for (int w = 0; w < 8; ++w)
    for (int tap = 0; tap < 9; ++tap)
      for (int q = 0; q < 4; ++q)
        for (int lane = 0; lane < 64; ++lane)
          for (int j = 0; j < 4; ++j) {
            const int ci = 16 * q + 4 * (lane >> 4) + j, co = 16 * w + (lane & 15);
            pk[m.enc32_w2 + ((size_t)((w * 36 + tap * 4 + q) * 64 + lane)) * 4 + j] = L1.kernel[((size_t)tap * 64 + ci) * 128 + co];
          }
  while (pk.size() % 64) pk.push_back(0.f);
  m.enc32_ok = true;
}

// ---------------------------------------------------------------------------
// model
// ---------------------------------------------------------------------------
Model::~Model() {
  if (device >= 0) {
    (void)hipSetDevice(device);
    free_workspace();
    if (d_pack) (void)hipFree(d_pack);
    if (d_pack_x3) (void)hipFree(d_pack_x3);
    fused_free(*this);
    drop_graph();
    if (graph_stream) (void)hipStreamDestroy(graph_stream);
    for (auto& e : prof_events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
  }
}

Switches Switches::from_env() {
  auto on = [](const char* name, bool dflt) { const char* e = getenv(name); return e ? atoi(e) != 0 : dflt; };
  Switches w;
  w.enc16 = on("SRCFD_ENC", true);
  w.mid16 = on("SRCFD_MID", true);
  { const char* e = getenv("SRCFD_MID"); const int v = e ? atoi(e) : 0; if (v >= 1 && v <= 3) w.mid_shape = v; }
  w.dense1_16 = on("SRCFD_DENSE1", true);
  w.enc32 = !on("SRCFD_NO_ENC32", false);
  w.skinny32 = !on("SRCFD_NO_DENSE_SKINNY", false);
  { const char* t = getenv("SRCFD_TAIL"); w.tail16s = t && t[0] == 's'; }
  { const char* e = getenv("SRCFD_MID_ORDER"); if (e) w.mid_order = atoi(e) == 1 ? 1 : 0; }
  { const char* e = getenv("SRCFD_MID_WAVES"); const int v = e ? atoi(e) : 0; w.mid_waves = (v == 4 || v == 16) ? v : 0; }
  const char* e = getenv("SRCFD_TAIL_SEG");
  const int seg = e ? atoi(e) : 0;
  w.tail_seg = (seg == 1 || seg == 2 || seg == 5 || seg == 10 || seg == 25) ? seg : 0;
  return w;
}

void Model::drop_graph() {
  if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
  graph_key = GraphKey();
}

void Model::free_workspace() {
  for (int i = 0; i < 2; ++i) if (buf[i]) { (void)hipFree(buf[i]); buf[i] = nullptr; }
  for (void* p : {(void*)d_x_stage, (void*)d_y_stage, (void*)d_y_stage2, (void*)d_aff, (void*)d_nonfinite, (void*)d_splitk, (void*)d_solver_state}) if (p) (void)hipFree(p);
  d_y_stage2 = nullptr;
  if (copy_stream) { (void)hipStreamDestroy(copy_stream); copy_stream = nullptr; }
  for (int b = 0; b < 2; ++b) {
    if (ev_computed[b]) { (void)hipEventDestroy(ev_computed[b]); ev_computed[b] = nullptr; }
    if (ev_copied[b]) { (void)hipEventDestroy(ev_copied[b]); ev_copied[b] = nullptr; }
  }
  d_solver_state = nullptr; solver_state_elems = 0; d_splitk = nullptr; splitk_floats = 0;
  d_x_stage = d_y_stage = nullptr; d_aff = nullptr; d_nonfinite = nullptr;
  ws_chunk = 0; stage_chunk = 0; ws_per_sample = 0;
}

int Model::init_device() {
  if (device < 0) return SRCFD_OK;
  int cnt = 0;
  if (hipGetDeviceCount(&cnt) != hipSuccess || cnt <= 0) {
    (void)hipGetLastError();
    set_error("no HIP device available (libsrcfd has no CPU fallback)");
    return SRCFD_ENODEV;
  }
  if (device >= cnt) { set_error("device index out of range"); return SRCFD_EINVAL; }
  HIPCHECK(hipSetDevice(device));
  { hipDeviceProp_t prop; HIPCHECK(hipGetDeviceProperties(&prop, device)); num_cus = prop.multiProcessorCount; }
  HIPCHECK(hipMalloc(&d_pack, pack.size() * sizeof(float)));
  HIPCHECK(hipMemcpy(d_pack, pack.data(), pack.size() * sizeof(float), hipMemcpyHostToDevice));
  if (!pack_x3.empty()) {
    HIPCHECK(hipMalloc(&d_pack_x3, pack_x3.size() * sizeof(uint16_t)));
    HIPCHECK(hipMemcpy(d_pack_x3, pack_x3.data(), pack_x3.size() * sizeof(uint16_t), hipMemcpyHostToDevice));
  }
  return SRCFD_OK;
}

static bool tail32_disabled() {
  static const bool off = [] { const char* e = getenv("SRCFD_NO_TAIL32"); return e && atoi(e) != 0; }();
  return off;
}

// Largest activation that sits in a ping-pong buffer.  With the streaming tail (tail32) the layers from its first one on
// never materialise: the largest tensor of the f32 path is then ConvT#1's output (640 KB per sample, not 5.12 MB).
size_t Model::max_act_elems() const {
  size_t m = (size_t)desc.in_shape[0] * desc.in_shape[1] * desc.in_shape[2];
  const bool streaming = tail32_op >= 0 && !tail32_disabled() && precision != SRCFD_PREC_FP32_NAIVE;   // the bring-up path runs layer by layer
  const int stop = streaming ? ops[tail32_op].layer : (int)desc.layers.size();
  for (int i = 0; i < stop; ++i) {
    const Layer& L = desc.layers[i];
    m = std::max(m, (size_t)L.out_shape[0] * L.out_shape[1] * L.out_shape[2]);
  }
  return m;
}

// Samples per pass of the layer-by-layer path: two buffers of the largest activation within 3 GB, at most 1024 samples
// (batch 256 = 768 samples runs as one pass with the streaming tail, as three without it).
int Model::chunk_cap() const {
  size_t per = 2 * max_act_elems() * sizeof(float);
  size_t cap = ((size_t)3 << 30) / std::max<size_t>(per, 1);
  return (int)std::max<size_t>(1, std::min<size_t>(cap, 1024));
}

int Model::ensure_workspace(int n) {
  int chunk = std::min(n, chunk_cap());
  const size_t per = max_act_elems();   // depends on the precision: the bring-up path materialises every layer
  if (chunk <= ws_chunk && per <= ws_per_sample) return SRCFD_OK;
  drop_graph();  // a captured forward holds the old buffers' addresses
  for (int i = 0; i < 2; ++i) if (buf[i]) { HIPCHECK(hipFree(buf[i])); buf[i] = nullptr; }
  ws_chunk = 0;
  size_t bytes = (size_t)chunk * per * sizeof(float);
  for (int i = 0; i < 2; ++i) HIPCHECK(hipMalloc(&buf[i], bytes));
  ws_per_sample = per;
  size_t need = 0;
  for (size_t i = 0; i < ops.size();) {  // the ops of one layer (ConvT output phases) are launched together: their slabs coexist
    size_t j = i;
    GemmDesc ds[4];
    int cnt = 0;
    for (; j < ops.size() && ops[j].layer == ops[i].layer && cnt < 4; ++j) { ds[cnt] = ops[j].d; ds[cnt].M = chunk * ds[cnt].MH * ds[cnt].MW; ++cnt; }
    static const bool no_big = [] { const char* e = getenv("SRCFD_NO_GEMM32_BIG"); return e && atoi(e) != 0; }();
    if (no_big || !gemm32_big_qualifies(ds, cnt))   // gemm32_big sums its K slabs in registers: no slab workspace for those launches
      need = std::max(need, gemm_group_ws_floats(ds, cnt));
    i = j;
  }
  if (d_splitk) { HIPCHECK(hipFree(d_splitk)); d_splitk = nullptr; }
  splitk_floats = need;
  if (need) HIPCHECK(hipMalloc(&d_splitk, need * sizeof(float)));
  ws_chunk = chunk;
  return SRCFD_OK;
}

int Model::launch(const char* name, hipStream_t s, const std::function<hipError_t()>& fn) {
  ProfEvent* pe = nullptr;
  if (profiling) {
    if (prof_used == prof_events.size()) {
      ProfEvent e;
      HIPCHECK(hipEventCreate(&e.a));
      HIPCHECK(hipEventCreate(&e.b));
      prof_events.push_back(e);
    }
    pe = &prof_events[prof_used++];
    pe->name = name;
    HIPCHECK(hipEventRecord(pe->a, s));
  }
  hipError_t e = fn();
  if (e != hipSuccess) { set_error(std::string("launch of ") + name + " failed: " + hipGetErrorString(e)); return SRCFD_EHIP; }
  if (pe) HIPCHECK(hipEventRecord(pe->b, s));
  return SRCFD_OK;
}

// Generic layer-by-layer f32 forward of one chunk (<= ws_chunk samples).
int Model::forward_generic(const float* x_dev, int n, const float* aff_in, const float* aff_out, void* y_dev, int out_dtype,
                           int flags, unsigned long long* nonfinite, hipStream_t s) {
  const int in_elems = desc.in_shape[0] * desc.in_shape[1] * desc.in_shape[2];
  const int* os = desc.out_shape();
  const int out_elems = os[0] * os[1] * os[2];
  const bool naive = precision == SRCFD_PREC_FP32_NAIVE;
  const bool x3 = precision == SRCFD_PREC_FP32X3;
  int cur = 0;
  int rc = SRCFD_OK;
  int prev_layer = -1;
  size_t first = 0;
  const bool no_enc32 = !sw.enc32;   // functional A/B switch of the tests (Switches, engine.h)
  if (enc32_ok && !naive && !no_enc32) {   // standardise + the encoder's four layers: one launch, latent vectors into buf[0]
    Enc32Params ep;
    ep.x = x_dev; ep.affine = aff_in; ep.n = n;
    ep.w1 = d_pack + ops[0].w_off; ep.b1 = d_pack + ops[0].b_off;
    ep.w2f = d_pack + enc32_w2; ep.b2 = d_pack + ops[1].b_off;
    ep.wd = d_pack + ops[2].w_off; ep.bd = d_pack + ops[2].b_off;
    ep.wl = d_pack + ops[3].w_off; ep.bl = d_pack + ops[3].b_off;
    ep.z = buf[0];
    ep.nl = ops[3].d.N; ep.nl_pad = ops[3].d.Npad;
    ep.act1 = ops[0].d.act; ep.act2 = ops[1].d.act; ep.act3 = ops[2].d.act; ep.act4 = ops[3].d.act;
    rc = launch("encoder(standardize..latent_vector)", s, [&] { return launch_enc32(ep, s); });
    if (rc) return rc;
    first = 4;
    prev_layer = ops[3].layer;
    cur = 1;          // the next op is a new layer: it flips to buf[0], where the latent vectors are
  } else {
    rc = launch("standardize", s, [&] { return launch_standardize(x_dev, buf[0], aff_in, in_elems, (int64_t)n * in_elems, s); });
    if (rc) return rc;
  }
  for (size_t i = first; i < ops.size(); ++i) {
    const Op& op = ops[i];
    if (op.layer != prev_layer && prev_layer >= 0) cur ^= 1;
    prev_layer = op.layer;
    GemmDesc d = op.d;
    d.M = n * d.MH * d.MW;
    const float* X = buf[cur];
    float* Y = buf[cur ^ 1];
    const float* B = d_pack + op.w_off;
    const float* bias = d_pack + op.b_off;
    if (!naive && i + 1 == ops.size() && gemm_fuses_finalize(d)) {  // last layer: epilogue writes the caller's buffer directly
      return launch(op.name.c_str(), s, [&] {
        return launch_gemm_finalize(d, X, B, bias, y_dev, out_dtype, aff_out, flags & SRCFD_FLAG_NAN_GUARD, nonfinite, s);
      });
    }
    if (!naive && !tail32_disabled() && (int)i == tail32_op) {  // ConvT#2 -> #3 -> #4 -> output conv + finalize: one streaming kernel
      Tail32Params tp;
      tp.in = X; tp.out = y_dev; tp.n = n; tp.H = d.MH; tp.W = d.MW;
      tp.w1f = d_pack + t32_w1; tp.b1 = d_pack + t32_b1; tp.w2f = d_pack + t32_w2; tp.b2 = d_pack + t32_b2;
      tp.w3f = d_pack + t32_w3; tp.b3 = d_pack + t32_b3; tp.wc = d_pack + t32_wc;
      tp.aff_out = aff_out; tp.nan_guard = flags & SRCFD_FLAG_NAN_GUARD; tp.nonfinite = nonfinite; tp.out_dtype = out_dtype;
      tp.seg = tail32_segments(n, d.MH, num_cus);
      if (x3 && d_pack_x3 && t32_w1x >= 0 && t32_w2x >= 0 && n >= 64) {   // SRCFD_PREC_FP32X3: the first two layers on the bf16 matrix cores
        tp.w1x = d_pack_x3 + t32_w1x;
        tp.w2x = d_pack_x3 + t32_w2x;
      }
#ifdef SRCFD_DIAG
      { static const int abl = [] { const char* e = getenv("SRCFD_TAIL32_ABLATE"); return e ? atoi(e) : 0; }(); tp.ablate = abl; }
#endif
      const std::string nm = op.name + "+" + ops[i + 1].name + "+" + ops[i + 2].name + "+" + ops[i + 3].name + (tp.w1x ? "(x3)" : "");
      return launch(nm.c_str(), s, [&] { return launch_tail32(tp, num_cus, s); });
    }
    static const bool no_pair = [] { const char* e = getenv("SRCFD_NO_PAIR"); return e && atoi(e) != 0; }();
    static const bool no_triple = [] { const char* e = getenv("SRCFD_NO_TRIPLE"); return e && atoi(e) != 0; }();
    if (!naive && !no_pair && !no_triple && (int)i == triple_op && i + 3 < ops.size()) {  // ConvT#2 -> #3 -> #4 in one kernel
      TripleDesc td;
      td.n = n; td.H = d.MH; td.W = d.MW; td.act1 = d.act; td.act2 = ops[i + 1].d.act; td.act3 = ops[i + 2].d.act;
      const std::string nm = op.name + "+" + ops[i + 1].name + "+" + ops[i + 2].name;
      rc = launch(nm.c_str(), s, [&] {
        return launch_convt_triple_f32(td, X, d_pack + tri_w1, d_pack + tri_b1, d_pack + tri_w2, d_pack + tri_b2, d_pack + tri_w3,
                                       d_pack + tri_b3, Y, s);
      });
      if (rc) return rc;
      prev_layer = ops[i + 2].layer;
      i += 2;
      continue;
    }
    if (!naive && !no_pair && (int)i == pair_op && i + 2 < ops.size()) {  // ConvT#3 -> ConvT#4 in one kernel (kernels.h, PairDesc)
      PairDesc pd;
      pd.n = n; pd.H = d.MH; pd.W = d.MW; pd.act_a = d.act; pd.act_b = ops[i + 1].d.act;
      const std::string nm = op.name + "+" + ops[i + 1].name;
      rc = launch(nm.c_str(), s, [&] {
        return launch_convt_pair_f32(pd, X, d_pack + pair_wa, d_pack + pair_ba, d_pack + pair_wb, d_pack + pair_bb, Y, s);
      });
      if (rc) return rc;
      prev_layer = ops[i + 1].layer;  // the pair's output sits where the first layer's would: the next layer toggles once
      ++i;
      continue;
    }
    size_t j = i + 1;
    while (j < ops.size() && ops[j].layer == op.layer && j - i < 4) ++j;
    // SRCFD_PREC_FP32X3: every GEMM of the layer on the split-bf16 kernel (one launch per output phase) -- from 64 samples on: below
    // that its serial k loops lose to the generic f32 kernels' split-K launches (tools/precision_sweep.py: 0.26 vs 0.13 ms at 3
    // samples, break-even between 48 and 96), so small calls run exactly the SRCFD_PREC_FP32 path.
    if (x3 && d_pack_x3 && n >= 64) {
      bool all = true;
      for (size_t q = i; q < j && all; ++q) {
        GemmDesc dq = ops[q].d;
        dq.M = n * dq.MH * dq.MW;
        all = x3_off[q] >= 0 && gemm_x3_qualifies(dq);
      }
      if (all) {
        GemmDesc dsx[4];
        const uint16_t* wx[4];
        const float* bx[4];
        const int cnt = (int)(j - i);
        for (int q = 0; q < cnt; ++q) {
          dsx[q] = ops[i + q].d;
          dsx[q].M = n * dsx[q].MH * dsx[q].MW;
          wx[q] = d_pack_x3 + x3_off[i + q];
          bx[q] = d_pack + ops[i + q].b_off;
        }
        // the layer's output phases: one launch (kernels_x3.hip, X3Group); a single-op layer: its own kernel
        std::string nm = ops[i].name;
        if (cnt > 1) { const size_t dot = nm.rfind(".ph"); if (dot != std::string::npos) nm.resize(dot); }
        nm += "(x3)";
        rc = launch(nm.c_str(), s, [&] { return launch_gemm_x3_group(dsx, cnt, X, wx, bx, Y, s, num_cus); });
        if (rc) return rc;
        i = j - 1;
        continue;
      }
    }
    if (!naive && j - i > 1) {  // the output phases of one transposed convolution: one launch (kernels_fp32.hip, GemmGroup)
      GemmDesc ds[4];
      const float* Bs[4];
      const float* biases[4];
      const int cnt = (int)(j - i);
      for (int q = 0; q < cnt; ++q) {
        ds[q] = ops[i + q].d; ds[q].M = n * ds[q].MH * ds[q].MW;
        Bs[q] = d_pack + ops[i + q].w_off; biases[q] = d_pack + ops[i + q].b_off;
      }
      static const bool no_big = [] { const char* e = getenv("SRCFD_NO_GEMM32_BIG"); return e && atoi(e) != 0; }();
      const bool big = !no_big && gemm32_big_qualifies(ds, cnt);
      rc = launch(desc.layers[op.layer].name.c_str(), s, [&] {
        return big ? launch_gemm32_big(ds, cnt, X, Bs, biases, Y, s) : launch_gemm_mfma_group(ds, cnt, X, Bs, biases, Y, s, d_splitk, splitk_floats);
      });
      if (rc) return rc;
      i = j - 1;
      continue;
    }
    static const bool no_big1 = [] { const char* e = getenv("SRCFD_NO_GEMM32_BIG"); return e && atoi(e) != 0; }();
    const bool no_skinny = !sw.skinny32;
    if (!naive && !no_skinny && dense_skinny32_qualifies(d)) {
      rc = launch(op.name.c_str(), s, [&] { return launch_dense_skinny32(d, X, B, bias, Y, s); });
      if (rc) return rc;
      continue;
    }
    if (!naive && !no_big1 && gemm32_big_qualifies(&d, 1)) {
      rc = launch(op.name.c_str(), s, [&] { return launch_gemm32_big(&d, 1, X, &B, &bias, Y, s); });
      if (rc) return rc;
      continue;
    }
    rc = launch(op.name.c_str(), s, [&] { return naive ? launch_gemm_naive(d, X, B, bias, Y, s) : launch_gemm_mfma(d, X, B, bias, Y, s, d_splitk, splitk_floats); });
    if (rc) return rc;
  }
  cur ^= 1;
  return launch("finalize", s, [&] {
    return launch_finalize(buf[cur], y_dev, out_dtype, aff_out, out_elems, (int64_t)n * out_elems, flags & SRCFD_FLAG_NAN_GUARD, nonfinite, s);
  });
}

int Model::predict_device(const void* x_dev, int n, const float* aff_in, const float* aff_out, void* y_dev, int out_dtype, int flags,
                          unsigned long long* nonfinite, hipStream_t s) {
  if (device < 0) { set_error("host-only handle: no device was requested at create"); return SRCFD_ENODEV; }
  if (n < 0) { set_error("negative batch"); return SRCFD_EINVAL; }
  if (out_dtype != SRCFD_F32 && out_dtype != SRCFD_BF16 && out_dtype != SRCFD_F16) { set_error("unsupported output dtype"); return SRCFD_EINVAL; }
  if (n == 0) return SRCFD_OK;
  HIPCHECK(hipSetDevice(device));
  prof_used = 0;
  sw = Switches::from_env();
  const bool use_fused = (precision == SRCFD_PREC_BF16 || precision == SRCFD_PREC_F16);
  plan = Plan();
  plan.sw = sw; plan.precision = precision; plan.fused = use_fused;
  if (use_fused && !has_fused) { set_error("bf16/f16 precision needs the encoder_10+decoder_400 layer graph; use SRCFD_PREC_FP32 for other graphs"); return SRCFD_EINVAL; }
  if (!use_fused) {
    int rc = ensure_workspace(n);  // (re)allocates before anything is captured; drops a stale graph when it does
    if (rc) return rc;
  }
  const int in_elems = desc.in_shape[0] * desc.in_shape[1] * desc.in_shape[2];
  const int* os = desc.out_shape();
  const size_t out_elems = (size_t)os[0] * os[1] * os[2];
  const size_t osz = out_dtype == SRCFD_F32 ? 4 : 2;
  auto run = [&](hipStream_t st) -> int {
    if (use_fused) return fused_forward(*this, (const float*)x_dev, n, aff_in, aff_out, y_dev, out_dtype, flags, nonfinite, st);
    for (int i = 0; i < n; i += ws_chunk) {
      int c = std::min(ws_chunk, n - i);
      int rc = forward_generic((const float*)x_dev + (size_t)i * in_elems, c, aff_in ? aff_in + 2 * (size_t)i : nullptr,
                               aff_out ? aff_out + 2 * (size_t)i : nullptr, (char*)y_dev + (size_t)i * out_elems * osz, out_dtype, flags,
                               nonfinite, st);
      if (rc) return rc;
    }
    return SRCFD_OK;
  };
  // hipGraph replay of the whole forward.  Measured on MI355X: at batch 256 replay is no faster than plain launches
  // (0.904 vs 0.894 ms: the gaps between the kernels are not host-bound), but a solver-side call of a few samples is
  // a chain of 8-16 kernels of 4-30 us each and replay takes ~20 us off it (bf16 one-field call 0.214 -> 0.195 ms).
  // Default: small batches only; SRCFD_GRAPH=0 never, SRCFD_GRAPH=1 always.
  static const int graph_env = [] { const char* e = getenv("SRCFD_GRAPH"); return e ? (atoi(e) != 0 ? 1 : 0) : -1; }();
  const bool graphs = !profiling && (graph_env == 1 || (graph_env == -1 && n <= 96));
  if (graphs) {
    GraphKey key;
    key.x = x_dev; key.y = y_dev; key.ain = aff_in; key.aout = aff_out; key.nf = nonfinite; key.n = n;
    key.out_dtype = out_dtype; key.flags = flags; key.precision = precision; key.switches = sw.bits();
    if (graph_exec && key == graph_key) {
      HIPCHECK(hipGraphLaunch(graph_exec, s));
      plan = graph_plan; plan.graph = 2;
      return SRCFD_OK;
    }
    if (key == last_key && !(key == graph_key)) {
      // second identical call: every buffer and attribute is set up, so the launches can be captured
      drop_graph();
      if (!graph_stream) HIPCHECK(hipStreamCreateWithFlags(&graph_stream, hipStreamNonBlocking));
      HIPCHECK(hipStreamBeginCapture(graph_stream, hipStreamCaptureModeThreadLocal));
      int rc = run(graph_stream);
      hipGraph_t g = nullptr;
      hipError_t e = hipStreamEndCapture(graph_stream, &g);
      if (rc == SRCFD_OK && e == hipSuccess && g) {
        e = hipGraphInstantiate(&graph_exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e == hipSuccess) {
          graph_key = key;
          graph_plan = plan;
          HIPCHECK(hipGraphLaunch(graph_exec, s));
          plan.graph = 1;
          return SRCFD_OK;
        }
        graph_exec = nullptr;
      } else if (g) (void)hipGraphDestroy(g);
      (void)hipGetLastError();  // capture not possible here: fall through to plain launches
    }
    last_key = key;
  }
  return run(s);
}

int Model::predict_host(const float* x, int n, const float* aff_in, const float* aff_out, float* y, int flags, int64_t* n_nonfinite,
                        const std::function<int(const float*, int, int)>& sink) {
  if (device < 0) { set_error("host-only handle: no device was requested at create"); return SRCFD_ENODEV; }
  if (n < 0 || (n > 0 && (!x || (!y && !sink)))) { set_error("bad arguments"); return SRCFD_EINVAL; }
  if (n_nonfinite) *n_nonfinite = 0;
  if (n == 0) return SRCFD_OK;
  HIPCHECK(hipSetDevice(device));
  const size_t in_elems = (size_t)desc.in_shape[0] * desc.in_shape[1] * desc.in_shape[2];
  const int* os = desc.out_shape();
  const size_t out_elems = (size_t)os[0] * os[1] * os[2];
  // Is the destination page-locked (srcfd_host_alloc, hipHostMalloc / hipHostRegister)?  Then the device-to-host copy of chunk i runs on
  // a second stream while chunk i + 1 is computed (two result buffers, events both ways) and the whole call moves at the PCIe rate;
  // into pageable memory the copy is staged by the runtime and nothing overlaps (one buffer, as before).
  // The WHOLE range [y, y + n out_elems) must be page-locked, not managed, and one allocation: its first and its last byte are asked
  // (a range that starts in a registered block and runs past its end, or managed memory, takes the staged path).
  bool pinned = false;
  if (y && !sink) {
    hipPointerAttribute_t a0, a1;
    const char* last = reinterpret_cast<const char*>(y) + (size_t)n * out_elems * sizeof(float) - 1;
    if (hipPointerGetAttributes(&a0, y) == hipSuccess && hipPointerGetAttributes(&a1, last) == hipSuccess)
      pinned = a0.type == hipMemoryTypeHost && a1.type == hipMemoryTypeHost && !a0.isManaged && !a1.isManaged &&
               reinterpret_cast<const char*>(a1.hostPointer) - reinterpret_cast<const char*>(a0.hostPointer) == last - reinterpret_cast<const char*>(y);
    else (void)hipGetLastError();
  }
  // On the overlapped path an error exit must not leave earlier chunks' copies into the caller's array in flight (the caller drops
  // the array, the pool hands the buffer to the next call): whatever way this function is left, both streams are drained first.
  struct Drain {
    bool armed; hipStream_t* cs;
    ~Drain() { if (armed) { if (*cs) (void)hipStreamSynchronize(*cs); (void)hipStreamSynchronize(nullptr); } }
  } drain{pinned, &copy_stream};
  int chunk = std::min(n, pinned ? 128 : 256);
  if (chunk > stage_chunk || (pinned && !d_y_stage2)) {
    chunk = std::max(chunk, stage_chunk);
    for (void* p : {(void*)d_x_stage, (void*)d_y_stage, (void*)d_y_stage2, (void*)d_aff}) if (p) HIPCHECK(hipFree(p));
    d_x_stage = d_y_stage = d_y_stage2 = nullptr; d_aff = nullptr; stage_chunk = 0;
    drop_graph();   // a captured forward holds the old staging addresses
    HIPCHECK(hipMalloc(&d_x_stage, chunk * in_elems * sizeof(float)));
    HIPCHECK(hipMalloc(&d_y_stage, chunk * out_elems * sizeof(float)));
    if (pinned) HIPCHECK(hipMalloc(&d_y_stage2, chunk * out_elems * sizeof(float)));
    HIPCHECK(hipMalloc(&d_aff, (size_t)chunk * 4 * sizeof(float)));
    stage_chunk = chunk;
  }
  if (pinned && !copy_stream) {
    HIPCHECK(hipStreamCreateWithFlags(&copy_stream, hipStreamNonBlocking));
    for (int b = 0; b < 2; ++b) {
      HIPCHECK(hipEventCreateWithFlags(&ev_computed[b], hipEventDisableTiming));
      HIPCHECK(hipEventCreateWithFlags(&ev_copied[b], hipEventDisableTiming));
    }
  }
  if (!d_nonfinite) HIPCHECK(hipMalloc(&d_nonfinite, sizeof(unsigned long long)));
  HIPCHECK(hipMemsetAsync(d_nonfinite, 0, sizeof(unsigned long long), nullptr));
  const int step = pinned ? std::min(stage_chunk, 128) : stage_chunk;
  int k = 0;
  for (int i = 0; i < n; i += step, ++k) {
    int c = std::min(step, n - i);
    const int b = pinned ? (k & 1) : 0;
    float* ydev = b ? d_y_stage2 : d_y_stage;
    HIPCHECK(hipMemcpyAsync(d_x_stage, x + (size_t)i * in_elems, c * in_elems * sizeof(float), hipMemcpyHostToDevice, nullptr));
    float* ain = nullptr;
    float* aout = nullptr;
    if (aff_in) { ain = d_aff; HIPCHECK(hipMemcpyAsync(ain, aff_in + 2 * (size_t)i, (size_t)c * 2 * sizeof(float), hipMemcpyHostToDevice, nullptr)); }
    if (aff_out) { aout = d_aff + 2 * (size_t)stage_chunk; HIPCHECK(hipMemcpyAsync(aout, aff_out + 2 * (size_t)i, (size_t)c * 2 * sizeof(float), hipMemcpyHostToDevice, nullptr)); }
    if (pinned && k >= 2) HIPCHECK(hipStreamWaitEvent(nullptr, ev_copied[b], 0));   // the buffer's previous contents have left
    int rc = predict_device(d_x_stage, c, ain, aout, ydev, SRCFD_F32, flags, d_nonfinite, nullptr);
    if (rc) return rc;
    if (sink) { rc = sink(ydev, i, c); if (rc) return rc; HIPCHECK(hipStreamSynchronize(nullptr)); }
    else if (pinned) {
      HIPCHECK(hipEventRecord(ev_computed[b], nullptr));
      HIPCHECK(hipStreamWaitEvent(copy_stream, ev_computed[b], 0));
      HIPCHECK(hipMemcpyAsync(y + (size_t)i * out_elems, ydev, c * out_elems * sizeof(float), hipMemcpyDeviceToHost, copy_stream));
      HIPCHECK(hipEventRecord(ev_copied[b], copy_stream));
      // the host arrays of this chunk (x, affines) were consumed by stream-ordered copies from pageable memory: complete on return
    } else {
      HIPCHECK(hipMemcpyAsync(y + (size_t)i * out_elems, ydev, c * out_elems * sizeof(float), hipMemcpyDeviceToHost, nullptr));
      HIPCHECK(hipStreamSynchronize(nullptr));
    }
  }
  if (pinned && !sink) { HIPCHECK(hipStreamSynchronize(copy_stream)); HIPCHECK(hipStreamSynchronize(nullptr)); drain.armed = false; }
  if (n_nonfinite) {
    unsigned long long v = 0;
    HIPCHECK(hipMemcpy(&v, d_nonfinite, sizeof(v), hipMemcpyDeviceToHost));
    *n_nonfinite = (int64_t)v;
  }
  return SRCFD_OK;
}

static int finish_create(std::unique_ptr<Model>& m, srcfd_model** out) {
  try {
    m->desc.infer_shapes();
    build_plan(m->desc, m->ops, m->pack);
    plan_convt_pair(*m);
    plan_convt_triple(*m);
    plan_tail32(*m);
    plan_enc32(*m);
    plan_x3(*m);
  } catch (const std::exception& e) {
    set_error(e.what());
    return SRCFD_EINVAL;
  }
  m->has_fused = m->desc.is_sr_10_400();
  int rc = m->init_device();
  if (rc) return rc;
  if (m->device >= 0 && m->has_fused) {
    rc = fused_init(*m);
    if (rc) return rc;
  }
  *out = reinterpret_cast<srcfd_model*>(m.release());
  return SRCFD_OK;
}

}  // namespace srcfd

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------
using srcfd::Model;
using srcfd::set_error;
static Model* M(srcfd_model* m) { return reinterpret_cast<Model*>(m); }
static const Model* M(const srcfd_model* m) { return reinterpret_cast<const Model*>(m); }

extern "C" {

const char* srcfd_last_error(void) { return srcfd::g_last_error.c_str(); }
const char* srcfd_version(void) { return "srcfd 0.1 (gfx950)"; }

int srcfd_device_count(void) {
  return srcfd::abi_guard("srcfd_device_count", [&]() -> int {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
  });
}

int srcfd_host_alloc(size_t bytes, void** out) {
  return srcfd::abi_guard("srcfd_host_alloc", [&]() -> int {
    if (!out || bytes == 0) { set_error("srcfd_host_alloc: bad arguments"); return SRCFD_EINVAL; }
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n < 1) { (void)hipGetLastError(); set_error("no HIP device: page-locked memory needs the runtime"); return SRCFD_ENODEV; }
    hipError_t e = hipHostMalloc(out, bytes, hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; set_error(std::string("hipHostMalloc failed: ") + hipGetErrorString(e)); return SRCFD_ENOMEM; }
    return SRCFD_OK;
  });
}

void srcfd_host_free(void* p) {
  if (p) (void)hipHostFree(p);
}

int srcfd_model_load_h5(const char* encoder_h5, const char* decoder_h5, int device, srcfd_model** out) {
  return srcfd::abi_guard("srcfd_model_load_h5", [&]() -> int {
    if (!out || (!encoder_h5 && !decoder_h5)) { set_error("srcfd_model_load_h5: bad arguments"); return SRCFD_EINVAL; }
    *out = nullptr;
    std::unique_ptr<Model> m(new Model());
    m->device = device;
    try {
      if (encoder_h5) srcfd::append_h5_submodel(m->desc, encoder_h5);
      if (decoder_h5) srcfd::append_h5_submodel(m->desc, decoder_h5);
    } catch (const srcfd::FileError& e) {
      set_error(e.msg);
      return e.code;
    } catch (const std::exception& e) {
      set_error(e.what());
      return SRCFD_EIO;
    }
    return srcfd::finish_create(m, out);
  });
}

int srcfd_model_load_superres_h5(const char* superres_h5, int device, srcfd_model** out) {
  return srcfd::abi_guard("srcfd_model_load_superres_h5", [&]() -> int {
    if (!out || !superres_h5) { set_error("srcfd_model_load_superres_h5: bad arguments"); return SRCFD_EINVAL; }
    *out = nullptr;
    std::unique_ptr<Model> m(new Model());
    m->device = device;
    try {
      srcfd::append_h5_whole(m->desc, superres_h5);
    } catch (const srcfd::FileError& e) {
      set_error(e.msg);
      return e.code;
    } catch (const std::exception& e) {
      set_error(e.what());
      return SRCFD_EIO;
    }
    return srcfd::finish_create(m, out);
  });
}

int srcfd_model_save_superres_h5(const srcfd_model* m, const char* superres_h5) {
  return srcfd::abi_guard("srcfd_model_save_superres_h5", [&]() -> int {
    if (!m || !superres_h5) { set_error("srcfd_model_save_superres_h5: bad arguments"); return SRCFD_EINVAL; }
    try {
      srcfd::save_h5_whole(M(m)->desc, superres_h5);
    } catch (const srcfd::FileError& e) {
      set_error(e.msg);
      return e.code;
    }
    return SRCFD_OK;
  });
}

int srcfd_model_create(const srcfd_layer* layers, int n_layers, const int in_shape[3], int device, srcfd_model** out) {
  return srcfd::abi_guard("srcfd_model_create", [&]() -> int {
    if (!out || !layers || n_layers <= 0 || !in_shape) { set_error("srcfd_model_create: bad arguments"); return SRCFD_EINVAL; }
    *out = nullptr;
    std::unique_ptr<Model> m(new Model());
    m->device = device;
    for (int i = 0; i < 3; ++i) m->desc.in_shape[i] = in_shape[i];
    int cur[3] = {in_shape[0], in_shape[1], in_shape[2]};
    (void)cur;
    for (int i = 0; i < n_layers; ++i) {
      const srcfd_layer& s = layers[i];
      srcfd::Layer L;
      L.kind = s.kind; L.act = s.activation; L.kh = s.kh; L.kw = s.kw; L.stride = s.stride > 0 ? s.stride : 1; L.same = s.same_padding;
      L.cin = s.cin; L.cout = s.cout;
      for (int k = 0; k < 3; ++k) L.reshape[k] = s.reshape[k];
      L.name = (s.name && *s.name) ? std::string(s.name) : "layer_" + std::to_string(i);
      if (s.kind == SRCFD_LAYER_CONV2D || s.kind == SRCFD_LAYER_CONV2D_TRANSPOSE || s.kind == SRCFD_LAYER_DENSE) {
        if (!s.kernel || s.cin <= 0 || s.cout <= 0) { set_error("layer " + std::to_string(i) + ": missing kernel / channels"); return SRCFD_EINVAL; }
        if (s.kind == SRCFD_LAYER_DENSE) { L.kh = L.kw = 1; }
        if (L.kh <= 0 || L.kw <= 0) { set_error("layer " + std::to_string(i) + ": bad kernel size"); return SRCFD_EINVAL; }
        size_t cnt = (size_t)L.kh * L.kw * s.cin * s.cout;
        L.kernel.assign(s.kernel, s.kernel + cnt);
        if (s.bias) L.bias.assign(s.bias, s.bias + s.cout);
        else L.bias.assign(s.cout, 0.f);
      }
      m->desc.layers.push_back(std::move(L));
    }
    // A graph with a `latent_vector` layer in the middle is the encoder/decoder pair of SuperResolutionAE
    // (sr-ae-conv.ipynb:c162-169, c277-287): keep the two halves as separate sub-models so that save_h5 writes the
    // `vanilla_encoder...h5` / `vanilla_decoder...h5` pair the solvers load (PyCFD_ML_accelerated.py:831-832).
    int cut = -1;
    for (int i = 0; i + 1 < n_layers; ++i)
      if (m->desc.layers[i].name == "latent_vector") cut = i + 1;
    if (cut > 0) {
      try { m->desc.infer_shapes(); } catch (const std::exception& e) { set_error(e.what()); return SRCFD_EINVAL; }
      srcfd::SubModel enc, dec;
      enc.name = "encoder_" + std::to_string(in_shape[0]); enc.input_name = enc.name + "_input"; enc.first = 0; enc.count = cut;
      dec.name = "decoder_" + std::to_string(m->desc.out_shape()[0]); dec.input_name = dec.name + "_input"; dec.first = cut; dec.count = n_layers - cut;
      m->desc.subs.push_back(enc);
      m->desc.subs.push_back(dec);
    } else {
      srcfd::SubModel sub;
      sub.name = "model"; sub.input_name = "model_input"; sub.first = 0; sub.count = n_layers;
      m->desc.subs.push_back(sub);
    }
    return srcfd::finish_create(m, out);
  });
}

void srcfd_model_destroy(srcfd_model* m) { delete M(m); }

int srcfd_model_input_shape(const srcfd_model* m, int shape[3]) {
  return srcfd::abi_guard("srcfd_model_input_shape", [&]() -> int {
    if (!m || !shape) { set_error("bad arguments"); return SRCFD_EINVAL; }
    for (int i = 0; i < 3; ++i) shape[i] = M(m)->desc.in_shape[i];
    return SRCFD_OK;
  });
}
int srcfd_model_output_shape(const srcfd_model* m, int shape[3]) {
  return srcfd::abi_guard("srcfd_model_output_shape", [&]() -> int {
    if (!m || !shape) { set_error("bad arguments"); return SRCFD_EINVAL; }
    for (int i = 0; i < 3; ++i) shape[i] = M(m)->desc.out_shape()[i];
    return SRCFD_OK;
  });
}
int srcfd_model_num_layers(const srcfd_model* m) { return m ? (int)M(m)->desc.layers.size() : SRCFD_EINVAL; }

int srcfd_model_get_layer(const srcfd_model* m, int i, srcfd_layer* layer, char* name, size_t name_len) {
  return srcfd::abi_guard("srcfd_model_get_layer", [&]() -> int {
    if (!m || !layer || i < 0 || i >= (int)M(m)->desc.layers.size()) { set_error("bad arguments"); return SRCFD_EINVAL; }
    const srcfd::Layer& L = M(m)->desc.layers[i];
    layer->kind = L.kind; layer->activation = L.act; layer->kh = L.kh; layer->kw = L.kw; layer->stride = L.stride;
    layer->same_padding = L.same; layer->cin = L.cin; layer->cout = L.cout;
    for (int k = 0; k < 3; ++k) layer->reshape[k] = L.reshape[k];
    layer->kernel = L.kernel.empty() ? nullptr : L.kernel.data();
    layer->bias = L.bias.empty() ? nullptr : L.bias.data();
    layer->name = L.name.c_str();
    if (name && name_len) { std::snprintf(name, name_len, "%s", L.name.c_str()); }
    return SRCFD_OK;
  });
}

int64_t srcfd_model_macs_per_sample(const srcfd_model* m) { return m ? M(m)->desc.macs_per_sample() : 0; }

int srcfd_model_set_precision(srcfd_model* m, int precision) {
  return srcfd::abi_guard("srcfd_model_set_precision", [&]() -> int {
    if (!m || precision < 0 || precision > SRCFD_PREC_FP32X3) { set_error("bad precision"); return SRCFD_EINVAL; }
    if ((precision == SRCFD_PREC_BF16 || precision == SRCFD_PREC_F16) && !M(m)->has_fused) {
      set_error("bf16/f16 precision needs the encoder_10+decoder_400 layer graph");
      return SRCFD_EINVAL;
    }
    M(m)->precision = precision;
    M(m)->drop_graph();
    return SRCFD_OK;
  });
}
int srcfd_model_reserve(srcfd_model* m, int n) {
  return srcfd::abi_guard("srcfd_model_reserve", [&]() -> int {
    if (!m || n < 0) { set_error("bad arguments"); return SRCFD_EINVAL; }
    srcfd::Model& mm = *M(m);
    if (mm.device < 0) { set_error("host-only handle: no device was requested at create"); return SRCFD_ENODEV; }
    if (n == 0) return SRCFD_OK;
    HIPCHECK(hipSetDevice(mm.device));
    if (mm.precision == SRCFD_PREC_BF16 || mm.precision == SRCFD_PREC_F16) {
      if (!mm.has_fused) { set_error("bf16/f16 precision needs the encoder_10+decoder_400 layer graph; use SRCFD_PREC_FP32 for other graphs"); return SRCFD_EINVAL; }
      return srcfd::fused_reserve(mm, n);
    }
    return mm.ensure_workspace(n);
  });
}

int srcfd_model_get_precision(const srcfd_model* m) { return m ? M(m)->precision : SRCFD_EINVAL; }
int srcfd_model_has_fused_path(const srcfd_model* m) { return m ? (int)M(m)->has_fused : 0; }

int srcfd_predict(srcfd_model* m, const float* x, int n, const float* in_affine, const float* out_affine, float* y, int flags,
                  int64_t* n_nonfinite) {
  return srcfd::abi_guard("srcfd_predict", [&]() -> int {
    if (!m) { set_error("null model"); return SRCFD_EINVAL; }
    return M(m)->predict_host(x, n, in_affine, out_affine, y, flags, n_nonfinite);
  });
}

int srcfd_predict_device(srcfd_model* m, const void* x_dev, int n, const float* in_affine_dev, const float* out_affine_dev,
                         void* y_dev, int out_dtype, int flags, int64_t* nonfinite_dev, void* hip_stream) {
  return srcfd::abi_guard("srcfd_predict_device", [&]() -> int {
    if (!m || (n > 0 && (!x_dev || !y_dev))) { set_error("bad arguments"); return SRCFD_EINVAL; }
    return M(m)->predict_device(x_dev, n, in_affine_dev, out_affine_dev, y_dev, out_dtype, flags,
                                reinterpret_cast<unsigned long long*>(nonfinite_dev), reinterpret_cast<hipStream_t>(hip_stream));
  });
}

int srcfd_model_workspace(srcfd_model* m, int n, size_t* bytes) {
  return srcfd::abi_guard("srcfd_model_workspace", [&]() -> int {
    if (!m) { set_error("null model"); return SRCFD_EINVAL; }
    int chunk = std::min(std::max(n, 1), M(m)->chunk_cap());
    if (bytes) *bytes = 2 * (size_t)chunk * M(m)->max_act_elems() * sizeof(float);
    return chunk;
  });
}

int srcfd_model_footprint(const srcfd_model* m, int n, int precision, size_t bytes[4]) {
  return srcfd::abi_guard("srcfd_model_footprint", [&]() -> int {
    if (!m || !bytes || n < 0 || precision < 0 || precision > SRCFD_PREC_FP32X3) { set_error("bad arguments"); return SRCFD_EINVAL; }
    const srcfd::Model& mm = *M(m);
    const bool lowp = precision == SRCFD_PREC_BF16 || precision == SRCFD_PREC_F16;
    if (lowp && !mm.has_fused) { set_error("bf16/f16 precision needs the encoder_10+decoder_400 layer graph"); return SRCFD_EINVAL; }
    int shp_in[3] = {0, 0, 0}, shp_out[3] = {0, 0, 0};
    (void)srcfd_model_input_shape(m, shp_in);
    (void)srcfd_model_output_shape(m, shp_out);
    const size_t in_elems = (size_t)shp_in[0] * shp_in[1] * shp_in[2], out_elems = (size_t)shp_out[0] * shp_out[1] * shp_out[2];
    size_t params = 0;
    for (const auto& L : mm.desc.layers) params += L.kernel.size() + L.bias.size();
    if (lowp) {
      const size_t want = (size_t)std::min(n, 1024);
      bytes[0] = want ? 2 * want * 160000 * sizeof(uint16_t) + 16 * want * 128 * sizeof(float) : 0;   // fused_reserve: two activation buffers + the dense split-K slabs
      bytes[1] = params * (sizeof(float) + 2 * sizeof(uint16_t));   // f32 weights + the 16-bit GEMM layout + fragment re-orderings (upper bound: every layer twice)
    } else {
      const int chunk = n ? std::min(n, mm.chunk_cap()) : 0;
      bytes[0] = 2 * (size_t)chunk * mm.max_act_elems() * sizeof(float);
      bytes[1] = params * sizeof(float) * 2 + mm.pack_x3.size() * sizeof(uint16_t);   // packed weights + the per-layer operand orders of the fused f32 kernels (upper bound) + the split-bf16 planes
    }
    const size_t stage = (size_t)std::min(n, 256);
    bytes[2] = stage * (in_elems + 2 * out_elems + 4) * sizeof(float);   // predict_host: input, two result buffers, affine pairs
    bytes[3] = (size_t)n * out_elems * sizeof(float);                    // what ONE host result of the call takes (pool buffer or caller's array)
    return SRCFD_OK;
  });
}

int srcfd_model_set_profiling(srcfd_model* m, int enable) {
  return srcfd::abi_guard("srcfd_model_set_profiling", [&]() -> int {
    if (!m) { set_error("null model"); return SRCFD_EINVAL; }
    M(m)->profiling = enable != 0;
    M(m)->prof_used = 0;
    return SRCFD_OK;
  });
}

int srcfd_model_get_profile(srcfd_model* m, char* names, size_t names_len, float* ms, int* count, int max_count) {
  return srcfd::abi_guard("srcfd_model_get_profile", [&]() -> int {
    if (!m || !count) { set_error("bad arguments"); return SRCFD_EINVAL; }
    Model* mm = M(m);
    if (mm->device < 0) { set_error("host-only handle"); return SRCFD_ENODEV; }
    HIPCHECK(hipSetDevice(mm->device));
    HIPCHECK(hipDeviceSynchronize());
    std::string joined;
    int n = 0;
    for (size_t i = 0; i < mm->prof_used && n < max_count; ++i, ++n) {
      float t = 0.f;
      HIPCHECK(hipEventElapsedTime(&t, mm->prof_events[i].a, mm->prof_events[i].b));
      if (ms) ms[n] = t;
      if (n) joined += '\n';
      joined += mm->prof_events[i].name;
    }
    *count = n;
    if (names && names_len) std::snprintf(names, names_len, "%s", joined.c_str());
    return SRCFD_OK;
  });
}

int srcfd_model_last_plan(const srcfd_model* m, char* buf, size_t buf_len) {
  return srcfd::abi_guard("srcfd_model_last_plan", [&]() -> int {
    if (!m || !buf || buf_len == 0) { set_error("bad arguments"); return SRCFD_EINVAL; }
    const srcfd::Plan& p = M(m)->plan;
    const char* prec = p.precision == SRCFD_PREC_BF16 ? "bf16" : p.precision == SRCFD_PREC_F16 ? "f16" : p.precision == SRCFD_PREC_FP32 ? "fp32" : p.precision == SRCFD_PREC_FP32X3 ? "fp32x3" : "fp32_naive";
    char tmp[256];
    if (p.fused)
      snprintf(tmp, sizeof(tmp), "precision=%s encoder=%s dense_1=%s middle=%s tail=%s tail_seg=%d graph=%s", prec, p.sw.enc16 ? "enc16" : "layers",
               p.sw.dense1_16 ? "dense1_16" : "gemm16", p.sw.mid16 ? (p.sw.mid_waves == 4 ? "mid16_4x32" : p.sw.mid_waves == 16 ? "mid16_16x32" : p.sw.mid_shape == 1 ? "mid16_8x32" : p.sw.mid_shape == 2 ? "mid16_8x64" : "mid16_4x64") : "gemm16", p.sw.tail16s ? "tail16s" : "tail16", p.tail_seg,
               p.graph == 2 ? "replay" : p.graph == 1 ? "capture" : "eager");
    else
      snprintf(tmp, sizeof(tmp), "precision=%s encoder=%s dense_1=%s graph=%s", prec, p.sw.enc32 ? "enc32" : "layers",
               p.sw.skinny32 ? "dense_skinny32" : "gemm32", p.graph == 2 ? "replay" : p.graph == 1 ? "capture" : "eager");
    snprintf(buf, buf_len, "%s", tmp);
    return SRCFD_OK;
  });
}

int srcfd_model_debug_activation(srcfd_model* m, int index, void* dst, size_t bytes) {
  return srcfd::abi_guard("srcfd_model_debug_activation", [&]() -> int {
    if (!m || !dst || index < 0 || index > 1) { set_error("bad arguments"); return SRCFD_EINVAL; }
    return srcfd::fused_debug_read(*M(m), index, dst, bytes);
  });
}

int srcfd_model_save_h5(const srcfd_model* m, const char* encoder_h5, const char* decoder_h5) {
  return srcfd::abi_guard("srcfd_model_save_h5", [&]() -> int {
    if (!m) { set_error("null model"); return SRCFD_EINVAL; }
    const Model* mm = M(m);
    try {
      const char* paths[2] = {encoder_h5, decoder_h5};
      int given = (encoder_h5 ? 1 : 0) + (decoder_h5 ? 1 : 0);
      if (given == 0) { set_error("no output path"); return SRCFD_EINVAL; }
      if ((int)mm->desc.subs.size() == 1) {
        srcfd::save_h5_submodel(mm->desc, 0, encoder_h5 ? encoder_h5 : decoder_h5);
      } else {
        for (int i = 0; i < 2 && i < (int)mm->desc.subs.size(); ++i)
          if (paths[i]) srcfd::save_h5_submodel(mm->desc, i, paths[i]);
      }
    } catch (const srcfd::FileError& e) {
      set_error(e.msg);
      return e.code;
    }
    return SRCFD_OK;
  });
}

}  // extern "C"
