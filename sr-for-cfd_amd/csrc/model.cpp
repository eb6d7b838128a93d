// Layer-graph bookkeeping + legacy Keras-H5 load/save.  See model.h.
#include "model.h"

#include <cmath>
#include <cstdio>
#include <sstream>
#include <stdexcept>

#include "h5lite.h"
#include "json_min.h"

namespace srcfd {

const char* act_name(int act) {
  switch (act) {
    case SRCFD_ACT_SWISH: return "silu";
    case SRCFD_ACT_RELU: return "relu";
    case SRCFD_ACT_SIGMOID: return "sigmoid";
    case SRCFD_ACT_TANH: return "tanh";
    default: return "linear";
  }
}

int act_from_name(const std::string& s) {
  if (s == "swish" || s == "silu") return SRCFD_ACT_SWISH;  // Keras 3 serialises swish as "silu"
  if (s == "linear" || s.empty()) return SRCFD_ACT_LINEAR;
  if (s == "relu") return SRCFD_ACT_RELU;
  if (s == "sigmoid") return SRCFD_ACT_SIGMOID;
  if (s == "tanh") return SRCFD_ACT_TANH;
  throw std::runtime_error("unsupported activation '" + s + "'");
}

static int same_out(int in, int s) { return (in + s - 1) / s; }

void ModelDesc::infer_shapes() {
  int cur[3] = {in_shape[0], in_shape[1], in_shape[2]};
  if (cur[0] <= 0 || cur[1] <= 0 || cur[2] <= 0) throw std::runtime_error("bad input shape");
  for (auto& L : layers) {
    for (int i = 0; i < 3; ++i) L.in_shape[i] = cur[i];
    L.macs = 0;
    switch (L.kind) {
      case SRCFD_LAYER_CONV2D: {
        if (L.cin != cur[2]) throw std::runtime_error("layer '" + L.name + "': channel mismatch");
        int oh = L.same ? same_out(cur[0], L.stride) : (cur[0] - L.kh) / L.stride + 1;
        int ow = L.same ? same_out(cur[1], L.stride) : (cur[1] - L.kw) / L.stride + 1;
        if (oh <= 0 || ow <= 0) throw std::runtime_error("layer '" + L.name + "': empty output");
        cur[0] = oh; cur[1] = ow; cur[2] = L.cout;
        L.macs = (int64_t)oh * ow * L.kh * L.kw * L.cin * L.cout;
        if ((int64_t)L.kernel.size() != (int64_t)L.kh * L.kw * L.cin * L.cout) throw std::runtime_error("layer '" + L.name + "': kernel size mismatch");
        break;
      }
      case SRCFD_LAYER_CONV2D_TRANSPOSE: {
        if (L.cin != cur[2]) throw std::runtime_error("layer '" + L.name + "': channel mismatch");
        if (L.same) throw std::runtime_error("layer '" + L.name + "': Conv2DTranspose padding='same' unsupported");
        int oh = (cur[0] - 1) * L.stride + L.kh, ow = (cur[1] - 1) * L.stride + L.kw;
        L.macs = (int64_t)cur[0] * cur[1] * L.kh * L.kw * L.cin * L.cout;
        cur[0] = oh; cur[1] = ow; cur[2] = L.cout;
        if ((int64_t)L.kernel.size() != (int64_t)L.kh * L.kw * L.cin * L.cout) throw std::runtime_error("layer '" + L.name + "': kernel size mismatch");
        break;
      }
      case SRCFD_LAYER_DENSE: {
        int in = cur[0] * cur[1] * cur[2];
        if (L.cin != in) throw std::runtime_error("layer '" + L.name + "': expects " + std::to_string(L.cin) + " inputs, gets " + std::to_string(in));
        cur[0] = 1; cur[1] = 1; cur[2] = L.cout;
        L.macs = (int64_t)L.cin * L.cout;
        if ((int64_t)L.kernel.size() != (int64_t)L.cin * L.cout) throw std::runtime_error("layer '" + L.name + "': kernel size mismatch");
        break;
      }
      case SRCFD_LAYER_FLATTEN: {
        int n = cur[0] * cur[1] * cur[2];
        cur[0] = 1; cur[1] = 1; cur[2] = n;
        break;
      }
      case SRCFD_LAYER_RESHAPE: {
        int n = cur[0] * cur[1] * cur[2];
        if (L.reshape[0] * L.reshape[1] * L.reshape[2] != n) throw std::runtime_error("layer '" + L.name + "': reshape size mismatch");
        for (int i = 0; i < 3; ++i) cur[i] = L.reshape[i];
        break;
      }
      default: throw std::runtime_error("unknown layer kind");
    }
    if (L.kind <= SRCFD_LAYER_DENSE && (int)L.bias.size() != L.cout && !L.bias.empty())
      throw std::runtime_error("layer '" + L.name + "': bias size mismatch");
    for (int i = 0; i < 3; ++i) L.out_shape[i] = cur[i];
  }
}

int64_t ModelDesc::macs_per_sample() const {
  int64_t t = 0;
  for (auto& L : layers) t += L.macs;
  return t;
}

bool ModelDesc::is_sr_10_400() const {
  // the compute layers, ignoring Flatten/Reshape
  struct P { int kind, k, s, cin, cout, act; };
  static const P pat[] = {
      {SRCFD_LAYER_CONV2D, 3, 2, 1, 64, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_CONV2D, 3, 1, 64, 128, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_DENSE, 1, 1, 3200, 128, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_DENSE, 1, 1, 128, 50, SRCFD_ACT_LINEAR},
      {SRCFD_LAYER_DENSE, 1, 1, 50, 36864, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_CONV2D_TRANSPOSE, 3, 2, 256, 128, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_CONV2D_TRANSPOSE, 2, 2, 128, 64, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_CONV2D_TRANSPOSE, 2, 2, 64, 32, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_CONV2D_TRANSPOSE, 2, 2, 32, 16, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_CONV2D_TRANSPOSE, 2, 2, 16, 8, SRCFD_ACT_SWISH},
      {SRCFD_LAYER_CONV2D, 3, 1, 8, 1, SRCFD_ACT_LINEAR},
  };
  if (in_shape[0] != 10 || in_shape[1] != 10 || in_shape[2] != 1) return false;
  size_t pi = 0;
  for (auto& L : layers) {
    if (L.kind == SRCFD_LAYER_FLATTEN || L.kind == SRCFD_LAYER_RESHAPE) continue;
    if (pi >= sizeof(pat) / sizeof(pat[0])) return false;
    const P& p = pat[pi++];
    if (L.kind != p.kind || L.cin != p.cin || L.cout != p.cout || L.act != p.act) return false;
    if (L.kind != SRCFD_LAYER_DENSE && (L.kh != p.k || L.kw != p.k || L.stride != p.s)) return false;
    if (L.kind == SRCFD_LAYER_CONV2D && !L.same) return false;
    if (L.bias.empty()) return false;
  }
  return pi == sizeof(pat) / sizeof(pat[0]) && layers.back().out_shape[0] == 400;
}

// ---------------------------------------------------------------------------
// load
// ---------------------------------------------------------------------------
static std::vector<int> int_list(const jsonmin::Value& v) {
  std::vector<int> r;
  if (v.kind == jsonmin::Value::Arr) for (auto& e : v.arr) r.push_back(e.kind == jsonmin::Value::Num ? e.as_int() : -1);
  else if (v.kind == jsonmin::Value::Num) r = {v.as_int(), v.as_int()};
  return r;
}

static std::string activation_of(const jsonmin::Value& cfg) {
  const jsonmin::Value* a = cfg.get("activation");
  if (!a || a->kind == jsonmin::Value::Null) return "linear";
  if (a->kind == jsonmin::Value::Str) return a->str;
  // serialized activation object {"class_name": ..., "config": "silu"} (some Keras 3 builds)
  if (a->kind == jsonmin::Value::Obj) {
    if (const jsonmin::Value* c = a->get("config")) if (c->kind == jsonmin::Value::Str) return c->str;
    if (const jsonmin::Value* c = a->get("class_name")) if (c->kind == jsonmin::Value::Str) return c->str;
  }
  throw std::runtime_error("unparseable activation");
}

// Appends the layers of ONE sub-model described by a Functional / Sequential config `mc`; `mw` is the group its weights hang
// under (`nested`: the whole-model layout, see append_h5_whole).
static void parse_submodel(ModelDesc& m, const jsonmin::Value& mc, h5lite::File* file, h5lite::Node* mw, bool nested) {
    SubModel sub;
    sub.name = mc.get("name") ? mc.at("name").str : "model";
    sub.first = (int)m.layers.size();
    int sub_in[3] = {0, 0, 0};
    for (auto& lj : mc.at("layers").arr) {
      const std::string lc = lj.at("class_name").str;
      const jsonmin::Value& c = lj.at("config");
      const std::string lname = c.at("name").str;
      if (lc == "InputLayer") {
        const jsonmin::Value* bs = c.get("batch_shape");
        if (!bs) bs = c.get("batch_input_shape");
        if (!bs) throw std::runtime_error("InputLayer without batch_shape");
        std::vector<int> s = int_list(*bs);
        if (s.size() == 4) { sub_in[0] = s[1]; sub_in[1] = s[2]; sub_in[2] = s[3]; }
        else if (s.size() == 2) { sub_in[0] = 1; sub_in[1] = 1; sub_in[2] = s[1]; }
        else throw std::runtime_error("InputLayer rank unsupported");
        sub.input_name = lname;
        continue;
      }
      Layer L;
      L.name = lname;
      bool has_w = false;
      if (lc == "Conv2D" || lc == "Conv2DTranspose") {
        L.kind = lc == "Conv2D" ? SRCFD_LAYER_CONV2D : SRCFD_LAYER_CONV2D_TRANSPOSE;
        std::vector<int> ks = int_list(c.at("kernel_size")), st = int_list(c.at("strides"));
        if (ks.size() != 2 || st.size() != 2 || st[0] != st[1]) throw std::runtime_error("layer '" + lname + "': anisotropic stride unsupported");
        L.kh = ks[0]; L.kw = ks[1]; L.stride = st[0];
        const std::string pad = c.at("padding").str;
        if (pad != "same" && pad != "valid") throw std::runtime_error("layer '" + lname + "': padding '" + pad + "'");
        L.same = pad == "same";
        L.cout = c.at("filters").as_int();
        if (const auto* df = c.get("data_format")) if (df->kind == jsonmin::Value::Str && df->str != "channels_last") throw std::runtime_error("channels_first unsupported");
        if (const auto* dr = c.get("dilation_rate")) { auto d = int_list(*dr); if (d.size() == 2 && (d[0] != 1 || d[1] != 1)) throw std::runtime_error("dilation unsupported"); }
        if (const auto* g = c.get("groups")) if (g->kind == jsonmin::Value::Num && g->as_int() != 1) throw std::runtime_error("groups unsupported");
        L.act = act_from_name(activation_of(c));
        has_w = true;
      } else if (lc == "Dense") {
        L.kind = SRCFD_LAYER_DENSE;
        L.cout = c.at("units").as_int();
        L.act = act_from_name(activation_of(c));
        has_w = true;
      } else if (lc == "Flatten") {
        L.kind = SRCFD_LAYER_FLATTEN;
      } else if (lc == "Reshape") {
        L.kind = SRCFD_LAYER_RESHAPE;
        std::vector<int> t = int_list(c.at("target_shape"));
        if (t.size() != 3) throw std::runtime_error("Reshape rank unsupported");
        for (int i = 0; i < 3; ++i) L.reshape[i] = t[i];
      } else throw std::runtime_error("layer class '" + lc + "' unsupported");
      if (has_w) {
        // sub-model file: model_weights/<layer> holds weight_names "<layer>/kernel", ... relative to itself; whole-model
        // file: model_weights/<sub-model> holds the names of all its layers' weights, the same relative paths
        h5lite::Node* lg = nested ? mw : mw->child(lname);
        if (!lg) throw std::runtime_error("model_weights/" + lname + " missing");
        const h5lite::Attr* wn_all = lg->attr("weight_names");
        if (!wn_all || wn_all->strings.empty()) throw std::runtime_error("layer '" + lname + "' has no weight_names");
        h5lite::Attr wn_own;
        if (nested) {
          for (auto& nm : wn_all->strings) if (nm.rfind(lname + "/", 0) == 0) wn_own.strings.push_back(nm);
          if (wn_own.strings.empty()) throw std::runtime_error("layer '" + lname + "' has no weights in its sub-model's weight_names");
        }
        const h5lite::Attr* wn = nested ? &wn_own : wn_all;
        bool use_bias = true;
        if (const auto* ub = c.get("use_bias")) use_bias = ub->kind != jsonmin::Value::Bool || ub->b;
        if (wn->strings.size() != (use_bias ? 2u : 1u)) throw std::runtime_error("layer '" + lname + "': unexpected weight count");
        for (size_t wi = 0; wi < wn->strings.size(); ++wi) {
          h5lite::Node* ds = nullptr;
          {  // weight_names entries are paths relative to the layer group
            h5lite::Node* n = lg;
            std::stringstream ss(wn->strings[wi]);
            std::string part;
            while (n && std::getline(ss, part, '/')) if (!part.empty()) n = n->child(part);
            ds = n;
          }
          if (!ds || ds->is_group) throw std::runtime_error("dataset for weight '" + wn->strings[wi] + "' missing");
          uint64_t cnt = 1;
          for (auto d : ds->dims)
            if (__builtin_mul_overflow(cnt, d, &cnt) || cnt > (1ull << 28)) throw std::runtime_error("weight '" + wn->strings[wi] + "' is implausibly large");
          std::vector<float>& dst = wi == 0 ? L.kernel : L.bias;
          dst.resize(cnt);
          file->read(ds, dst.data(), cnt * sizeof(float), h5lite::F32);
          if (wi == 0) {
            if (L.kind == SRCFD_LAYER_DENSE) {
              if (ds->dims.size() != 2) throw std::runtime_error("Dense kernel rank");
              L.cin = (int)ds->dims[0];
              if ((int)ds->dims[1] != L.cout) throw std::runtime_error("Dense kernel units mismatch");
            } else {
              if (ds->dims.size() != 4 || (int)ds->dims[0] != L.kh || (int)ds->dims[1] != L.kw) throw std::runtime_error("conv kernel shape mismatch");
              int a = (int)ds->dims[2], b = (int)ds->dims[3];
              if (L.kind == SRCFD_LAYER_CONV2D) { L.cin = a; if (b != L.cout) throw std::runtime_error("Conv2D filters mismatch"); }
              else { L.cin = b; if (a != L.cout) throw std::runtime_error("Conv2DTranspose filters mismatch"); }
            }
          }
        }
        if (L.bias.empty()) L.bias.assign(L.cout, 0.f);
      }
      m.layers.push_back(std::move(L));
    }
    sub.count = (int)m.layers.size() - sub.first;
    if (sub.count == 0) throw std::runtime_error("model has no layers");
    if (m.subs.empty()) {
      for (int i = 0; i < 3; ++i) m.in_shape[i] = sub_in[i];
    } else {
      m.infer_shapes();  // shapes of what precedes; check the seam
      const int* prev = m.layers[sub.first - 1].out_shape;
      if (prev[0] * prev[1] * prev[2] != sub_in[0] * sub_in[1] * sub_in[2])
        throw std::runtime_error("sub-model '" + sub.name + "' expects " + std::to_string(sub_in[0] * sub_in[1] * sub_in[2]) +
                                 " inputs but the preceding model yields " + std::to_string(prev[0] * prev[1] * prev[2]));
    }
    m.subs.push_back(sub);
}

void append_h5_submodel(ModelDesc& m, const std::string& path) {
  {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw FileError{SRCFD_ENOENT, "model file '" + path + "' not found"};
    std::fclose(f);
  }
  try {
    auto file = h5lite::File::open(path);
    const h5lite::Attr* cfg = file->root()->attr("model_config");
    if (!cfg || cfg->strings.empty()) throw std::runtime_error("no model_config attribute (not a legacy Keras .h5 model)");
    jsonmin::Value root = jsonmin::parse(cfg->strings[0]);
    const std::string cls = root.at("class_name").str;
    if (cls != "Functional" && cls != "Sequential" && cls != "Model") throw std::runtime_error("model class '" + cls + "' unsupported");
    h5lite::Node* mw = file->find("model_weights");
    if (!mw) throw std::runtime_error("no model_weights group");
    parse_submodel(m, root.at("config"), file.get(), mw, false);
    m.infer_shapes();
  } catch (const FileError&) {
    throw;
  } catch (const std::exception& e) {
    throw FileError{SRCFD_EIO, "cannot load model '" + path + "': " + e.what()};
  }
}

// ---------------------------------------------------------------------------
// save (sr-ae-conv.ipynb:c584-585 `encoder.save(...h5)` layout)
// ---------------------------------------------------------------------------
static const char* DT = "{\"module\": \"keras\", \"class_name\": \"DTypePolicy\", \"config\": {\"name\": \"float32\"}, \"registered_name\": null}";
static const char* KINIT = "{\"module\": \"keras.initializers\", \"class_name\": \"GlorotUniform\", \"config\": {\"seed\": null}, \"registered_name\": null}";
static const char* BINIT = "{\"module\": \"keras.initializers\", \"class_name\": \"Zeros\", \"config\": {}, \"registered_name\": null}";

static std::string shape_json(const int s[3], bool flat) {
  std::ostringstream o;
  if (flat) o << "[null, " << s[0] * s[1] * s[2] << "]";
  else o << "[null, " << s[0] << ", " << s[1] << ", " << s[2] << "]";
  return o.str();
}

static std::string layer_json(const Layer& L, const std::string& prev_name, bool in_flat) {
  std::ostringstream o;
  const char* cls = L.kind == SRCFD_LAYER_CONV2D ? "Conv2D" : L.kind == SRCFD_LAYER_CONV2D_TRANSPOSE ? "Conv2DTranspose"
                  : L.kind == SRCFD_LAYER_DENSE ? "Dense" : L.kind == SRCFD_LAYER_FLATTEN ? "Flatten" : "Reshape";
  o << "{\"class_name\": \"" << cls << "\", \"config\": {\"name\": \"" << L.name << "\", \"trainable\": true, \"dtype\": " << DT;
  if (L.kind == SRCFD_LAYER_CONV2D || L.kind == SRCFD_LAYER_CONV2D_TRANSPOSE) {
    o << ", \"filters\": " << L.cout << ", \"kernel_size\": [" << L.kh << ", " << L.kw << "], \"strides\": [" << L.stride << ", " << L.stride
      << "], \"padding\": \"" << (L.same ? "same" : "valid") << "\", \"data_format\": \"channels_last\", \"dilation_rate\": [1, 1]";
    if (L.kind == SRCFD_LAYER_CONV2D) o << ", \"groups\": 1";
    o << ", \"activation\": \"" << act_name(L.act) << "\", \"use_bias\": true, \"kernel_initializer\": " << KINIT << ", \"bias_initializer\": " << BINIT
      << ", \"kernel_regularizer\": null, \"bias_regularizer\": null, \"activity_regularizer\": null, \"kernel_constraint\": null, \"bias_constraint\": null";
    if (L.kind == SRCFD_LAYER_CONV2D_TRANSPOSE) o << ", \"output_padding\": null";
  } else if (L.kind == SRCFD_LAYER_DENSE) {
    o << ", \"units\": " << L.cout << ", \"activation\": \"" << act_name(L.act) << "\", \"use_bias\": true, \"kernel_initializer\": " << KINIT
      << ", \"bias_initializer\": " << BINIT << ", \"kernel_regularizer\": null, \"bias_regularizer\": null, \"kernel_constraint\": null, \"bias_constraint\": null";
  } else if (L.kind == SRCFD_LAYER_FLATTEN) {
    o << ", \"data_format\": \"channels_last\"";
  } else {
    o << ", \"target_shape\": [" << L.reshape[0] << ", " << L.reshape[1] << ", " << L.reshape[2] << "]";
  }
  o << "}, \"name\": \"" << L.name << "\", \"inbound_nodes\": [{\"args\": [{\"class_name\": \"__keras_tensor__\", \"config\": {\"shape\": "
    << shape_json(L.in_shape, in_flat) << ", \"dtype\": \"float32\", \"keras_history\": [\"" << prev_name << "\", 0, 0]}}], \"kwargs\": {}}]}";
  return o.str();
}

// {"name": ..., "trainable": true, "layers": [...], "input_layers": ..., "output_layers": ...} of sub-model si
static std::string submodel_config_json(const ModelDesc& m, int si, std::vector<std::string>* layer_names_out) {
  const SubModel& sub = m.subs[si];
  const Layer& first = m.layers[sub.first];
  bool flat_in = first.in_shape[0] == 1 && first.in_shape[1] == 1 && first.kind == SRCFD_LAYER_DENSE;
  std::ostringstream cfg;
  cfg << "{\"name\": \"" << sub.name << "\", \"trainable\": true, \"layers\": [";
  cfg << "{\"class_name\": \"InputLayer\", \"config\": {\"batch_shape\": " << shape_json(first.in_shape, flat_in)
      << ", \"dtype\": \"float32\", \"sparse\": false, \"name\": \"" << sub.input_name << "\"}, \"name\": \"" << sub.input_name
      << "\", \"inbound_nodes\": []}";
  std::string prev = sub.input_name;
  bool flat = flat_in;
  if (layer_names_out) layer_names_out->push_back(sub.input_name);
  for (int i = 0; i < sub.count; ++i) {
    const Layer& L = m.layers[sub.first + i];
    cfg << ", " << layer_json(L, prev, flat);
    prev = L.name;
    flat = L.kind == SRCFD_LAYER_DENSE || L.kind == SRCFD_LAYER_FLATTEN;
    if (layer_names_out) layer_names_out->push_back(L.name);
  }
  cfg << "], \"input_layers\": [[\"" << sub.input_name << "\", 0, 0]], \"output_layers\": [[\"" << prev << "\", 0, 0]]}";
  return cfg.str();
}

static std::vector<uint64_t> kernel_dims(const Layer& L) {
  if (L.kind == SRCFD_LAYER_DENSE) return {(uint64_t)L.cin, (uint64_t)L.cout};
  if (L.kind == SRCFD_LAYER_CONV2D) return {(uint64_t)L.kh, (uint64_t)L.kw, (uint64_t)L.cin, (uint64_t)L.cout};
  return {(uint64_t)L.kh, (uint64_t)L.kw, (uint64_t)L.cout, (uint64_t)L.cin};
}

void save_h5_submodel(const ModelDesc& m, int si, const std::string& path) {
  if (si < 0 || si >= (int)m.subs.size()) throw FileError{SRCFD_EINVAL, "no such sub-model"};
  const SubModel& sub = m.subs[si];
  try {
    auto f = h5lite::File::create();
    std::vector<std::string> layer_names;
    std::ostringstream cfg;
    cfg << "{\"class_name\": \"Functional\", \"config\": " << submodel_config_json(m, si, &layer_names) << "}";

    auto str_attr = [](const std::vector<std::string>& s, bool scalar, bool utf8) {
      h5lite::Attr a;
      a.dtype = h5lite::STR; a.scalar = scalar; a.utf8 = utf8; a.strings = s;
      if (!scalar) a.dims = {s.size()};
      return a;
    };
    auto empty_attr = [] {  // h5py stores an empty list as float64 shape (0,)
      h5lite::Attr a;
      a.dtype = h5lite::F64; a.scalar = false; a.dims = {0};
      return a;
    };
    f->root()->attrs.push_back({"backend", str_attr({"tensorflow"}, true, true)});
    f->root()->attrs.push_back({"keras_version", str_attr({"3.8.0"}, true, true)});
    f->root()->attrs.push_back({"model_config", str_attr({cfg.str()}, true, false)});
    h5lite::Node* mw = f->make_group("model_weights");
    mw->attrs.push_back({"backend", str_attr({"tensorflow"}, true, false)});
    mw->attrs.push_back({"keras_version", str_attr({"3.8.0"}, true, false)});
    mw->attrs.push_back({"layer_names", str_attr(layer_names, false, false)});
    f->make_group("model_weights/" + sub.input_name)->attrs.push_back({"weight_names", empty_attr()});
    f->make_group("model_weights/top_level_model_weights")->attrs.push_back({"weight_names", empty_attr()});
    for (int i = 0; i < sub.count; ++i) {
      const Layer& L = m.layers[sub.first + i];
      h5lite::Node* g = f->make_group("model_weights/" + L.name);
      if (L.kernel.empty()) { g->attrs.push_back({"weight_names", empty_attr()}); continue; }
      g->attrs.push_back({"weight_names", str_attr({L.name + "/kernel", L.name + "/bias"}, false, false)});
      const std::vector<uint64_t> kd = kernel_dims(L);
      const std::string base = "model_weights/" + L.name + "/" + L.name + "/";
      f->make_dataset(base + "kernel", h5lite::F32, kd, L.kernel.data());
      f->make_dataset(base + "bias", h5lite::F32, {(uint64_t)L.cout}, L.bias.data());
    }
    f->save(path);
  } catch (const FileError&) {
    throw;
  } catch (const std::exception& e) {
    throw FileError{SRCFD_EIO, "cannot save model '" + path + "': " + e.what()};
  }
}

// ---------------------------------------------------------------------------
// whole-model file: `superres_model.save("superres_{lr}to{hr}_vanilla_ae_{suffix}.h5")` (sr-ae-conv.ipynb:c586)
//
// LAYOUT UNPINNED: the reference's three superres_*.h5 files are absent from its checkout (.MISSING_LARGE_BLOBS:29-31) and
// Keras is not installable here, so this follows what Keras 3.8's legacy-H5 saver (keras/src/legacy/saving/
// legacy_h5_format.py: save_model_to_hdf5 -> save_weights_to_hdf5_group) does for a subclassed Model whose `layers` are
// the two Functional sub-models:
//   /            attrs backend, keras_version, model_config = {"class_name": "SuperResolutionAE", "config": {...}}
//   /model_weights                      attrs backend, keras_version, layer_names = [<encoder name>, <decoder name>]
//   /model_weights/<sub-model>          attr  weight_names = ["conv2d/kernel", "conv2d/bias", ...]  (variable paths)
//   /model_weights/<sub-model>/<layer>/{kernel,bias}
//   /model_weights/top_level_model_weights   attr weight_names = []
// The subclass's auto-generated config serialises its constructor arguments, i.e. the two sub-models (`encoder_lr`,
// `decoder_hr`); the reader takes the architecture from there when present and otherwise assumes the reference's
// encoder_10 / decoder_400 definition (sr-ae-conv.ipynb:c162-169, c277-287), as `load_weights` into a rebuilt model would.
// ---------------------------------------------------------------------------
void save_h5_whole(const ModelDesc& m, const std::string& path) {
  if (m.subs.size() != 2) throw FileError{SRCFD_EINVAL, "whole-model file needs the encoder + decoder pair"};
  try {
    auto f = h5lite::File::create();
    auto str_attr = [](const std::vector<std::string>& s, bool scalar, bool utf8) {
      h5lite::Attr a;
      a.dtype = h5lite::STR; a.scalar = scalar; a.utf8 = utf8; a.strings = s;
      if (!scalar) a.dims = {s.size()};
      return a;
    };
    auto empty_attr = [] { h5lite::Attr a; a.dtype = h5lite::F64; a.scalar = false; a.dims = {0}; return a; };
    std::ostringstream cfg;
    cfg << "{\"class_name\": \"SuperResolutionAE\", \"config\": {\"name\": \"super_resolution_ae\", \"trainable\": true";
    const char* keys[2] = {"encoder_lr", "decoder_hr"};   // SuperResolutionAE.__init__(encoder_lr, decoder_hr) (sr-ae-conv.ipynb:c290)
    for (int si = 0; si < 2; ++si)
      cfg << ", \"" << keys[si] << "\": {\"module\": \"keras\", \"class_name\": \"Functional\", \"config\": " << submodel_config_json(m, si, nullptr)
          << ", \"registered_name\": \"Functional\"}";
    cfg << "}}";
    f->root()->attrs.push_back({"backend", str_attr({"tensorflow"}, true, true)});
    f->root()->attrs.push_back({"keras_version", str_attr({"3.8.0"}, true, true)});
    f->root()->attrs.push_back({"model_config", str_attr({cfg.str()}, true, false)});
    h5lite::Node* mw = f->make_group("model_weights");
    mw->attrs.push_back({"backend", str_attr({"tensorflow"}, true, false)});
    mw->attrs.push_back({"keras_version", str_attr({"3.8.0"}, true, false)});
    mw->attrs.push_back({"layer_names", str_attr({m.subs[0].name, m.subs[1].name}, false, false)});
    f->make_group("model_weights/top_level_model_weights")->attrs.push_back({"weight_names", empty_attr()});
    for (int si = 0; si < 2; ++si) {
      const SubModel& sub = m.subs[si];
      h5lite::Node* g = f->make_group("model_weights/" + sub.name);
      std::vector<std::string> names;
      for (int i = 0; i < sub.count; ++i) {
        const Layer& L = m.layers[sub.first + i];
        if (L.kernel.empty()) continue;
        names.push_back(L.name + "/kernel");
        names.push_back(L.name + "/bias");
        const std::string base = "model_weights/" + sub.name + "/" + L.name + "/";
        f->make_dataset(base + "kernel", h5lite::F32, kernel_dims(L), L.kernel.data());
        f->make_dataset(base + "bias", h5lite::F32, {(uint64_t)L.cout}, L.bias.data());
      }
      g->attrs.push_back({"weight_names", str_attr(names, false, false)});
    }
    f->save(path);
  } catch (const FileError&) {
    throw;
  } catch (const std::exception& e) {
    throw FileError{SRCFD_EIO, "cannot save model '" + path + "': " + e.what()};
  }
}

// The reference architecture as Functional configs, for whole-model files whose model_config does not carry the sub-models.
static std::string reference_submodel_config(const std::string& name, h5lite::Node* g) {
  auto shape_of = [&](const std::string& layer, const char* what) -> std::vector<uint64_t> {
    h5lite::Node* l = g->child(layer);
    h5lite::Node* d = l ? l->child(what) : nullptr;
    if (!d || d->is_group) throw std::runtime_error("whole-model file: " + name + "/" + layer + "/" + what + " missing");
    return d->dims;
  };
  auto conv = [&](const std::string& ln, const char* cls, int stride, const char* pad, const char* act) {
    auto kd = shape_of(ln, "kernel");
    if (kd.size() != 4) throw std::runtime_error("whole-model file: kernel rank of " + ln);
    const uint64_t filters = std::string(cls) == "Conv2D" ? kd[3] : kd[2];
    std::ostringstream o;
    o << "{\"class_name\": \"" << cls << "\", \"config\": {\"name\": \"" << ln << "\", \"filters\": " << filters << ", \"kernel_size\": [" << kd[0] << ", "
      << kd[1] << "], \"strides\": [" << stride << ", " << stride << "], \"padding\": \"" << pad << "\", \"activation\": \"" << act << "\"}}";
    return o.str();
  };
  auto dense = [&](const std::string& ln, const char* act) {
    auto kd = shape_of(ln, "kernel");
    if (kd.size() != 2) throw std::runtime_error("whole-model file: kernel rank of " + ln);
    std::ostringstream o;
    o << "{\"class_name\": \"Dense\", \"config\": {\"name\": \"" << ln << "\", \"units\": " << kd[1] << ", \"activation\": \"" << act << "\"}}";
    return o.str();
  };
  // Layer-group NAMES say nothing about strides and padding, and the notebook's other variants (Conv2DTranspose(.., 3, strides=2,
  // padding='same'), encoders with several stride-2 convolutions) use the same names: the fallback accepts a file only when its
  // kernel shapes are exactly encoder_lr's / decoder_hr's (3x3 s2 + 3x3 s1 convolutions; transposed kernels 3,2,2,2,2 with
  // 'valid' padding; 3x3 output convolution; dense widths consistent with a square feature map), otherwise it refuses.
  auto expect = [&](bool ok, const std::string& what) {
    if (!ok) throw std::runtime_error("whole-model file without sub-model configs: architecture not recoverable from this file (" + name + ": " + what +
                                      " differs from the reference encoder_lr / decoder_hr)");
  };
  auto ksize = [&](const std::string& ln, uint64_t k) {
    auto kd = shape_of(ln, "kernel");
    expect(kd.size() == 4 && kd[0] == k && kd[1] == k, ln + " kernel size");
  };
  std::ostringstream o;
  o << "{\"name\": \"" << name << "\", \"layers\": [";
  if (g->child("conv2d") && g->child("latent_vector")) {   // encoder_lr (sr-ae-conv.ipynb:c162-169)
    ksize("conv2d", 3); ksize("conv2d_1", 3);
    expect(!g->child("conv2d_2"), "number of convolutions");
    const uint64_t lat_in = shape_of("dense", "kernel")[0], c2 = shape_of("conv2d_1", "kernel")[3];
    const int side2 = (int)std::lround(std::sqrt((double)lat_in / (double)c2));   // conv output side: stride-2 SAME of the input
    expect((uint64_t)side2 * side2 * c2 == lat_in, "dense input width");
    expect(shape_of("conv2d_1", "kernel")[2] == shape_of("conv2d", "kernel")[3], "conv2d_1 input channels");
    expect(shape_of("latent_vector", "kernel")[0] == shape_of("dense", "kernel")[1], "latent_vector input width");
    o << "{\"class_name\": \"InputLayer\", \"config\": {\"name\": \"" << name << "_input\", \"batch_shape\": [null, " << 2 * side2 << ", " << 2 * side2 << ", "
      << shape_of("conv2d", "kernel")[2] << "]}}, " << conv("conv2d", "Conv2D", 2, "same", "swish") << ", " << conv("conv2d_1", "Conv2D", 1, "same", "swish")
      << ", {\"class_name\": \"Flatten\", \"config\": {\"name\": \"flatten\"}}, " << dense("dense", "swish") << ", " << dense("latent_vector", "linear");
  } else if (g->child("dense_1") && g->child("conv2d_transpose")) {   // decoder_hr (sr-ae-conv.ipynb:c277-287)
    const uint64_t units = shape_of("dense_1", "kernel")[1], cin = shape_of("conv2d_transpose", "kernel")[3];
    const int side = (int)std::lround(std::sqrt((double)units / (double)cin));
    expect((uint64_t)side * side * cin == units, "dense_1 units");
    o << "{\"class_name\": \"InputLayer\", \"config\": {\"name\": \"" << name << "_input\", \"batch_shape\": [null, " << shape_of("dense_1", "kernel")[0] << "]}}, "
      << dense("dense_1", "swish") << ", {\"class_name\": \"Reshape\", \"config\": {\"name\": \"reshape\", \"target_shape\": [" << side << ", " << side << ", " << cin << "]}}";
    for (int i = 0;; ++i) {
      const std::string ln = i == 0 ? "conv2d_transpose" : "conv2d_transpose_" + std::to_string(i);
      if (!g->child(ln)) { expect(i == 5, "number of transposed convolutions"); break; }
      ksize(ln, i == 0 ? 3 : 2);
      o << ", " << conv(ln, "Conv2DTranspose", 2, "valid", "swish");
    }
    std::string outl;
    for (auto& c : g->children) if (c.first.rfind("output_image", 0) == 0) outl = c.first;
    if (outl.empty()) throw std::runtime_error("whole-model file: no output_image_* layer");
    ksize(outl, 3);
    o << ", " << conv(outl, "Conv2D", 1, "same", "linear");
  } else throw std::runtime_error("whole-model file: sub-model '" + name + "' is neither the reference encoder nor decoder");
  o << "]}";
  return o.str();
}

void append_h5_whole(ModelDesc& m, const std::string& path) {
  {
    FILE* f = std::fopen(path.c_str(), "rb");
    if (!f) throw FileError{SRCFD_ENOENT, "model file '" + path + "' not found"};
    std::fclose(f);
  }
  try {
    auto file = h5lite::File::open(path);
    h5lite::Node* mw = file->find("model_weights");
    if (!mw) throw std::runtime_error("no model_weights group");
    const h5lite::Attr* ln = mw->attr("layer_names");
    if (!ln || ln->strings.size() != 2) throw std::runtime_error("not a whole-model file: model_weights/layer_names must name the two sub-models");
    jsonmin::Value root;
    bool have_cfg = false;
    if (const h5lite::Attr* cfg = file->root()->attr("model_config"))
      if (!cfg->strings.empty()) { root = jsonmin::parse(cfg->strings[0]); have_cfg = true; }
    for (const std::string& sname : ln->strings) {
      h5lite::Node* g = mw->child(sname);
      if (!g || !g->is_group) throw std::runtime_error("model_weights/" + sname + " missing");
      const jsonmin::Value* sub_cfg = nullptr;
      if (have_cfg)
        if (const jsonmin::Value* c = root.get("config"))
          if (c->kind == jsonmin::Value::Obj)
            for (auto& kv : c->obj) {
              const jsonmin::Value& v = kv.second;
              if (v.kind != jsonmin::Value::Obj) continue;
              const jsonmin::Value* cc = v.get("config");
              if (cc && cc->kind == jsonmin::Value::Obj && cc->get("layers") && cc->get("name") && cc->at("name").str == sname) sub_cfg = cc;
            }
      if (sub_cfg) parse_submodel(m, *sub_cfg, file.get(), g, true);
      else {
        jsonmin::Value ref = jsonmin::parse(reference_submodel_config(sname, g));
        parse_submodel(m, ref, file.get(), g, true);
      }
    }
  } catch (const FileError&) {
    throw;
  } catch (const std::exception& e) {
    throw FileError{SRCFD_EIO, "cannot load model '" + path + "': " + e.what()};
  }
}

}  // namespace srcfd
