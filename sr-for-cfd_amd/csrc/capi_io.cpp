// File-format half of the C ABI (include/srcfd.h): stats txt and the HDF5
// subset.  Host-only code.
#include <cerrno>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>

#include "../../include/srcfd.h"
#include "abi_guard.h"
#include "h5lite.h"

namespace srcfd {
void set_error(const std::string& m);
}
using srcfd::set_error;

struct srcfd_h5 {
  std::unique_ptr<h5lite::File> f;
};
struct srcfd_h5w {
  std::unique_ptr<h5lite::File> f;
};

static int copy_out(const std::string& s, char* buf, size_t buf_len, size_t* needed) {
  if (needed) *needed = s.size() + 1;
  if (buf && buf_len) {
    size_t n = std::min(buf_len - 1, s.size());
    std::memcpy(buf, s.data(), n);
    buf[n] = 0;
  }
  return SRCFD_OK;
}

static bool file_exists(const char* p) {
  FILE* f = std::fopen(p, "rb");
  if (!f) return false;
  std::fclose(f);
  return true;
}

extern "C" {

// ---- stats ----------------------------------------------------------------
int srcfd_stats_load(const char* path, int lr_dim, int hr_dim, double out[12]) {
  return srcfd::abi_guard("srcfd_stats_load", [&]() -> int {
    if (!path || !out) { set_error("srcfd_stats_load: bad arguments"); return SRCFD_EINVAL; }
    std::ifstream in(path);
    if (!in) { set_error(std::string("stats file '") + path + "' not found"); return SRCFD_ENOENT; }
    std::map<std::string, double> stats;
    std::string line;
    while (std::getline(in, line)) {
      // strip, skip comments/blank, accept exactly two whitespace-separated tokens
      // (PyCFD_ML_accelerated.py:789-797)
      std::istringstream ss(line);
      std::string a, b, c;
      if (!(ss >> a)) continue;
      if (a[0] == '#') continue;
      if (!(ss >> b)) continue;
      if (ss >> c) continue;
      char* end = nullptr;
      errno = 0;
      double v = std::strtod(b.c_str(), &end);
      if (end == b.c_str() || *end) { set_error("stats file '" + std::string(path) + "': could not convert '" + b + "' to float"); return SRCFD_EIO; }
      stats[a] = v;
    }
    static const char* comps[3] = {"u", "v", "p"};
    int dims[2] = {lr_dim, hr_dim};
    for (int d = 0; d < 2; ++d)
      for (int c = 0; c < 3; ++c)
        for (int k = 0; k < 2; ++k) {
          std::string key = std::string(k ? "std" : "mean") + std::to_string(dims[d]) + "_" + comps[c];
          auto it = stats.find(key);
          if (it == stats.end()) { set_error(key); return SRCFD_EKEY; }
          out[d * 6 + c * 2 + k] = it->second;
        }
    return SRCFD_OK;
  });
}

int srcfd_stats_save(const char* path, int lr_dim, int hr_dim, const double in[12]) {
  return srcfd::abi_guard("srcfd_stats_save", [&]() -> int {
    if (!path || !in) { set_error("srcfd_stats_save: bad arguments"); return SRCFD_EINVAL; }
    FILE* f = std::fopen(path, "w");
    if (!f) { set_error(std::string("cannot create '") + path + "'"); return SRCFD_EIO; }
    // sr-ae-conv.ipynb:c589-603
    std::fprintf(f, "# Component-specific standardization statistics\n# Format: mean<resolution>_<component> value\n");
    static const char* comps[3] = {"u", "v", "p"};
    int dims[2] = {lr_dim, hr_dim};
    for (int d = 0; d < 2; ++d)
      for (int c = 0; c < 3; ++c) {
        std::fprintf(f, "mean%d_%s %.17g\n", dims[d], comps[c], in[d * 6 + c * 2]);
        std::fprintf(f, "std%d_%s %.17g\n", dims[d], comps[c], in[d * 6 + c * 2 + 1]);
      }
    return std::fclose(f) == 0 ? SRCFD_OK : SRCFD_EIO;
  });
}

// ---- h5 read --------------------------------------------------------------
int srcfd_h5_open(const char* path, srcfd_h5** out) {
  return srcfd::abi_guard("srcfd_h5_open", [&]() -> int {
    if (!path || !out) { set_error("srcfd_h5_open: bad arguments"); return SRCFD_EINVAL; }
    *out = nullptr;
    if (!file_exists(path)) { set_error(std::string("file '") + path + "' not found"); return SRCFD_ENOENT; }
    try {
      std::unique_ptr<srcfd_h5> h(new srcfd_h5());
      h->f = h5lite::File::open(path);
      *out = h.release();
      return SRCFD_OK;
    } catch (const std::exception& e) {
      set_error(e.what());
      return SRCFD_EIO;
    }
  });
}

void srcfd_h5_close(srcfd_h5* f) { delete f; }

int srcfd_h5_kind(srcfd_h5* f, const char* path) {
  return srcfd::abi_guard("srcfd_h5_kind", [&]() -> int {
    if (!f || !path) return 0;
    h5lite::Node* n = f->f->find(path);
    return !n ? 0 : (n->is_group ? 1 : 2);
  });
}

int srcfd_h5_list(srcfd_h5* f, const char* group, char* buf, size_t buf_len, size_t* needed) {
  return srcfd::abi_guard("srcfd_h5_list", [&]() -> int {
    if (!f || !group) { set_error("bad arguments"); return SRCFD_EINVAL; }
    h5lite::Node* n = f->f->find(group);
    if (!n || !n->is_group) { set_error(std::string("no group '") + group + "'"); return SRCFD_EKEY; }
    std::string s;
    for (auto& c : n->children) { if (!s.empty()) s += '\n'; s += c.first; }
    return copy_out(s, buf, buf_len, needed);
  });
}

int srcfd_h5_dataset_info(srcfd_h5* f, const char* path, int* dtype, int* rank, uint64_t dims[8]) {
  return srcfd::abi_guard("srcfd_h5_dataset_info", [&]() -> int {
    if (!f || !path) { set_error("bad arguments"); return SRCFD_EINVAL; }
    h5lite::Node* n = f->f->find(path);
    if (!n || n->is_group) { set_error(std::string("no dataset '") + path + "'"); return SRCFD_EKEY; }
    if (n->dims.size() > 8) { set_error("rank > 8"); return SRCFD_EIO; }
    if (dtype) *dtype = (int)n->dtype;
    if (rank) *rank = (int)n->dims.size();
    if (dims) for (size_t i = 0; i < n->dims.size(); ++i) dims[i] = n->dims[i];
    return SRCFD_OK;
  });
}

int srcfd_h5_read(srcfd_h5* f, const char* path, void* dst, size_t dst_bytes, int as_dtype) {
  return srcfd::abi_guard("srcfd_h5_read", [&]() -> int {
    if (!f || !path || !dst) { set_error("bad arguments"); return SRCFD_EINVAL; }
    h5lite::Node* n = f->f->find(path);
    if (!n || n->is_group) { set_error(std::string("no dataset '") + path + "'"); return SRCFD_EKEY; }
    try {
      f->f->read(n, dst, dst_bytes, (h5lite::DType)as_dtype);
      return SRCFD_OK;
    } catch (const std::exception& e) {
      set_error(e.what());
      return SRCFD_EIO;
    }
  });
}

static const h5lite::Attr* find_attr(srcfd_h5* f, const char* obj, const char* name) {
  h5lite::Node* n = f->f->find(obj ? obj : "/");
  if (!n) { set_error(std::string("no object '") + (obj ? obj : "/") + "'"); return nullptr; }
  const h5lite::Attr* a = n->attr(name);
  if (!a) set_error(std::string("no attribute '") + name + "'");
  return a;
}

int srcfd_h5_attr_string(srcfd_h5* f, const char* obj, const char* name, char* buf, size_t buf_len, size_t* needed) {
  return srcfd::abi_guard("srcfd_h5_attr_string", [&]() -> int {
    if (!f || !name) { set_error("bad arguments"); return SRCFD_EINVAL; }
    const h5lite::Attr* a = find_attr(f, obj, name);
    if (!a) return SRCFD_EKEY;
    std::string s;
    if (a->dtype == h5lite::STR)
      for (size_t i = 0; i < a->strings.size(); ++i) { if (i) s += '\n'; s += a->strings[i]; }
    else if (!a->raw.empty()) { set_error("attribute is numeric"); return SRCFD_EINVAL; }
    return copy_out(s, buf, buf_len, needed);
  });
}

int srcfd_h5_attr_numeric(srcfd_h5* f, const char* obj, const char* name, double* out, int max_count, int* count) {
  return srcfd::abi_guard("srcfd_h5_attr_numeric", [&]() -> int {
    if (!f || !name || !count) { set_error("bad arguments"); return SRCFD_EINVAL; }
    const h5lite::Attr* a = find_attr(f, obj, name);
    if (!a) return SRCFD_EKEY;
    size_t sz = h5lite::dtype_size(a->dtype);
    if (!sz) { set_error("attribute is not numeric"); return SRCFD_EINVAL; }
    int n = (int)(a->raw.size() / sz);
    *count = n;
    for (int i = 0; i < n && i < max_count && out; ++i) {
      const uint8_t* p = a->raw.data() + (size_t)i * sz;
      switch (a->dtype) {
        case h5lite::F32: { float v; std::memcpy(&v, p, 4); out[i] = v; break; }
        case h5lite::F64: { double v; std::memcpy(&v, p, 8); out[i] = v; break; }
        case h5lite::I32: { int32_t v; std::memcpy(&v, p, 4); out[i] = v; break; }
        case h5lite::I64: { int64_t v; std::memcpy(&v, p, 8); out[i] = (double)v; break; }
        case h5lite::U8: out[i] = *p; break;
        default: break;
      }
    }
    return SRCFD_OK;
  });
}

int srcfd_h5_attr_names(srcfd_h5* f, const char* obj, char* buf, size_t buf_len, size_t* needed) {
  return srcfd::abi_guard("srcfd_h5_attr_names", [&]() -> int {
    if (!f) { set_error("bad arguments"); return SRCFD_EINVAL; }
    h5lite::Node* n = f->f->find(obj ? obj : "/");
    if (!n) { set_error("no such object"); return SRCFD_EKEY; }
    std::string s;
    for (auto& a : n->attrs) { if (!s.empty()) s += '\n'; s += a.first; }
    return copy_out(s, buf, buf_len, needed);
  });
}

// ---- h5 write -------------------------------------------------------------
int srcfd_h5w_create(srcfd_h5w** out) {
  return srcfd::abi_guard("srcfd_h5w_create", [&]() -> int {
    if (!out) return SRCFD_EINVAL;
    std::unique_ptr<srcfd_h5w> w(new srcfd_h5w());
    w->f = h5lite::File::create();
    *out = w.release();
    return SRCFD_OK;
  });
}
void srcfd_h5w_free(srcfd_h5w* w) { delete w; }

#define H5W_TRY(body)                         \
  try { body; return SRCFD_OK; }              \
  catch (const std::exception& e) { set_error(e.what()); return SRCFD_EIO; }

int srcfd_h5w_group(srcfd_h5w* w, const char* path) {
  return srcfd::abi_guard("srcfd_h5w_group", [&]() -> int {
    if (!w || !path) { set_error("bad arguments"); return SRCFD_EINVAL; }
    H5W_TRY(w->f->make_group(path))
  });
}

int srcfd_h5w_dataset(srcfd_h5w* w, const char* path, int dtype, int rank, const uint64_t* dims, const void* data) {
  return srcfd::abi_guard("srcfd_h5w_dataset", [&]() -> int {
    if (!w || !path || rank < 0 || rank > 8 || (rank && !dims)) { set_error("bad arguments"); return SRCFD_EINVAL; }
    std::vector<uint64_t> d(dims, dims + rank);
    uint64_t n = 1;
    for (auto x : d) n *= x;
    if (n && !data) { set_error("null data"); return SRCFD_EINVAL; }
    H5W_TRY(w->f->make_dataset(path, (h5lite::DType)dtype, d, data))
  });
}

static h5lite::Node* wnode(srcfd_h5w* w, const char* obj) {
  h5lite::Node* n = w->f->find(obj ? obj : "/");
  if (!n) n = w->f->make_group(obj);
  return n;
}

static void set_attr(h5lite::Node* n, const char* name, h5lite::Attr a) {
  for (auto& kv : n->attrs) if (kv.first == name) { kv.second = std::move(a); return; }
  n->attrs.emplace_back(name, std::move(a));
}

int srcfd_h5w_attr_strings(srcfd_h5w* w, const char* obj, const char* name, const char* const* strings, int n_strings, int is_scalar, int utf8) {
  return srcfd::abi_guard("srcfd_h5w_attr_strings", [&]() -> int {
    if (!w || !name || n_strings < 0 || (n_strings && !strings) || (is_scalar && n_strings != 1)) { set_error("bad arguments"); return SRCFD_EINVAL; }
    H5W_TRY({
      h5lite::Attr a;
      a.dtype = h5lite::STR; a.scalar = is_scalar != 0; a.utf8 = utf8 != 0;
      for (int i = 0; i < n_strings; ++i) a.strings.emplace_back(strings[i]);
      if (!a.scalar) a.dims = {(uint64_t)n_strings};
      if (!a.scalar && n_strings == 0) { a.dtype = h5lite::F64; }  // h5py stores [] as float64 (0,)
      set_attr(wnode(w, obj), name, std::move(a));
    })
  });
}

int srcfd_h5w_attr_numeric(srcfd_h5w* w, const char* obj, const char* name, int dtype, const void* value, int count, int is_scalar) {
  return srcfd::abi_guard("srcfd_h5w_attr_numeric", [&]() -> int {
    size_t sz = h5lite::dtype_size((h5lite::DType)dtype);
    if (!w || !name || !sz || count < 0 || (count && !value) || (is_scalar && count != 1)) { set_error("bad arguments"); return SRCFD_EINVAL; }
    H5W_TRY({
      h5lite::Attr a;
      a.dtype = (h5lite::DType)dtype; a.scalar = is_scalar != 0;
      if (!a.scalar) a.dims = {(uint64_t)count};
      a.raw.assign((const uint8_t*)value, (const uint8_t*)value + sz * count);
      set_attr(wnode(w, obj), name, std::move(a));
    })
  });
}

int srcfd_h5w_save(srcfd_h5w* w, const char* path) {
  return srcfd::abi_guard("srcfd_h5w_save", [&]() -> int {
    if (!w || !path) { set_error("bad arguments"); return SRCFD_EINVAL; }
    H5W_TRY(w->f->save(path))
  });
}

}  // extern "C"
