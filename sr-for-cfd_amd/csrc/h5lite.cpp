// h5lite implementation -- see h5lite.h for scope.  Layouts follow the HDF5
// File Format Specification v1/v2 sections III.A-IV.A (superblock v0, v1 B-tree,
// symbol-table nodes, local/global heaps, v1 object headers).
#include "h5lite.h"

#include <algorithm>
#include <cstdio>
#include <cstring>
#include <stdexcept>

namespace h5lite {

static const uint64_t UNDEF = ~0ull;
static const uint8_t SIG[8] = {0x89, 'H', 'D', 'F', '\r', '\n', 0x1a, '\n'};

size_t dtype_size(DType t) {
  switch (t) {
    case F32: case I32: return 4;
    case F64: case I64: return 8;
    case U8: return 1;
    default: return 0;
  }
}

Node* Node::child(const std::string& name) {
  for (auto& c : children)
    if (c.first == name) return c.second.get();
  return nullptr;
}
const Attr* Node::attr(const std::string& name) const {
  for (auto& a : attrs)
    if (a.first == name) return &a.second;
  return nullptr;
}

[[noreturn]] static void fail(const std::string& m) { throw std::runtime_error("h5lite: " + m); }

// product of the dimensions times `elem`, refusing anything that wraps in 64 bits (a crafted file can carry dims whose
// product wraps to a small number: every size check downstream would then see the wrapped value)
static uint64_t checked_bytes(const std::vector<uint64_t>& dims, uint64_t elem, uint64_t* count = nullptr) {
  uint64_t n = 1;
  for (auto d : dims)
    if (__builtin_mul_overflow(n, d, &n)) fail("dataspace too large");
  uint64_t bytes = 0;
  if (__builtin_mul_overflow(n, elem, &bytes)) fail("dataspace too large");
  if (count) *count = n;
  return bytes;
}

// ---------------------------------------------------------------------------
// reader
// ---------------------------------------------------------------------------
struct Parser {
  const std::vector<uint8_t>& b;
  int so = 8, sl = 8;  // size of offsets / lengths
  int budget = 200000;  // objects + B-tree nodes + heap entries visited: a cyclic or self-similar file cannot loop for ever
  void spend() { if (--budget < 0) fail("file structure too large or cyclic"); }
  explicit Parser(const std::vector<uint8_t>& buf) : b(buf) {}

  void need(uint64_t off, uint64_t n) const {
    if (off > b.size() || n > b.size() - off) fail("truncated file (offset " + std::to_string(off) + ")");
  }
  uint64_t u(uint64_t off, int n) const {
    need(off, n);
    uint64_t v = 0;
    for (int i = n - 1; i >= 0; --i) v = (v << 8) | b[off + i];
    return v;
  }
  uint64_t off_at(uint64_t p) const {
    uint64_t v = u(p, so);
    if (so < 8 && v == ((1ull << (8 * so)) - 1)) return UNDEF;
    return v;
  }

  struct TypeInfo { DType t = UNKNOWN; uint32_t size = 0; bool vlen_str = false; bool fixed_str = false; bool utf8 = false; };

  TypeInfo parse_type(uint64_t p) const {
    TypeInfo ti;
    uint8_t cv = (uint8_t)u(p, 1);
    int cls = cv & 0xf;
    uint32_t bits = (uint32_t)u(p + 1, 3);
    ti.size = (uint32_t)u(p + 4, 4);
    switch (cls) {
      case 0:  // fixed point
        if (bits & 1) fail("big-endian integers unsupported");
        if (ti.size == 4) ti.t = I32; else if (ti.size == 8) ti.t = I64; else if (ti.size == 1) ti.t = U8;
        break;
      case 1:  // float
        if (bits & 1) fail("big-endian floats unsupported");
        if (ti.size == 4) ti.t = F32; else if (ti.size == 8) ti.t = F64;
        break;
      case 3:  // fixed string
        ti.t = STR; ti.fixed_str = true; ti.utf8 = ((bits >> 4) & 0xf) == 1;
        break;
      case 9:  // variable length
        if ((bits & 0xf) == 1) { ti.t = STR; ti.vlen_str = true; ti.utf8 = ((bits >> 8) & 0xf) == 1; }
        break;
      default: break;
    }
    return ti;
  }

  void parse_space(uint64_t p, bool& scalar, std::vector<uint64_t>& dims) const {
    int ver = (int)u(p, 1), rank = (int)u(p + 1, 1);
    uint64_t q;
    if (ver == 1) q = p + 8;
    else if (ver == 2) q = p + 4;
    else fail("dataspace version " + std::to_string(ver));
    dims.clear();
    if (rank > 32) fail("dataspace rank " + std::to_string(rank));
    for (int i = 0; i < rank; ++i) dims.push_back(u(q + (uint64_t)i * sl, sl));
    scalar = (rank == 0);
  }

  std::string gheap_string(uint64_t p) const {  // 16-byte vlen descriptor at p
    uint32_t len = (uint32_t)u(p, 4);
    uint64_t col = off_at(p + 4);
    uint32_t idx = (uint32_t)u(p + 4 + so, 4);
    if (col == UNDEF || len == 0) return std::string();
    need(col, 16);
    if (std::memcmp(&b[col], "GCOL", 4)) fail("bad global heap signature");
    uint64_t csize = u(col + 8, sl);
    need(col, csize);  // the whole collection lies inside the file, so `end` cannot wrap
    uint64_t q = col + 8 + sl, end = col + csize;
    while (q + 8 + sl <= end) {
      const_cast<Parser*>(this)->spend();
      uint32_t oidx = (uint32_t)u(q, 2);
      uint64_t osize = u(q + 8, sl);
      if (oidx == idx) {
        if (len > osize) fail("vlen string longer than its heap object");
        need(q + 8 + sl, len);
        return std::string((const char*)&b[q + 8 + sl], len);
      }
      if (oidx == 0) break;
      if (osize > end - q) fail("global heap object runs past its collection");  // also keeps the padded step from wrapping
      q += 8 + sl + ((osize + 7) & ~7ull);  // always > q: forward progress
    }
    fail("global heap object not found");
  }

  Attr parse_attr(uint64_t p) const {
    Attr a;
    int ver = (int)u(p, 1);
    uint32_t nsz = (uint32_t)u(p + 2, 2), tsz = (uint32_t)u(p + 4, 2), ssz = (uint32_t)u(p + 6, 2);
    uint64_t q = p + 8;
    if (ver == 3) q += 1;
    else if (ver != 1 && ver != 2) fail("attribute version " + std::to_string(ver));
    auto pad = [&](uint32_t n) { return ver == 1 ? ((n + 7u) & ~7u) : n; };
    q += pad(nsz);
    uint64_t tp = q; q += pad(tsz);
    uint64_t sp = q; q += pad(ssz);
    TypeInfo ti = parse_type(tp);
    parse_space(sp, a.scalar, a.dims);
    uint64_t n = 1;
    const uint64_t abytes = checked_bytes(a.dims, ti.t == STR ? (ti.vlen_str ? 8u + so : std::max<uint32_t>(ti.size, 1)) : std::max<uint32_t>(ti.size, 1), &n);
    a.dtype = ti.t;
    a.utf8 = ti.utf8;
    if (ti.t != UNKNOWN) need(q, abytes);  // payload bounds before any per-element loop
    if (ti.t == STR) {
      for (uint64_t i = 0; i < n; ++i) {
        if (ti.vlen_str) a.strings.push_back(gheap_string(q + i * (8 + so)));
        else {
          need(q + i * ti.size, ti.size);
          std::string s((const char*)&b[q + i * ti.size], ti.size);
          s.erase(std::find(s.begin(), s.end(), '\0'), s.end());
          a.strings.push_back(s);
        }
      }
    } else if (ti.t != UNKNOWN) {
      need(q, n * ti.size);
      a.raw.assign(b.begin() + q, b.begin() + q + n * ti.size);
    }
    return a;
  }

  static std::string attr_name(const Parser& P, uint64_t p) {
    int ver = (int)P.u(p, 1);
    uint32_t nsz = (uint32_t)P.u(p + 2, 2);
    uint64_t q = p + 8 + (ver == 3 ? 1 : 0);
    P.need(q, nsz);
    std::string s((const char*)&P.b[q], nsz);
    s.erase(std::find(s.begin(), s.end(), '\0'), s.end());
    return s;
  }

  void walk_btree(uint64_t bt, uint64_t heap_data, Node& g, int depth) {
    if (depth > 64) fail("b-tree too deep");
    spend();
    need(bt, 8);
    if (std::memcmp(&b[bt], "TREE", 4)) fail("bad B-tree signature");
    int level = (int)u(bt + 5, 1), used = (int)u(bt + 6, 2);
    uint64_t q = bt + 8 + 2 * so;  // skip siblings
    for (int i = 0; i < used; ++i) {
      uint64_t child = off_at(q + sl + (uint64_t)i * (sl + so));
      if (level > 0) walk_btree(child, heap_data, g, depth + 1);
      else {
        need(child, 8);
        if (std::memcmp(&b[child], "SNOD", 4)) fail("bad symbol node signature");
        int ns = (int)u(child + 6, 2);
        for (int s = 0; s < ns; ++s) {
          uint64_t e = child + 8 + (uint64_t)s * (2 * so + 24);
          uint64_t noff = u(e, so), oh = off_at(e + so);
          uint64_t np = 0;
          if (__builtin_add_overflow(heap_data, noff, &np)) fail("link name offset");
          need(np, 1);
          // NUL-terminated inside the file buffer, never an unbounded strlen past its end
          const void* z = std::memchr(&b[np], 0, b.size() - np);
          if (!z) fail("unterminated link name");
          std::string name((const char*)&b[np], (const char*)z - (const char*)&b[np]);
          g.children.emplace_back(name, parse_object(oh, depth + 1));
        }
      }
    }
  }

  std::shared_ptr<Node> parse_object(uint64_t oh, int depth) {
    if (depth > 64) fail("group nesting too deep");
    spend();
    auto node = std::make_shared<Node>();
    need(oh, 16);
    int ver = (int)u(oh, 1);
    if (ver != 1) fail("object header version " + std::to_string(ver) + " (only v1 / libver earliest supported)");
    int nmsg = (int)u(oh + 2, 2);
    uint64_t hsize = u(oh + 8, 4);
    struct Chunk { uint64_t p, end; };
    std::vector<Chunk> chunks{{oh + 16, oh + 16 + hsize}};
    bool have_layout = false, have_symtab = false, have_links = false;
    TypeInfo ti;
    uint64_t bt = UNDEF, lh = UNDEF;
    int seen = 0;
    for (size_t ci = 0; ci < chunks.size(); ++ci) {
      uint64_t p = chunks[ci].p;
      while (p + 8 <= chunks[ci].end && seen < nmsg) {
        uint32_t type = (uint32_t)u(p, 2), size = (uint32_t)u(p + 2, 2);
        uint64_t d = p + 8;
        need(d, size);
        ++seen;
        switch (type) {
          case 0x01: { bool sc; parse_space(d, sc, node->dims); break; }
          case 0x03: ti = parse_type(d); node->dtype = ti.t; break;
          case 0x08: {
            int lver = (int)u(d, 1);
            if (lver != 3) fail("data layout version " + std::to_string(lver));
            int cls = (int)u(d + 1, 1);
            if (cls == 1) { node->data_addr = off_at(d + 2); node->data_size = u(d + 2 + so, sl); }
            else if (cls == 0) { uint32_t n = (uint32_t)u(d + 2, 2); need(d + 4, n); node->data.assign(b.begin() + d + 4, b.begin() + d + 4 + n); node->data_size = n; node->data_addr = UNDEF; }
            else fail("chunked datasets unsupported");
            have_layout = true;
            break;
          }
          case 0x0C: node->attrs.emplace_back(attr_name(*this, d), parse_attr(d)); break;
          case 0x10: chunks.push_back({off_at(d), off_at(d) + u(d + so, sl)}); break;
          case 0x11: bt = off_at(d); lh = off_at(d + so); have_symtab = true; break;
          case 0x06: {  // link message (compact new-style group)
            int lv = (int)u(d, 1); uint8_t fl = (uint8_t)u(d + 1, 1);
            if (lv != 1) fail("link message version");
            uint64_t q = d + 2;
            int ltype = 0;
            if (fl & 0x08) { ltype = (int)u(q, 1); q += 1; }
            if (fl & 0x04) q += 8;
            if (fl & 0x10) q += 1;
            int lsz = 1 << (fl & 3);
            uint64_t nlen = u(q, lsz); q += lsz;
            need(q, nlen);
            std::string name((const char*)&b[q], nlen); q += nlen;
            if (ltype == 0) node->children.emplace_back(name, parse_object(off_at(q), depth + 1));
            have_links = true;
            break;
          }
          default: break;
        }
        p = d + size;
      }
    }
    if (have_symtab) {
      need(lh, 8 + 2 * sl + so);
      if (std::memcmp(&b[lh], "HEAP", 4)) fail("bad local heap signature");
      uint64_t heap_data = off_at(lh + 8 + 2 * sl);
      walk_btree(bt, heap_data, *node, depth);
    }
    node->is_group = have_symtab || have_links || !have_layout;
    if (!node->is_group) {
      if (ti.t == UNKNOWN || ti.t == STR) node->dtype = UNKNOWN;
      else {
        // every later read copies dims x element size bytes: the storage the layout message names must hold them
        const uint64_t bytes = checked_bytes(node->dims, ti.size);
        if (node->data_addr != UNDEF) need(node->data_addr, bytes);
        else if (bytes > node->data.size()) fail("dataset storage smaller than its dataspace (compact or unallocated layout)");
      }
    }
    return node;
  }
};

std::unique_ptr<File> File::open(const std::string& path) {
  FILE* f = std::fopen(path.c_str(), "rb");
  if (!f) fail("cannot open '" + path + "'");
  std::unique_ptr<File> file(new File());
  std::fseek(f, 0, SEEK_END);
  long n = std::ftell(f);
  std::fseek(f, 0, SEEK_SET);
  file->buf_.resize(n > 0 ? (size_t)n : 0);
  size_t got = n > 0 ? std::fread(file->buf_.data(), 1, (size_t)n, f) : 0;
  std::fclose(f);
  if (got != (size_t)n || n < 96) fail("short read on '" + path + "'");
  Parser P(file->buf_);
  if (std::memcmp(file->buf_.data(), SIG, 8)) fail("'" + path + "' is not an HDF5 file");
  int sver = (int)P.u(8, 1);
  if (sver != 0 && sver != 1) fail("superblock version " + std::to_string(sver) + " unsupported (need libver earliest)");
  P.so = (int)P.u(13, 1);
  P.sl = (int)P.u(14, 1);
  uint64_t q = 24 + (sver == 1 ? 4 : 0);
  uint64_t base = P.off_at(q);
  if (base != 0) fail("non-zero base address unsupported");
  uint64_t rootent = q + 4 * (uint64_t)P.so;
  uint64_t root_oh = P.off_at(rootent + P.so);
  file->root_ = P.parse_object(root_oh, 0);
  return file;
}

Node* File::find(const std::string& path) {
  Node* n = root_.get();
  size_t i = 0;
  while (i < path.size() && n) {
    while (i < path.size() && path[i] == '/') ++i;
    size_t j = path.find('/', i);
    if (j == std::string::npos) j = path.size();
    if (j > i) n = n->child(path.substr(i, j - i));
    i = j;
  }
  return n;
}

template <typename S, typename D>
static void conv(const uint8_t* src, void* dst, uint64_t n) {
  D* d = (D*)dst;
  for (uint64_t i = 0; i < n; ++i) { S v; std::memcpy(&v, src + i * sizeof(S), sizeof(S)); d[i] = (D)v; }
}

void File::read(Node* ds, void* dst, size_t dst_bytes, DType as) {
  if (!ds || ds->is_group) fail("read: not a dataset");
  size_t ssz = dtype_size(ds->dtype), dsz = dtype_size(as);
  if (!ssz || !dsz) fail("read: unsupported dtype");
  uint64_t n = 1;
  const uint64_t sbytes = checked_bytes(ds->dims, ssz, &n), dbytes = checked_bytes(ds->dims, dsz);
  if (dst_bytes < dbytes) fail("read: destination too small");
  const bool in_file = ds->data_addr != UNDEF && !buf_.empty();
  if (in_file ? (ds->data_addr > buf_.size() || sbytes > buf_.size() - ds->data_addr) : sbytes > ds->data.size())
    fail("read: dataset storage smaller than its dataspace");
  const uint8_t* src = in_file ? buf_.data() + ds->data_addr : ds->data.data();
  if (ds->dtype == as) { std::memcpy(dst, src, n * ssz); return; }
#define CV(S, ST, D, DT) if (ds->dtype == S && as == D) { conv<ST, DT>(src, dst, n); return; }
  CV(F32, float, F64, double) CV(F64, double, F32, float)
  CV(I32, int32_t, I64, int64_t) CV(I64, int64_t, I32, int32_t)
  CV(I32, int32_t, F64, double) CV(I64, int64_t, F64, double)
  CV(I32, int32_t, F32, float) CV(I64, int64_t, F32, float)
  CV(U8, uint8_t, F32, float) CV(U8, uint8_t, F64, double) CV(U8, uint8_t, I32, int32_t) CV(U8, uint8_t, I64, int64_t)
#undef CV
  fail("read: unsupported conversion");
}

// ---------------------------------------------------------------------------
// writer
// ---------------------------------------------------------------------------
std::unique_ptr<File> File::create() { return std::unique_ptr<File>(new File()); }

Node* File::make_group(const std::string& path) {
  Node* n = root_.get();
  size_t i = 0;
  while (i < path.size()) {
    while (i < path.size() && path[i] == '/') ++i;
    size_t j = path.find('/', i);
    if (j == std::string::npos) j = path.size();
    if (j > i) {
      std::string name = path.substr(i, j - i);
      Node* c = n->child(name);
      if (!c) {
        auto nn = std::make_shared<Node>();
        n->children.emplace_back(name, nn);
        c = nn.get();
      } else if (!c->is_group) fail("make_group: '" + name + "' is a dataset");
      n = c;
    }
    i = j;
  }
  return n;
}

Node* File::make_dataset(const std::string& path, DType t, const std::vector<uint64_t>& dims, const void* data) {
  size_t k = path.find_last_of('/');
  Node* g = (k == std::string::npos) ? root_.get() : make_group(path.substr(0, k));
  std::string name = (k == std::string::npos) ? path : path.substr(k + 1);
  if (name.empty()) fail("make_dataset: empty name");
  auto it = std::find_if(g->children.begin(), g->children.end(), [&](auto& c) { return c.first == name; });
  if (it != g->children.end()) g->children.erase(it);
  auto nn = std::make_shared<Node>();
  nn->is_group = false;
  nn->dtype = t;
  nn->dims = dims;
  uint64_t n = 1;
  for (auto d : dims) n *= d;
  size_t sz = dtype_size(t);
  if (!sz) fail("make_dataset: unsupported dtype");
  nn->data.assign((const uint8_t*)data, (const uint8_t*)data + n * sz);
  g->children.emplace_back(name, nn);
  return nn.get();
}

namespace {
struct Out {
  std::vector<uint8_t> b;
  uint64_t alloc(uint64_t n, uint64_t align = 8) {
    uint64_t p = (b.size() + align - 1) / align * align;
    b.resize(p + n, 0);
    return p;
  }
  void put(uint64_t p, uint64_t v, int n) { for (int i = 0; i < n; ++i) b[p + i] = (uint8_t)(v >> (8 * i)); }
  void bytes(uint64_t p, const void* s, size_t n) { if (n) std::memcpy(&b[p], s, n); }
};

struct Msg { uint16_t type; std::vector<uint8_t> d; };

void pad8(std::vector<uint8_t>& v) { while (v.size() % 8) v.push_back(0); }
void app(std::vector<uint8_t>& v, uint64_t x, int n) { for (int i = 0; i < n; ++i) v.push_back((uint8_t)(x >> (8 * i))); }

std::vector<uint8_t> type_msg(DType t, bool utf8) {
  std::vector<uint8_t> v;
  switch (t) {
    case F32: v = {0x11, 0x20, 0x1f, 0x00}; app(v, 4, 4); app(v, 0, 2); app(v, 32, 2); v.insert(v.end(), {23, 8, 0, 23}); app(v, 127, 4); break;
    case F64: v = {0x11, 0x20, 0x3f, 0x00}; app(v, 8, 4); app(v, 0, 2); app(v, 64, 2); v.insert(v.end(), {52, 11, 0, 52}); app(v, 1023, 4); break;
    case I32: v = {0x10, 0x08, 0x00, 0x00}; app(v, 4, 4); app(v, 0, 2); app(v, 32, 2); break;
    case I64: v = {0x10, 0x08, 0x00, 0x00}; app(v, 8, 4); app(v, 0, 2); app(v, 64, 2); break;
    case U8:  v = {0x10, 0x00, 0x00, 0x00}; app(v, 1, 4); app(v, 0, 2); app(v, 8, 2); break;
    case STR:  // variable-length string, null-terminated
      // base type = 1-byte fixed point, exactly what h5py emits for str attrs
      v = {0x19, 0x01, (uint8_t)(utf8 ? 1 : 0), 0x00}; app(v, 16, 4);
      v.insert(v.end(), {0x10, 0x00, 0x00, 0x00}); app(v, 1, 4); app(v, 0, 2); app(v, 8, 2);
      break;
    default: fail("type_msg");
  }
  return v;
}

std::vector<uint8_t> space_msg(bool scalar, const std::vector<uint64_t>& dims) {
  std::vector<uint8_t> v = {1, (uint8_t)(scalar ? 0 : dims.size()), 0, 0, 0, 0, 0, 0};
  if (!scalar) for (auto d : dims) app(v, d, 8);
  return v;
}

struct Writer {
  Out o;
  // one global heap collection for every vlen string in the file
  std::vector<std::string> gstrings;
  uint64_t gcol_addr = 0;

  uint64_t gcol_size() const {
    uint64_t s = 16;
    for (auto& x : gstrings) s += 16 + ((x.size() + 7) & ~7ull);
    s += 16;  // room for the free-space object header
    return (std::max<uint64_t>(s, 4096) + 4095) & ~4095ull;
  }

  void collect(Node& n) {
    std::sort(n.children.begin(), n.children.end(), [](auto& a, auto& b) { return a.first < b.first; });
    for (auto& a : n.attrs)
      if (a.second.dtype == STR) for (auto& s : a.second.strings) gstrings.push_back(s);
    for (auto& c : n.children) collect(*c.second);
  }

  void write_gcol() {
    uint64_t size = gcol_size();
    gcol_addr = o.alloc(size);
    o.bytes(gcol_addr, "GCOL", 4);
    o.put(gcol_addr + 4, 1, 1);
    o.put(gcol_addr + 8, size, 8);
    uint64_t p = gcol_addr + 16;
    for (size_t i = 0; i < gstrings.size(); ++i) {
      o.put(p, i + 1, 2); o.put(p + 2, 1, 2); o.put(p + 8, gstrings[i].size(), 8);
      o.bytes(p + 16, gstrings[i].data(), gstrings[i].size());
      p += 16 + ((gstrings[i].size() + 7) & ~7ull);
    }
    uint64_t left = gcol_addr + size - p;
    o.put(p, 0, 2); o.put(p + 8, left, 8);  // object 0 = free space (size includes its header)
  }

  size_t gnext = 0;  // next string index, same traversal order as collect()

  Msg attr_msg(const std::string& name, const Attr& a) {
    Msg m{0x0C, {}};
    auto tm = type_msg(a.dtype, a.utf8), sm = space_msg(a.scalar, a.dims);
    auto& v = m.d;
    v = {1, 0}; app(v, name.size() + 1, 2); app(v, tm.size(), 2); app(v, sm.size(), 2);
    v.insert(v.end(), name.begin(), name.end()); v.push_back(0); pad8(v);
    v.insert(v.end(), tm.begin(), tm.end()); pad8(v);
    v.insert(v.end(), sm.begin(), sm.end()); pad8(v);
    if (a.dtype == STR) {
      for (auto& s : a.strings) { app(v, s.size(), 4); app(v, gcol_addr, 8); app(v, ++gnext, 4); }
    } else v.insert(v.end(), a.raw.begin(), a.raw.end());
    pad8(v);
    if (v.size() > 0xffff) fail("attribute '" + name + "' too large for a v1 header message");
    return m;
  }

  uint64_t write_header(std::vector<Msg>& msgs) {
    uint64_t total = 0;
    for (auto& m : msgs) { pad8(m.d); total += 8 + m.d.size(); }
    uint64_t oh = o.alloc(16 + total);
    o.put(oh, 1, 1); o.put(oh + 2, msgs.size(), 2); o.put(oh + 4, 1, 4); o.put(oh + 8, total, 4);
    uint64_t p = oh + 16;
    for (auto& m : msgs) {
      o.put(p, m.type, 2); o.put(p + 2, m.d.size(), 2);
      o.bytes(p + 8, m.d.data(), m.d.size());
      p += 8 + m.d.size();
    }
    return oh;
  }

  static const int LEAF_K = 4, INT_K = 16;

  // returns object header address; fills btree/heap for the caller's symbol entry
  uint64_t write_node(Node& n, uint64_t* bt_out, uint64_t* heap_out) {
    std::vector<Msg> msgs;
    // attribute string indices must follow collect() order: attrs of this node
    // first, then children -- so build attr messages before recursing.
    std::vector<Msg> amsgs;
    for (auto& a : n.attrs) amsgs.push_back(attr_msg(a.first, a.second));
    if (!n.is_group) {
      uint64_t da = n.data.empty() ? UNDEF : o.alloc(n.data.size());
      if (!n.data.empty()) o.bytes(da, n.data.data(), n.data.size());
      msgs.push_back({0x01, space_msg(false, n.dims)});
      msgs.push_back({0x03, type_msg(n.dtype, false)});
      msgs.push_back({0x05, {2, 2, 2, 0}});  // fill value v2: late alloc, if-set, undefined
      Msg lay{0x08, {3, 1}}; app(lay.d, da, 8); app(lay.d, n.data.size(), 8);
      msgs.push_back(lay);
      for (auto& m : amsgs) msgs.push_back(m);
      if (bt_out) *bt_out = 0;
      return write_header(msgs);
    }
    std::sort(n.children.begin(), n.children.end(), [](auto& a, auto& b) { return a.first < b.first; });
    struct Ent { uint64_t name_off, oh, bt, heap; bool grp; };
    std::vector<Ent> ents;
    // local heap data: "" at 0, then names
    std::vector<uint8_t> hd(8, 0);
    for (auto& c : n.children) {
      Ent e{};
      e.name_off = hd.size();
      hd.insert(hd.end(), c.first.begin(), c.first.end()); hd.push_back(0); pad8(hd);
      ents.push_back(e);
    }
    for (size_t i = 0; i < n.children.size(); ++i) {
      Node& c = *n.children[i].second;
      ents[i].grp = c.is_group;
      ents[i].oh = write_node(c, &ents[i].bt, &ents[i].heap);
    }
    // free block at the tail keeps the library's heap code on its common path
    uint64_t free_off = hd.size();
    hd.resize(hd.size() + 32, 0);
    { uint64_t one = 1, sz = 32; std::memcpy(&hd[free_off], &one, 8); std::memcpy(&hd[free_off + 8], &sz, 8); }
    uint64_t hdat = o.alloc(hd.size());
    o.bytes(hdat, hd.data(), hd.size());
    uint64_t heap = o.alloc(32);
    o.bytes(heap, "HEAP", 4); o.put(heap + 8, hd.size(), 8); o.put(heap + 16, free_off, 8); o.put(heap + 24, hdat, 8);
    // symbol nodes, 2*LEAF_K entries each
    const size_t per = 2 * LEAF_K;
    size_t nsn = std::max<size_t>(1, (ents.size() + per - 1) / per);
    if (nsn > 2 * INT_K) fail("group with more than " + std::to_string(per * 2 * INT_K) + " members");
    std::vector<uint64_t> snods;
    for (size_t s = 0; s < nsn; ++s) {
      uint64_t sn = o.alloc(8 + per * 40);
      size_t lo = s * per, hi = std::min(ents.size(), lo + per);
      o.bytes(sn, "SNOD", 4); o.put(sn + 4, 1, 1); o.put(sn + 6, hi > lo ? hi - lo : 0, 2);
      for (size_t i = lo; i < hi; ++i) {
        uint64_t e = sn + 8 + (i - lo) * 40;
        o.put(e, ents[i].name_off, 8); o.put(e + 8, ents[i].oh, 8);
        if (ents[i].grp) { o.put(e + 16, 1, 4); o.put(e + 24, ents[i].bt, 8); o.put(e + 32, ents[i].heap, 8); }
      }
      snods.push_back(sn);
    }
    uint64_t bt = o.alloc(24 + (2 * INT_K + 1) * 8 + 2 * INT_K * 8);
    o.bytes(bt, "TREE", 4); o.put(bt + 4, 0, 1); o.put(bt + 5, 0, 1);
    o.put(bt + 6, ents.empty() ? 0 : nsn, 2); o.put(bt + 8, UNDEF, 8); o.put(bt + 16, UNDEF, 8);
    uint64_t q = bt + 24;
    o.put(q, 0, 8); q += 8;  // key 0: offset of ""
    if (!ents.empty())
      for (size_t s = 0; s < nsn; ++s) {
        size_t hi = std::min(ents.size(), (s + 1) * per);
        o.put(q, snods[s], 8); o.put(q + 8, ents[hi - 1].name_off, 8);
        q += 16;
      }
    Msg st{0x11, {}}; app(st.d, bt, 8); app(st.d, heap, 8);
    msgs.push_back(st);
    for (auto& m : amsgs) msgs.push_back(m);
    if (bt_out) *bt_out = bt;
    if (heap_out) *heap_out = heap;
    return write_header(msgs);
  }
};
}  // namespace

void File::save(const std::string& path) {
  Writer w;
  w.o.alloc(96);
  w.collect(*root_);
  // collect() visits attrs of a node before its children; write_node builds the
  // attr messages in the same order, so indices line up.
  w.write_gcol();
  uint64_t bt = 0, heap = 0;
  uint64_t root_oh = w.write_node(*root_, &bt, &heap);
  Out& o = w.o;
  o.bytes(0, SIG, 8);
  o.put(13, 8, 1); o.put(14, 8, 1);
  o.put(16, Writer::LEAF_K, 2); o.put(18, Writer::INT_K, 2);
  o.put(24, 0, 8); o.put(32, UNDEF, 8); o.put(40, o.b.size(), 8); o.put(48, UNDEF, 8);
  o.put(56, 0, 8); o.put(64, root_oh, 8); o.put(72, 1, 4); o.put(80, bt, 8); o.put(88, heap, 8);
  FILE* f = std::fopen(path.c_str(), "wb");
  if (!f) fail("cannot create '" + path + "'");
  size_t n = std::fwrite(o.b.data(), 1, o.b.size(), f);
  if (std::fclose(f) != 0 || n != o.b.size()) fail("short write on '" + path + "'");
}

}  // namespace h5lite
