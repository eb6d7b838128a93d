// Host-only harness for the file-format code (h5lite.cpp, model.cpp, capi_io.cpp): opens every path given on the command
// line through both entry points the product uses -- the generic HDF5 C ABI (srcfd_h5_*: every dataset read with and
// without a dtype conversion, every attribute) and the Keras-H5 model loader (append_h5_submodel) -- and reports how many
// files loaded and how many were refused.  A refusal is an exception turned into a status; anything else (a crash, an
// out-of-bounds read, a hang) is a bug.  Built by `make -C sr-for-cfd_amd/csrc asan` with -fsanitize=address,undefined
// ("sanitizers"); tests/test_io_abi.py runs the mutation corpus through it.
#include <cstdio>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#include "../include/srcfd.h"
#include "../sr-for-cfd_amd/csrc/model.h"

namespace srcfd {
thread_local std::string g_last_error;
void set_error(const std::string& m) { g_last_error = m; }
}  // namespace srcfd

static void walk(srcfd_h5* f, const std::string& path, int depth, long& reads) {
  if (depth > 16) return;
  std::vector<char> buf(1 << 16);
  size_t need = 0;
  if (srcfd_h5_attr_names(f, path.c_str(), buf.data(), buf.size(), &need) == SRCFD_OK) {
    std::stringstream ss(buf.data());
    std::string a;
    while (std::getline(ss, a)) {
      if (a.empty()) continue;
      std::vector<char> v(1 << 16);
      size_t n2 = 0;
      (void)srcfd_h5_attr_string(f, path.c_str(), a.c_str(), v.data(), v.size(), &n2);
      double nv[16]; int cnt = 0;
      (void)srcfd_h5_attr_numeric(f, path.c_str(), a.c_str(), nv, 16, &cnt);
    }
  }
  int dtype = 0, rank = 0;
  uint64_t dims[8];
  if (srcfd_h5_dataset_info(f, path.c_str(), &dtype, &rank, dims) == SRCFD_OK) {
    // small destination buffers on purpose: the library must refuse, not overrun; then one large enough (bounded)
    std::vector<char> small(64);
    (void)srcfd_h5_read(f, path.c_str(), small.data(), small.size(), 0);
    (void)srcfd_h5_read(f, path.c_str(), small.data(), small.size(), 1);
    unsigned long long n = 1;
    bool huge = false;
    for (int i = 0; i < rank; ++i) { if (dims[i] > (1ull << 26) || n * dims[i] > (1ull << 26)) { huge = true; break; } n *= dims[i]; }
    if (!huge) {
      std::vector<char> big(n * 8 + 8);
      if (srcfd_h5_read(f, path.c_str(), big.data(), big.size(), 1) == SRCFD_OK) ++reads;
      (void)srcfd_h5_read(f, path.c_str(), big.data(), big.size(), 0);
      (void)srcfd_h5_read(f, path.c_str(), big.data(), big.size(), 3);
    }
    return;
  }
  if (srcfd_h5_list(f, path.c_str(), buf.data(), buf.size(), &need) != SRCFD_OK) return;
  std::stringstream ss(buf.data());
  std::string c;
  while (std::getline(ss, c))
    if (!c.empty()) walk(f, (path == "/" ? "/" : path + "/") + c, depth + 1, reads);
}

int main(int argc, char** argv) {
  int ok = 0, bad = 0;
  long reads = 0;
  for (int i = 1; i < argc; ++i) {
    srcfd_h5* f = nullptr;
    if (srcfd_h5_open(argv[i], &f) == SRCFD_OK && f) {
      walk(f, "/", 0, reads);
      srcfd_h5_close(f);
    }
    try {
      srcfd::ModelDesc m;
      srcfd::append_h5_submodel(m, argv[i]);
      m.infer_shapes();
      ++ok;
    } catch (const srcfd::FileError&) { ++bad; } catch (const std::exception&) { ++bad; }
    try {   // and as a whole-model (superres_*.h5) file
      srcfd::ModelDesc w;
      srcfd::append_h5_whole(w, argv[i]);
      w.infer_shapes();
    } catch (const srcfd::FileError&) {} catch (const std::exception&) {}
  }
  std::printf("%d %d %ld\n", ok, bad, reads);
  return 0;
}
