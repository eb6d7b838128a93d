// h5lite: dependency-free reader/writer for the subset of HDF5 that the SR hot
// path's artefacts use (legacy Keras-H5 weight files and the solver's flat
// float64 field dumps).  Replaces, for this path only, what the reference gets
// from h5py/libhdf5 through `tf.keras.models.load_model`
// (PyCFD_ML_accelerated.py:831-832) and `h5py.File` (PyCFD_ML_accelerated.py:517-544).
//
// Supported on read : superblock v0/v1, v1 object headers (+continuations),
//   old-style groups (v1 B-tree + local heap + SNOD) and compact new-style link
//   messages, contiguous/compact datasets of f32/f64/i32/i64/u8, attributes
//   v1-v3 holding numeric scalars/arrays, fixed strings and variable-length
//   strings (global heap).
// Written           : superblock v0, old-style groups, contiguous datasets,
//   vlen-string and numeric attributes -- the layout h5py emits with
//   libver='earliest', which is what Keras' legacy `.h5` saver produces.
#pragma once
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <vector>

namespace h5lite {

enum DType : int { F32 = 0, F64 = 1, I32 = 2, I64 = 3, U8 = 4, STR = 5, UNKNOWN = -1 };
size_t dtype_size(DType t);

struct Attr {
  DType dtype = UNKNOWN;
  bool scalar = true;
  std::vector<uint64_t> dims;
  std::vector<uint8_t> raw;          // numeric payload, little endian
  std::vector<std::string> strings;  // STR payload
  bool utf8 = false;
};

struct Node {
  bool is_group = true;
  // group
  std::vector<std::pair<std::string, std::shared_ptr<Node>>> children;  // sorted on write
  // dataset
  DType dtype = UNKNOWN;
  std::vector<uint64_t> dims;
  uint64_t data_addr = 0;  // reader: file offset of contiguous data
  uint64_t data_size = 0;
  std::vector<uint8_t> data;  // writer payload, or compact data on read
  std::vector<std::pair<std::string, Attr>> attrs;

  Node* child(const std::string& name);
  const Attr* attr(const std::string& name) const;
};

class File {
 public:
  // Reader.  Throws std::runtime_error with a descriptive message.
  static std::unique_ptr<File> open(const std::string& path);
  // Writer.
  static std::unique_ptr<File> create();

  Node* root() { return root_.get(); }
  Node* find(const std::string& path);        // "/a/b/c" or "a/b/c"; nullptr if absent
  Node* make_group(const std::string& path);  // mkdir -p
  Node* make_dataset(const std::string& path, DType t, const std::vector<uint64_t>& dims,
                     const void* data);
  // Copy out (with conversion between the numeric types when `as` differs).
  void read(Node* ds, void* dst, size_t dst_bytes, DType as);
  void save(const std::string& path);

 private:
  std::shared_ptr<Node> root_ = std::make_shared<Node>();
  std::vector<uint8_t> buf_;  // whole file (reader)
  friend struct Parser;
};

}  // namespace h5lite
