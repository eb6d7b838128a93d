// Host-side description of a sequential Keras sub-model chain (encoder_10 then
// decoder_400 in the reference: sr-ae-conv.ipynb:c162-169, c277-287) and its
// legacy Keras-H5 (de)serialisation.  No HIP in this file.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "../../include/srcfd.h"

namespace srcfd {

struct Layer {
  int kind = 0;
  int act = SRCFD_ACT_LINEAR;
  int kh = 1, kw = 1, stride = 1, same = 0;
  int cin = 0, cout = 0;
  int reshape[3] = {0, 0, 0};
  std::string name;
  std::vector<float> kernel, bias;
  // shapes, filled by infer_shapes(): (h,w,c); dense/flatten outputs are (1,1,c)
  int in_shape[3] = {0, 0, 0}, out_shape[3] = {0, 0, 0};
  int64_t macs = 0;  // per sample
};

struct SubModel {  // one .h5 file
  std::string name;        // "encoder_10"
  std::string input_name;  // "encoder_10_input"
  int first = 0, count = 0;  // layer range in ModelDesc::layers
};

struct ModelDesc {
  int in_shape[3] = {0, 0, 0};
  std::vector<Layer> layers;
  std::vector<SubModel> subs;

  void infer_shapes();  // throws std::runtime_error on inconsistent graphs
  const int* out_shape() const { return layers.back().out_shape; }
  int64_t macs_per_sample() const;
  // true for the exact decoder_400 tail pattern the fused kernels implement
  bool is_sr_10_400() const;
};

// Appends the layers of one legacy Keras-H5 sub-model file.  Throws
// FileError(SRCFD_ENOENT / SRCFD_EIO) with a message.
struct FileError {
  int code;
  std::string msg;
};
void append_h5_submodel(ModelDesc& m, const std::string& path);
void save_h5_submodel(const ModelDesc& m, int sub_index, const std::string& path);
// Whole-model file of SuperResolutionAE (`superres_model.save(...)`, sr-ae-conv.ipynb:c586): both sub-models in one file.
void append_h5_whole(ModelDesc& m, const std::string& path);
void save_h5_whole(const ModelDesc& m, const std::string& path);

const char* act_name(int act);  // Keras 3 serialised names ("silu", "linear", ...)
int act_from_name(const std::string& s);

}  // namespace srcfd
