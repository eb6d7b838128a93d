// bf16/f16 fused path -- placeholder until the kernels land.
#include "engine.h"
namespace srcfd {
int fused_init(Model&) { return SRCFD_OK; }
void fused_free(Model&) {}
int fused_forward(Model&, const float*, int, const float*, const float*, void*, int, int, unsigned long long*, hipStream_t) {
  set_error("bf16/f16 fused path not built");
  return SRCFD_EINVAL;
}
}  // namespace srcfd
