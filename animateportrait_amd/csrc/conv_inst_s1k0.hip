// conv_igemm_f32 instantiations: stride 1, run-time taps in a 2x2 window (transposed-conv phases) (see conv_registry.h)
#include "conv_registry.h"
namespace apamd {
void register_s1k0(std::vector<ConvKernelInfo>& v) { APAMD_REGISTER_ALL(1, 0) }
}  // namespace apamd
