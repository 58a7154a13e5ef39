// conv_igemm_f32 instantiations: stride 1, 4x4 taps (see conv_registry.h)
#include "conv_registry.h"
namespace apamd {
void register_s1k4(std::vector<ConvKernelInfo>& v) { APAMD_REGISTER_ALL(1, 4) }
}  // namespace apamd
