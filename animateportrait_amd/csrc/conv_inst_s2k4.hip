// conv_igemm_f32 instantiations: stride 2, 4x4 taps (see conv_registry.h)
#include "conv_registry.h"
namespace apamd {
void register_s2k4(std::vector<ConvKernelInfo>& v) { APAMD_REGISTER_ALL(2, 4) }
}  // namespace apamd
