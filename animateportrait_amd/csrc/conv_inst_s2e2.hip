// conv_igemm_f32 instantiations: stride 2, tap-window extent 2 (see conv_registry.h)
#include "conv_registry.h"
namespace apamd {
void register_s2e2(std::vector<ConvKernelInfo>& v) { APAMD_REGISTER_ALL(2, 2) }
}  // namespace apamd
