// conv_igemm_f32 instantiations: stride 1, tap-window extent 6 (see conv_registry.h)
#include "conv_registry.h"
namespace apamd {
void register_s1e6(std::vector<ConvKernelInfo>& v) { APAMD_REGISTER_ALL(1, 6) }
}  // namespace apamd
