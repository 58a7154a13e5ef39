// conv_bf16x3 instantiations: 4x4 stride 1 (PatchGAN): 16 taps -> 32-cout tiles (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_register_k4(std::vector<Bf3Kernel>& v) { v.push_back(bk2<1, 4, 1, 1, 4, 4>("Bf3Cfg<1, 4, 1, 1, 4, 4>")); }
}  // namespace apamd
