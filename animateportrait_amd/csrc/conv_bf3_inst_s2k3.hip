// conv_bf16x3 instantiations: 3x3 stride 2, 64 couts x 4 rows (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_register_s2k3(std::vector<Bf3Kernel>& v) { v.push_back(bk2<2, 3, 2, 1, 2, 2>("Bf3Cfg<2, 3, 2, 1, 2, 2>")); }
}  // namespace apamd
