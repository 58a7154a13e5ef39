// conv_bf16x3 instantiations: run-time taps in a 2x2 window, 4 taps, 64 couts x 16 rows (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_register_taps4(std::vector<Bf3Kernel>& v) { v.push_back(bk2<1, 0, 1, 2, 4, 4, 4>("Bf3Cfg<1, 0, 1, 2, 4, 4, 4>")); }
}  // namespace apamd
