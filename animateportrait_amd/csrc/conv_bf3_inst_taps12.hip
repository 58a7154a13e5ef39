// conv_bf16x3 instantiations: run-time taps in a 2x2 window, 1 and 2 taps (sub-pixel phases of ConvTranspose2d(s=2)) (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_register_taps12(std::vector<Bf3Kernel>& v) {
    v.push_back(bk2<1, 0, 1, 2, 4, 4, 1>("Bf3Cfg<1, 0, 1, 2, 4, 4, 1>"));
    v.push_back(bk2<1, 0, 1, 2, 4, 4, 2>("Bf3Cfg<1, 0, 1, 2, 4, 4, 2>"));
}
}  // namespace apamd
