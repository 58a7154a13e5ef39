// conv_bf16x3 instantiations: the half-height 4-tap tile with the compile-time tap sets of a space-to-depth 3x3 layer (Bf3Cfg::S2D3) (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_s2d3_kernels(const void*& fn, const void*& fn1) {
    fn = reinterpret_cast<const void*>(&conv_bf16x3<Bf3Cfg<1, 0, 1, 2, 4, 2, 4, 0, 2, 1>>);
    fn1 = reinterpret_cast<const void*>(&conv_bf16x3<Bf3Cfg<1, 0, 1, 2, 4, 2, 4, 0, 1, 1>>);
}
}  // namespace apamd
