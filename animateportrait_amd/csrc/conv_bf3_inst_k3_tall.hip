// conv_bf16x3 instantiations: 3x3 stride 1, 64 couts x 16 rows (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_register_k3_tall(std::vector<Bf3Kernel>& v) { 
    v.push_back(bk2<1, 3, 1, 2, 4, 4>("Bf3Cfg<1, 3, 1, 2, 4, 4>"));
    v.back().fn1_ob16 = reinterpret_cast<const void*>(&conv_bf16x3<Bf3Cfg<1, 3, 1, 2, 4, 4, 0, 0, 1, 0, 0, 1>>);
}
}  // namespace apamd
