// conv_bf16x3 instantiations: 4 run-time taps, half-height tile (64 couts x 8 rows, 74 KB of LDS): two workgroups per CU for the
// launches whose tiles are short in K and heavy in output (fused phases, space-to-depth layers) (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_s2d3_kernels(const void*& fn, const void*& fn1);
void bf3_register_taps4_half(std::vector<Bf3Kernel>& v) {
    v.push_back(bk2<1, 0, 1, 2, 4, 2, 4>("Bf3Cfg<1, 0, 1, 2, 4, 2, 4>"));
    bf3_s2d3_kernels(v.back().fn_s2d3, v.back().fn1_s2d3);
}
}  // namespace apamd
