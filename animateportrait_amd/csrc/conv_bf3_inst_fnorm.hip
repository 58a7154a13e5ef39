// conv_bf16x3 instantiations: 3x3 stride 1 tall tile with InstanceNorm in the epilogue (ap_conv2d_fwd_norm, opt-in) (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
const void* bf3_fnorm_kernel() { return reinterpret_cast<const void*>(&conv_bf16x3<Bf3Cfg<1, 3, 1, 2, 4, 4, 0, 0, 2, 0, 1>>); }
}  // namespace apamd
