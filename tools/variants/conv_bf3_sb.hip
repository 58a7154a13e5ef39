// Rejected variant of the dominant 3x3 kernel (HISTORY.md section 3.12), built only by `make -C animateportrait_amd/csrc variants`
// into libapamd_variants.so and selected there with APAMD_CONV_SB=1 (tools/sb_check.py).
#include "conv_bf16x3_sb.h"
namespace apamd {
const void* bf3_sb_kernel(size_t* lds_bytes) {
    using CS = Bf3Cfg<1, 3, 1, 2, 4, 4, 0, 0, 2>;
    *lds_bytes = (size_t)(CS::X_SLOTS + CS::w_slots(9)) * 16;
    return reinterpret_cast<const void*>(&conv_bf16x3_sb<CS>);
}
}  // namespace apamd
