// conv_ph4 instantiations: the four sub-pixel phases of a stride-2 transposed layer in one tile, 32 couts x 8 rows x 4 phases
#include "conv_bf3_registry.h"
#include "conv_ph4.h"
namespace apamd {
template <int KK>
static const Bf3Kernel* ph4_entry() {
    using C = Ph4Cfg<KK, 2>;
    using C1 = Ph4Cfg<KK, 1>;
    static const Bf3Kernel k{1, 0, C::CO_TILE, C::TH, 4, 0, reinterpret_cast<const void*>(&conv_ph4<C>), &C::wfloats, &C::lds_bytes,
                             KK == 3 ? "Ph4Cfg<3>" : "Ph4Cfg<4>", reinterpret_cast<const void*>(&conv_ph4<C1>), &C1::lds_bytes};
    return &k;
}
const Bf3Kernel* ph4_kernel(int KK) { return KK == 3 ? ph4_entry<3>() : ph4_entry<4>(); }
}  // namespace apamd
