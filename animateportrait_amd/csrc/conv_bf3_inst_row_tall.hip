// conv_bf16x3 instantiations: 1x7 over row channels (7x7 stems), 64 couts x 16 rows (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_register_row_tall(std::vector<Bf3Kernel>& v) { v.push_back(bk2<1, 7, 1, 2, 4, 4, 0, 1>("Bf3Cfg<1, 7, 1, 2, 4, 4, 0, 1>")); }
}  // namespace apamd
