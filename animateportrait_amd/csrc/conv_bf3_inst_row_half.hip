// conv_bf16x3 instantiations: 1x7 over row channels, 32 couts x 8 rows (70 KB of LDS: two workgroups per CU, so that the output
// burst of one tile's epilogue runs under the other workgroup's MFMAs -- the stems are write-bound: 2 chunks of K) (see conv_bf3_registry.h)
#include "conv_bf3_registry.h"
namespace apamd {
void bf3_register_row_half(std::vector<Bf3Kernel>& v) { v.push_back(bk2<1, 7, 1, 1, 4, 2, 0, 1>("Bf3Cfg<1, 7, 1, 1, 4, 2, 0, 1>")); }
}  // namespace apamd
