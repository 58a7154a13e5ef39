// conv_registry.h -- table of compiled conv_igemm_f32 instantiations.
#pragma once
#include "conv_igemm.h"
#include <vector>

namespace apamd {
// Tile configurations <WCO, MT, WPX, NT>:
//   A: 128 couts x 4 rows x 32 cols   (2x2 waves, 2x2 MFMA tiles per wave)
//   B:  64 couts x 8 rows x 32 cols   (1x4 waves, 2x2)
//   C:  32 couts x 8 rows x 32 cols   (1x4 waves, 1x2)
#define APAMD_CFG_A 2, 2, 2, 2
#define APAMD_CFG_B 1, 2, 4, 2
#define APAMD_CFG_C 1, 1, 4, 2

void register_s1k0(std::vector<ConvKernelInfo>&);
void register_s1k3(std::vector<ConvKernelInfo>&);
void register_s1k4(std::vector<ConvKernelInfo>&);
void register_s1k7(std::vector<ConvKernelInfo>&);
void register_s2k3(std::vector<ConvKernelInfo>&);
void register_s2k4(std::vector<ConvKernelInfo>&);

#define APAMD_REGISTER_ALL(S, K)                                            \
    v.push_back(make_info<ConvCfg<2, S, K, APAMD_CFG_A>>());                \
    v.push_back(make_info<ConvCfg<4, S, K, APAMD_CFG_A>>());                \
    v.push_back(make_info<ConvCfg<8, S, K, APAMD_CFG_A>>());                \
    v.push_back(make_info<ConvCfg<2, S, K, APAMD_CFG_B>>());                \
    v.push_back(make_info<ConvCfg<4, S, K, APAMD_CFG_B>>());                \
    v.push_back(make_info<ConvCfg<8, S, K, APAMD_CFG_B>>());                \
    v.push_back(make_info<ConvCfg<2, S, K, APAMD_CFG_C>>());                \
    v.push_back(make_info<ConvCfg<4, S, K, APAMD_CFG_C>>());                \
    v.push_back(make_info<ConvCfg<8, S, K, APAMD_CFG_C>>());
}  // namespace apamd
