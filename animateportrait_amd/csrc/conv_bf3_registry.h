// conv_bf3_registry.h -- table of the compiled conv_bf16x3 / conv_ph4 instantiations.
//
// Every tile family is instantiated in its own translation unit (conv_bf3_inst_*.hip), as the fp32 kernels are
// (conv_registry.h): the instantiations are independent and the build runs them in parallel, and one family can be
// rebuilt without the host file.  conv_host.hip sees only the entries below.
#pragma once
#include <vector>

#include "conv_bf16x3.h"

namespace apamd {

struct Bf3Kernel {
    int S, K, CO_TILE, TH, TMAX, ROW;
    const void* fn;              // split-bf16 arithmetic (AP_PRECISION_BF16X3): head + tail staged, 3 MFMAs per product
    int (*wfloats)(int);
    size_t (*lds_bytes)(int);
    const char* name;
    const void* fn1;             // plain bf16 arithmetic (AP_PRECISION_BF16): the same tile with head parts only
    size_t (*lds_bytes1)(int);
    const void* fn_s2d3 = nullptr;    // the same tile with the compile-time tap sets of a space-to-depth 3x3 layer (Bf3Cfg::S2D3)
    const void* fn1_s2d3 = nullptr;
    const void* fn1_ob16 = nullptr;   // plain-bf16 arithmetic with the OUTPUT stored as bf16 (Bf3Cfg::OB16; the 3x3 stride-1 tiles)
    const void* kernel(int precision) const { return precision == AP_PRECISION_BF16 ? fn1 : fn; }
    const void* kernel_s2d3(int precision) const { return precision == AP_PRECISION_BF16 ? fn1_s2d3 : fn_s2d3; }
    size_t lds(int precision, int ntaps) const { return precision == AP_PRECISION_BF16 ? lds_bytes1(ntaps) : lds_bytes(ntaps); }
};

// both arithmetic modes of one tile
template <int S, int K, int WCO, int MT, int WPX, int NT, int NTAP = 0, int ROW = 0>
static Bf3Kernel bk2(const char* name) {
    using C = Bf3Cfg<S, K, WCO, MT, WPX, NT, NTAP, ROW, 2>;
    using C1 = Bf3Cfg<S, K, WCO, MT, WPX, NT, NTAP, ROW, 1>;
    return Bf3Kernel{C::S, C::K, C::CO_TILE, C::TH, C::TMAX, C::ROW, reinterpret_cast<const void*>(&conv_bf16x3<C>),
                     &C::wfloats, &C::lds_bytes, name, reinterpret_cast<const void*>(&conv_bf16x3<C1>), &C1::lds_bytes};
}

// one function per translation unit; the order of the calls in conv_host.hip is the registry's order
void bf3_register_k3_tall(std::vector<Bf3Kernel>&);      // 3x3 s1: 64 couts x 16 rows (the generator's dominant kernel)
void bf3_register_k3_short(std::vector<Bf3Kernel>&);     // 3x3 s1, small batches: 64 couts x 4 rows
void bf3_register_s2k3(std::vector<Bf3Kernel>&);         // 3x3 s2: 64 couts x 4 rows
void bf3_register_k4(std::vector<Bf3Kernel>&);           // 4x4 s1 (PatchGAN): 16 taps -> 32-cout tiles
void bf3_register_row_tall(std::vector<Bf3Kernel>&);     // 1x7 over row channels (7x7 stems)
void bf3_register_row_half(std::vector<Bf3Kernel>&);     // ... 32 couts x 8 rows, two workgroups per CU
void bf3_register_taps12(std::vector<Bf3Kernel>&);       // run-time taps: 1 and 2 taps (sub-pixel phases)
void bf3_register_taps4(std::vector<Bf3Kernel>&);        // run-time taps: 4 taps, full height
void bf3_register_taps4_half(std::vector<Bf3Kernel>&);   // ... half height (two per CU) + its space-to-depth 3x3 tap sets
const void* bf3_fnorm_kernel();                          // Bf3Cfg<1,3,1,2,4,4,...,FNORM>: convolution + InstanceNorm in one launch
const Bf3Kernel* ph4_kernel(int KK);                     // conv_ph4.h: 3 or 4

}  // namespace apamd
