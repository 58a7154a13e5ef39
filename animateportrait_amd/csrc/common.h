// common.h -- error reporting shared by the C-ABI translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include "../../include/animateportrait_amd.h"

// Ablation switches (skip the DMA refill / the barriers / the epilogue of the matrix kernels, APAMD_ABLATE=<bits>)
// exist only in the experiment build (`make ablate` -> libapamd_ablate.so, -DAPAMD_ABLATION); in the product library
// the tests compile to nothing and the environment variable is not read.
#ifdef APAMD_ABLATION
#define AP_ABLATE(p, bits) ((p).ablate & (bits))
#else
#define AP_ABLATE(p, bits) 0
#endif

namespace apamd {
char* last_error_buf();   // thread-local, 512 bytes
int fail(int code, const char* fmt, ...);
inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "%s: %s", what, hipGetErrorString(e));
    return AP_OK;
}
// InstanceNorm statistics: planes with mean^2 > ratio * var (as estimated from the conv epilogue's fp32 sums) are
// recomputed from the data (instnorm.hip: instnorm_finalize_kernel, conv_bf16x3.h: norm_split_kernel)
constexpr float kInstNormRefineRatio = 32.f;

// Workgroup b of a launch runs on XCD b % 8 (observed dispatch order; linear id = x fastest, then y, then z) and every
// XCD has its own L2.  xcd_logical_block gives each XCD a CONTIGUOUS range of the logical block list, so blocks that
// share input lines (neighbouring rows of a gather, tile halos, the channel groups of one image) meet in one L2 instead
// of being fetched from the fabric by all eight.  A bijection on [0, nblk); pure speed, any placement is correct.
__device__ __forceinline__ unsigned xcd_logical_block(unsigned nblk, unsigned b) {
    const unsigned q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// V accumulators of a lane summed over the 64 lanes of its wave, into out[0 .. V).  By halving: at the step of lane bit b
// a lane keeps one half of its values and receives the partner's partial sums of that half, so the V values cost
// V (1 - 1/64) exchanges instead of 6 V butterfly steps -- with 128 accumulators the butterfly (768 dependent
// ds_bpermute, each waited for) WAS the narrow weight-gradient kernel: ~40 us per workgroup against a main loop of a few.
template <int V>
__device__ __forceinline__ void wave_sums_to_lds(const float (&acc)[V], int tid, float* out) {
    constexpr int VP = V <= 64 ? 64 : (V <= 128 ? 128 : 256);          // padded to a power of two >= 64
    static_assert(V <= 256, "at most 256 accumulators per lane");
    const int lane = tid & 63;
    float v[VP];
#pragma unroll
    for (int i = 0; i < VP; ++i) v[i] = i < V ? acc[i < V ? i : 0] : 0.f;
    int cnt = VP;
#pragma unroll
    for (int bit = 32; bit >= 1; bit >>= 1) {
        const int h = cnt >> 1;
        const bool up = (lane & bit) != 0;
#pragma unroll
        for (int j = 0; j < VP / 2; ++j) {
            if (j < h) {
                const float keep = up ? v[h + j] : v[j], give = up ? v[j] : v[h + j];
                v[j] = keep + __shfl_xor(give, bit, 64);
            }
        }
        cnt = h;
    }
    // lane l now holds the wave totals of values (VP / 64) l .. (VP / 64) l + VP / 64 - 1
    constexpr int PER = VP / 64;
#pragma unroll
    for (int t = 0; t < PER; ++t)
        if (PER * lane + t < V) out[PER * lane + t] = v[t];
}
}  // namespace apamd
