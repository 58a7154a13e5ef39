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
}  // namespace apamd
