// Synthetic aggressors for tools/conc_warp.py (AGGR=synth:<flags>): which ingredient of the matrix kernels disturbs a
// kernel of another stream?  flags: 1 = LDS-DMA (raw global_load_lds_dwordx4), 2 = MFMA + ds_read_b128 loop with AGPR
// accumulators, 4 = a dynamically indexed private array (8 more VGPRs; no scratch in practice), 8 = builtin LDS-DMA instead of raw asm
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void glds16(const void* sbase, unsigned voff, unsigned lds_addr) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

template <int F>
__global__ __launch_bounds__(256, 1) void aggr(const unsigned char* __restrict__ src, float* sink, int iters, int pieces, int sel) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
    const unsigned char* base = src + (size_t)blockIdx.x * pieces * 4096;
    constexpr int NA = (F & 16) ? 10 : 8;      // 16: two more accumulator tiles (more than 256 registers in total)
    f32x16 acc[NA];
    for (int i = 0; i < NA; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int priv[5];
    if (F & 4)
        for (int i = 0; i < 5; ++i) priv[i] = tid * (i + 1);
    for (int it = 0; it < iters; ++it) {
        if (F & 1) {
            for (int p = 0; p < pieces; ++p) {
                const unsigned voff = (unsigned)((p * 4 + wave) * 64 + lane) * 16u;
                if (F & 8)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                                     (__attribute__((address_space(3))) void*)(smem + (size_t)((p * 4 + wave) * 64) * 16), 16, 0, 0);
                else glds16(base, voff, lds0 + (unsigned)((p * 4 + wave) * 64) * 16u);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (F & 2) {
            const uint4* L = reinterpret_cast<const uint4*>(smem);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 0) * 64 + lane)), a1 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 1) * 64 + lane));
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 2) * 64 + lane)), b1 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 3) * 64 + lane));
#pragma unroll
                for (int i = 0; i < NA; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
            }
        }
        if (F & 4) priv[(it + sel) % 5] += it;      // run-time index: the array lives in scratch memory
        if (F & 1) __syncthreads();
    }
    float s = 0.f;
    for (int i = 0; i < NA; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (F & 4) s += (float)priv[sel % 5];
    if (s == 123.456f) sink[0] = s;
}

extern "C" int aggr_launch(int flags, const void* src, void* sink, int iters, void* stream) {
    const int pieces = 24, nblk = 256;
    const size_t lds = (size_t)pieces * 4096;
#define CASE(F) case F: { auto k = aggr<F>; hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const unsigned char*)src, (float*)sink, iters, pieces, 1); break; }
    switch (flags) {
        CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(9) CASE(11) CASE(15) CASE(18) CASE(19) CASE(22)
        default: return -1;
    }
    return (int)hipGetLastError();
}
