// aggressor.hip -- library-free reproducer of the co-residency hazard (profiles/r04_cohazard.md), part 1 of 3.
// A compiler-only LDS-DMA + MFMA loop for gfx950: 256 workgroups of 4 waves, 96 KB of LDS each (one workgroup per CU, one wave
// per SIMD), every iteration stages 24 x 4 KiB with global_load_lds_dwordx4, multiplies 48 v_mfma_f32_32x32x16_bf16 on it and
// -- the ingredient that separates the failing from the clean variant -- updates a five-element private array at a RUN-TIME
// index (no scratch: the compiler turns it into s_cmp / s_cselect_b64 / v_cndmask on SGPR-pair masks between the LDS-DMA pieces
// and the MFMAs).  PRIV = 0 is the control: the same loop without the array ("synth:0" of the lab, clean in 200 of 200).
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -c aggressor.hip
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int PRIV>
__device__ __forceinline__ void aggressor_body(const unsigned char* __restrict__ src, float* sink, int iters, int pieces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned char* base = src + (size_t)blockIdx.x * pieces * 4096;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    int priv[5];
    if (PRIV)
        for (int i = 0; i < 5; ++i) priv[i] = tid * (i + 1);
    for (int it = 0; it < iters; ++it) {
        for (int p = 0; p < pieces; ++p) {
            const unsigned voff = (unsigned)((p * 4 + wave) * 64 + lane) * 16u;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                             (__attribute__((address_space(3))) void*)(smem + (size_t)((p * 4 + wave) * 64) * 16), 16, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const uint4* L = reinterpret_cast<const uint4*>(smem);
#pragma unroll
        for (int t = 0; t < 6; ++t) {
            const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 0) * 64 + lane)), a1 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 1) * 64 + lane));
            const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 2) * 64 + lane)), b1 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 3) * 64 + lane));
#pragma unroll
            for (int i = 0; i < 8; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
        }
        if (PRIV) priv[(it + pieces) % 5] += it;
        __syncthreads();
    }
    float s = 0.f;
    if (PRIV) s += (float)priv[pieces % 5];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) sink[0] = s;
}

__global__ __launch_bounds__(256, 1) void aggressor_failing(const unsigned char* __restrict__ src, float* sink, int iters, int pieces) {
    aggressor_body<1>(src, sink, iters, pieces);
}
__global__ __launch_bounds__(256, 1) void aggressor_control(const unsigned char* __restrict__ src, float* sink, int iters, int pieces) {
    aggressor_body<0>(src, sink, iters, pieces);
}

// src: >= 256 * 24 * 4096 bytes of anything finite; sink: one float.  which: 1 = failing, 0 = control.
extern "C" hipError_t launch_aggressor(int which, const void* src, void* sink, int iters, hipStream_t stream) {
    const int pieces = 24, nblk = 256;
    const size_t lds = (size_t)pieces * 4096;
    auto k = which ? aggressor_failing : aggressor_control;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, stream, (const unsigned char*)src, (float*)sink, iters, pieces);
    return hipGetLastError();
}
