// Co-residency hazard lab (DESIGN.md section 3.9 "Concurrent streams"): kernels for tools/hazard/run_hazard.py.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -shared -fPIC -o tools/hazard/libhazard.so tools/hazard/hazard_kernels.hip
//
// AGGRESSORS  hz_aggr_launch(variant, ...): a compiler-only LDS-DMA + MFMA loop (no inline assembly: the builtin LDS-DMA) whose
//   register footprint is the knob -- the product's matrix kernels disturb a co-resident kernel of another stream, rocBLAS
//   GEMMs do not (round 3).  Variants differ in ONE thing each:
//     0  baseline: 8 accumulator tiles (128 AGPRs) + whatever arch VGPRs the loop needs, one wave per SIMD (launch_bounds(256,1))
//     1  the same with the arch VGPR count padded to 160 (a clobbered v159): accum_offset 160, 288 registers in all
//     2  padded to 192 arch VGPRs (320 in all)                3  padded to 128 arch (256 in all: two waves per SIMD would fit)
//     4  baseline with amdgpu_waves_per_eu(1, 1) spelled out   5  LDS-DMA only (no MFMA)          6  MFMA only (no LDS-DMA)
//     7  baseline with s_nop 7 after every LDS-DMA piece       8 / 9 / 10  arch VGPRs padded to 136 / 144 / 152 (264 / 272 / 280 in all:
//        round 3's failing reproducer had 136 + 128)
//     11 baseline + a dynamically indexed private array (round 3's failing flags 15)   12 the same with the raw-assembly LDS-DMA
//        (flags 7)   13 = 11 padded to 160 arch VGPRs
// VICTIMS
//   hz_reduce_launch: a ring-reduce-shaped kernel (RCCL's reduceCopy inner loop: few workgroups, each thread streams 16-byte
//     loads from two buffers, adds, stores; 4 loads in flight) -- stands for RCCL's all-reduce kernels beside the backward pass.
//   hz_interp_launch: the address pattern of the warp kernel's failing branch reduced to its core: four float2 gathers at
//     computed offsets (bilinear interpolation of a coarse map), the address registers dead right after the loads.
#include <hip/hip_runtime.h>
#include <cstdint>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int V>
struct AggrAttr {};

template <int V>
__device__ __forceinline__ void aggr_body(const unsigned char* __restrict__ src, float* sink, int iters, int pieces) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const unsigned char* base = src + (size_t)blockIdx.x * pieces * 4096;
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    if (V == 1) asm volatile("" ::: "v159");
    if (V == 2) asm volatile("" ::: "v191");
    if (V == 3) asm volatile("" ::: "v127");
    if (V == 8) asm volatile("" ::: "v135");
    if (V == 9) asm volatile("" ::: "v143");
    if (V == 10) asm volatile("" ::: "v151");
    if (V == 13) asm volatile("" ::: "v159");
    int priv[5];                                 // variants 11-13: round 3's failing reproducer carried a dynamically indexed array
    if (V >= 11)
        for (int i = 0; i < 5; ++i) priv[i] = tid * (i + 1);
    for (int it = 0; it < iters; ++it) {
        if (V != 6) {
            for (int p = 0; p < pieces; ++p) {
                const unsigned voff = (unsigned)((p * 4 + wave) * 64 + lane) * 16u;
                if (V == 12) {          // the product's raw-assembly LDS-DMA (scalar base, M0 written by hand)
                    const unsigned lds_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem + (unsigned)((p * 4 + wave) * 64) * 16u;
                    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds_addr) : "memory", "m0");
                    continue;
                }
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(base + voff),
                                                 (__attribute__((address_space(3))) void*)(smem + (size_t)((p * 4 + wave) * 64) * 16), 16, 0, 0);
                if (V == 7) asm volatile("s_nop 7");
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
        if (V != 5) {
            const uint4* L = reinterpret_cast<const uint4*>(smem);
#pragma unroll
            for (int t = 0; t < 6; ++t) {
                const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 0) * 64 + lane)), a1 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 1) * 64 + lane));
                const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 2) * 64 + lane)), b1 = *reinterpret_cast<const bf16x8*>(L + ((t * 4 + 3) * 64 + lane));
#pragma unroll
                for (int i = 0; i < 8; ++i)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16((i & 1) ? a1 : a0, (i & 2) ? b1 : b0, acc[i], 0, 0, 0);
            }
        }
        if (V >= 11) priv[(it + pieces) % 5] += it;
        if (V != 6) __syncthreads();
    }
    float s = 0.f;
    if (V >= 11) s += (float)priv[pieces % 5];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 123.456f) sink[0] = s;
}

#define AGGR_KERNEL(V, ATTR)                                                                                                   \
    __global__ ATTR void aggr##V(const unsigned char* __restrict__ src, float* sink, int iters, int pieces) {                \
        aggr_body<V>(src, sink, iters, pieces);                                                                               \
    }
AGGR_KERNEL(0, __launch_bounds__(256, 1))
AGGR_KERNEL(1, __launch_bounds__(256, 1))
AGGR_KERNEL(2, __launch_bounds__(256, 1))
AGGR_KERNEL(3, __launch_bounds__(256, 1))
AGGR_KERNEL(4, __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1))))
AGGR_KERNEL(5, __launch_bounds__(256, 1))
AGGR_KERNEL(6, __launch_bounds__(256, 1))
AGGR_KERNEL(7, __launch_bounds__(256, 1))
AGGR_KERNEL(8, __launch_bounds__(256, 1))
AGGR_KERNEL(9, __launch_bounds__(256, 1))
AGGR_KERNEL(10, __launch_bounds__(256, 1))
AGGR_KERNEL(11, __launch_bounds__(256, 1))
AGGR_KERNEL(12, __launch_bounds__(256, 1))
AGGR_KERNEL(13, __launch_bounds__(256, 1))

extern "C" int hz_aggr_launch(int variant, const void* src, void* sink, int iters, void* stream) {
    const int pieces = 24, nblk = 256;
    const size_t lds = (size_t)pieces * 4096;
#define CASE(V) case V: { auto k = aggr##V; hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); \
        hipLaunchKernelGGL(k, dim3(nblk), dim3(256), lds, (hipStream_t)stream, (const unsigned char*)src, (float*)sink, iters, pieces); break; }
    switch (variant) {
        CASE(0) CASE(1) CASE(2) CASE(3) CASE(4) CASE(5) CASE(6) CASE(7) CASE(8) CASE(9) CASE(10) CASE(11) CASE(12) CASE(13)
        default: return -1;
    }
    return (int)hipGetLastError();
}

extern "C" int hz_aggr_registers(int variant, int* arch_plus_acc) {
    hipFuncAttributes a;
    const void* f = nullptr;
    switch (variant) {
        case 0: f = (const void*)aggr0; break; case 1: f = (const void*)aggr1; break; case 2: f = (const void*)aggr2; break;
        case 3: f = (const void*)aggr3; break; case 4: f = (const void*)aggr4; break; case 5: f = (const void*)aggr5; break;
        case 6: f = (const void*)aggr6; break; case 7: f = (const void*)aggr7; break; case 8: f = (const void*)aggr8; break;
        case 9: f = (const void*)aggr9; break; case 10: f = (const void*)aggr10; break; case 11: f = (const void*)aggr11; break;
        case 12: f = (const void*)aggr12; break; case 13: f = (const void*)aggr13; break; default: return -1;
    }
    if (hipFuncGetAttributes(&a, f) != hipSuccess) return -2;
    *arch_plus_acc = a.numRegs;
    return 0;
}

// ---- ring-reduce-shaped victim: out[i] = a[i] + b[i], float4 lanes, `nblk` workgroups of 512 threads striding the buffers
__global__ __launch_bounds__(512) void reduce_copy_kernel(const float4* __restrict__ a, const float4* __restrict__ b, float4* __restrict__ out, long long n4) {
    const long long stride = (long long)gridDim.x * 512 * 4;
    for (long long i = ((long long)blockIdx.x * 512 + threadIdx.x) * 4; i < n4; i += stride) {
        float4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u < n4) { x[u] = a[i + u]; y[u] = b[i + u]; }
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i + u < n4) out[i + u] = make_float4(x[u].x + y[u].x, x[u].y + y[u].y, x[u].z + y[u].z, x[u].w + y[u].w);
    }
}
extern "C" int hz_reduce_launch(const void* a, const void* b, void* out, long long n_floats, int nblk, void* stream) {
    hipLaunchKernelGGL(reduce_copy_kernel, dim3(nblk), dim3(512), 0, (hipStream_t)stream, (const float4*)a, (const float4*)b, (float4*)out, n_floats / 4);
    return (int)hipGetLastError();
}

// ---- interpolating-gather victim: out[n][y][x] = bilinear sample of a coarse (S x S) float2 map at an (H x W) pixel, align_corners
__global__ __launch_bounds__(256) void interp_kernel(const float2* __restrict__ map, float2* __restrict__ out, int S, int H, int W) {
    const int pix = blockIdx.x * 256 + threadIdx.x, n = blockIdx.y;
    if (pix >= H * W) return;
    const int oy = pix / W, ox = pix - oy * W;
    const float sy = (float)oy * (float)(S - 1) / (float)(H - 1), sx = (float)ox * (float)(S - 1) / (float)(W - 1);
    const int y0 = (int)sy, x0 = (int)sx, y1 = y0 + 1 < S ? y0 + 1 : S - 1, x1 = x0 + 1 < S ? x0 + 1 : S - 1;
    const float fy = sy - (float)y0, fx = sx - (float)x0;
    const float2* m = map + (long long)n * S * S;
    const float2 a = m[y0 * S + x0], b = m[y0 * S + x1], c = m[y1 * S + x0], d = m[y1 * S + x1];
    float2 r;
    r.x = (a.x * (1.f - fx) + b.x * fx) * (1.f - fy) + (c.x * (1.f - fx) + d.x * fx) * fy;
    r.y = (a.y * (1.f - fx) + b.y * fx) * (1.f - fy) + (c.y * (1.f - fx) + d.y * fx) * fy;
    out[(long long)n * H * W + pix] = r;
}
extern "C" int hz_interp_launch(const void* map, void* out, int N, int S, int H, int W, void* stream) {
    hipLaunchKernelGGL(interp_kernel, dim3((H * W + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, (const float2*)map, (float2*)out, S, H, W);
    return (int)hipGetLastError();
}
