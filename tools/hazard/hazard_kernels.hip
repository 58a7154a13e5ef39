// Co-residency hazard lab (DESIGN.md section 3.9 "Concurrent streams"; VERDICT r3 item 5): kernels for tools/hazard/run_hazard.py.
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

// ---- poison (victim-side experiment, VERDICT r4 item 6a): one workgroup per CU (160 KB of LDS), one wave per SIMD with all 256
// arch VGPRs + 256 AGPRs of every lane and every LDS dword set to a NaN pattern.  If a kernel launched alone right behind this one
// computes wrong values, it reads state it did not initialise.
__global__ __launch_bounds__(256, 1) void poison_kernel(unsigned* sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < 40960; i += 256) lds[i] = 0x7fc12345u;
    __syncthreads();
    asm volatile("v_mov_b32 v1, 0x7fc12345\n\tv_mov_b32 v2, 0x7fc12345\n\tv_mov_b32 v3, 0x7fc12345\n\tv_mov_b32 v4, 0x7fc12345\n\tv_mov_b32 v5, 0x7fc12345\n\tv_mov_b32 v6, 0x7fc12345\n\tv_mov_b32 v7, 0x7fc12345\n\tv_mov_b32 v8, 0x7fc12345\n\tv_mov_b32 v9, 0x7fc12345\n\tv_mov_b32 v10, 0x7fc12345\n\tv_mov_b32 v11, 0x7fc12345\n\tv_mov_b32 v12, 0x7fc12345\n\tv_mov_b32 v13, 0x7fc12345\n\tv_mov_b32 v14, 0x7fc12345\n\tv_mov_b32 v15, 0x7fc12345\n\tv_mov_b32 v16, 0x7fc12345\n\tv_mov_b32 v17, 0x7fc12345\n\tv_mov_b32 v18, 0x7fc12345\n\tv_mov_b32 v19, 0x7fc12345\n\tv_mov_b32 v20, 0x7fc12345\n\tv_mov_b32 v21, 0x7fc12345\n\tv_mov_b32 v22, 0x7fc12345\n\tv_mov_b32 v23, 0x7fc12345\n\tv_mov_b32 v24, 0x7fc12345\n\tv_mov_b32 v25, 0x7fc12345\n\tv_mov_b32 v26, 0x7fc12345\n\tv_mov_b32 v27, 0x7fc12345\n\tv_mov_b32 v28, 0x7fc12345\n\tv_mov_b32 v29, 0x7fc12345\n\tv_mov_b32 v30, 0x7fc12345\n\tv_mov_b32 v31, 0x7fc12345\n\tv_mov_b32 v32, 0x7fc12345\n\tv_mov_b32 v33, 0x7fc12345\n\tv_mov_b32 v34, 0x7fc12345\n\tv_mov_b32 v35, 0x7fc12345\n\tv_mov_b32 v36, 0x7fc12345\n\tv_mov_b32 v37, 0x7fc12345\n\tv_mov_b32 v38, 0x7fc12345\n\tv_mov_b32 v39, 0x7fc12345\n\tv_mov_b32 v40, 0x7fc12345\n\tv_mov_b32 v41, 0x7fc12345\n\tv_mov_b32 v42, 0x7fc12345\n\tv_mov_b32 v43, 0x7fc12345\n\tv_mov_b32 v44, 0x7fc12345\n\tv_mov_b32 v45, 0x7fc12345\n\tv_mov_b32 v46, 0x7fc12345\n\tv_mov_b32 v47, 0x7fc12345\n\tv_mov_b32 v48, 0x7fc12345\n\tv_mov_b32 v49, 0x7fc12345\n\tv_mov_b32 v50, 0x7fc12345\n\tv_mov_b32 v51, 0x7fc12345\n\tv_mov_b32 v52, 0x7fc12345\n\tv_mov_b32 v53, 0x7fc12345\n\tv_mov_b32 v54, 0x7fc12345\n\tv_mov_b32 v55, 0x7fc12345\n\tv_mov_b32 v56, 0x7fc12345\n\tv_mov_b32 v57, 0x7fc12345\n\tv_mov_b32 v58, 0x7fc12345\n\tv_mov_b32 v59, 0x7fc12345\n\tv_mov_b32 v60, 0x7fc12345\n\tv_mov_b32 v61, 0x7fc12345\n\tv_mov_b32 v62, 0x7fc12345\n\tv_mov_b32 v63, 0x7fc12345\n\tv_mov_b32 v64, 0x7fc12345\n\tv_mov_b32 v65, 0x7fc12345\n\tv_mov_b32 v66, 0x7fc12345\n\tv_mov_b32 v67, 0x7fc12345\n\tv_mov_b32 v68, 0x7fc12345\n\tv_mov_b32 v69, 0x7fc12345\n\tv_mov_b32 v70, 0x7fc12345\n\tv_mov_b32 v71, 0x7fc12345\n\tv_mov_b32 v72, 0x7fc12345\n\tv_mov_b32 v73, 0x7fc12345\n\tv_mov_b32 v74, 0x7fc12345\n\tv_mov_b32 v75, 0x7fc12345\n\tv_mov_b32 v76, 0x7fc12345\n\tv_mov_b32 v77, 0x7fc12345\n\tv_mov_b32 v78, 0x7fc12345\n\tv_mov_b32 v79, 0x7fc12345\n\tv_mov_b32 v80, 0x7fc12345\n\tv_mov_b32 v81, 0x7fc12345\n\tv_mov_b32 v82, 0x7fc12345\n\tv_mov_b32 v83, 0x7fc12345\n\tv_mov_b32 v84, 0x7fc12345\n\tv_mov_b32 v85, 0x7fc12345\n\tv_mov_b32 v86, 0x7fc12345\n\tv_mov_b32 v87, 0x7fc12345\n\tv_mov_b32 v88, 0x7fc12345\n\tv_mov_b32 v89, 0x7fc12345\n\tv_mov_b32 v90, 0x7fc12345\n\tv_mov_b32 v91, 0x7fc12345\n\tv_mov_b32 v92, 0x7fc12345\n\tv_mov_b32 v93, 0x7fc12345\n\tv_mov_b32 v94, 0x7fc12345\n\tv_mov_b32 v95, 0x7fc12345\n\tv_mov_b32 v96, 0x7fc12345\n\tv_mov_b32 v97, 0x7fc12345\n\tv_mov_b32 v98, 0x7fc12345\n\tv_mov_b32 v99, 0x7fc12345\n\tv_mov_b32 v100, 0x7fc12345\n\tv_mov_b32 v101, 0x7fc12345\n\tv_mov_b32 v102, 0x7fc12345\n\tv_mov_b32 v103, 0x7fc12345\n\tv_mov_b32 v104, 0x7fc12345\n\tv_mov_b32 v105, 0x7fc12345\n\tv_mov_b32 v106, 0x7fc12345\n\tv_mov_b32 v107, 0x7fc12345\n\tv_mov_b32 v108, 0x7fc12345\n\tv_mov_b32 v109, 0x7fc12345\n\tv_mov_b32 v110, 0x7fc12345\n\tv_mov_b32 v111, 0x7fc12345\n\tv_mov_b32 v112, 0x7fc12345\n\tv_mov_b32 v113, 0x7fc12345\n\tv_mov_b32 v114, 0x7fc12345\n\tv_mov_b32 v115, 0x7fc12345\n\tv_mov_b32 v116, 0x7fc12345\n\tv_mov_b32 v117, 0x7fc12345\n\tv_mov_b32 v118, 0x7fc12345\n\tv_mov_b32 v119, 0x7fc12345\n\tv_mov_b32 v120, 0x7fc12345\n\tv_mov_b32 v121, 0x7fc12345\n\tv_mov_b32 v122, 0x7fc12345\n\tv_mov_b32 v123, 0x7fc12345\n\tv_mov_b32 v124, 0x7fc12345\n\tv_mov_b32 v125, 0x7fc12345\n\tv_mov_b32 v126, 0x7fc12345\n\tv_mov_b32 v127, 0x7fc12345\n\tv_mov_b32 v128, 0x7fc12345\n\tv_mov_b32 v129, 0x7fc12345\n\tv_mov_b32 v130, 0x7fc12345\n\tv_mov_b32 v131, 0x7fc12345\n\tv_mov_b32 v132, 0x7fc12345\n\tv_mov_b32 v133, 0x7fc12345\n\tv_mov_b32 v134, 0x7fc12345\n\tv_mov_b32 v135, 0x7fc12345\n\tv_mov_b32 v136, 0x7fc12345\n\tv_mov_b32 v137, 0x7fc12345\n\tv_mov_b32 v138, 0x7fc12345\n\tv_mov_b32 v139, 0x7fc12345\n\tv_mov_b32 v140, 0x7fc12345\n\tv_mov_b32 v141, 0x7fc12345\n\tv_mov_b32 v142, 0x7fc12345\n\tv_mov_b32 v143, 0x7fc12345\n\tv_mov_b32 v144, 0x7fc12345\n\tv_mov_b32 v145, 0x7fc12345\n\tv_mov_b32 v146, 0x7fc12345\n\tv_mov_b32 v147, 0x7fc12345\n\tv_mov_b32 v148, 0x7fc12345\n\tv_mov_b32 v149, 0x7fc12345\n\tv_mov_b32 v150, 0x7fc12345\n\tv_mov_b32 v151, 0x7fc12345\n\tv_mov_b32 v152, 0x7fc12345\n\tv_mov_b32 v153, 0x7fc12345\n\tv_mov_b32 v154, 0x7fc12345\n\tv_mov_b32 v155, 0x7fc12345\n\tv_mov_b32 v156, 0x7fc12345\n\tv_mov_b32 v157, 0x7fc12345\n\tv_mov_b32 v158, 0x7fc12345\n\tv_mov_b32 v159, 0x7fc12345\n\tv_mov_b32 v160, 0x7fc12345\n\tv_mov_b32 v161, 0x7fc12345\n\tv_mov_b32 v162, 0x7fc12345\n\tv_mov_b32 v163, 0x7fc12345\n\tv_mov_b32 v164, 0x7fc12345\n\tv_mov_b32 v165, 0x7fc12345\n\tv_mov_b32 v166, 0x7fc12345\n\tv_mov_b32 v167, 0x7fc12345\n\tv_mov_b32 v168, 0x7fc12345\n\tv_mov_b32 v169, 0x7fc12345\n\tv_mov_b32 v170, 0x7fc12345\n\tv_mov_b32 v171, 0x7fc12345\n\tv_mov_b32 v172, 0x7fc12345\n\tv_mov_b32 v173, 0x7fc12345\n\tv_mov_b32 v174, 0x7fc12345\n\tv_mov_b32 v175, 0x7fc12345\n\tv_mov_b32 v176, 0x7fc12345\n\tv_mov_b32 v177, 0x7fc12345\n\tv_mov_b32 v178, 0x7fc12345\n\tv_mov_b32 v179, 0x7fc12345\n\tv_mov_b32 v180, 0x7fc12345\n\tv_mov_b32 v181, 0x7fc12345\n\tv_mov_b32 v182, 0x7fc12345\n\tv_mov_b32 v183, 0x7fc12345\n\tv_mov_b32 v184, 0x7fc12345\n\tv_mov_b32 v185, 0x7fc12345\n\tv_mov_b32 v186, 0x7fc12345\n\tv_mov_b32 v187, 0x7fc12345\n\tv_mov_b32 v188, 0x7fc12345\n\tv_mov_b32 v189, 0x7fc12345\n\tv_mov_b32 v190, 0x7fc12345\n\tv_mov_b32 v191, 0x7fc12345\n\tv_mov_b32 v192, 0x7fc12345\n\tv_mov_b32 v193, 0x7fc12345\n\tv_mov_b32 v194, 0x7fc12345\n\tv_mov_b32 v195, 0x7fc12345\n\tv_mov_b32 v196, 0x7fc12345\n\tv_mov_b32 v197, 0x7fc12345\n\tv_mov_b32 v198, 0x7fc12345\n\tv_mov_b32 v199, 0x7fc12345\n\tv_mov_b32 v200, 0x7fc12345\n\tv_mov_b32 v201, 0x7fc12345\n\tv_mov_b32 v202, 0x7fc12345\n\tv_mov_b32 v203, 0x7fc12345\n\tv_mov_b32 v204, 0x7fc12345\n\tv_mov_b32 v205, 0x7fc12345\n\tv_mov_b32 v206, 0x7fc12345\n\tv_mov_b32 v207, 0x7fc12345\n\tv_mov_b32 v208, 0x7fc12345\n\tv_mov_b32 v209, 0x7fc12345\n\tv_mov_b32 v210, 0x7fc12345\n\tv_mov_b32 v211, 0x7fc12345\n\tv_mov_b32 v212, 0x7fc12345\n\tv_mov_b32 v213, 0x7fc12345\n\tv_mov_b32 v214, 0x7fc12345\n\tv_mov_b32 v215, 0x7fc12345\n\tv_mov_b32 v216, 0x7fc12345\n\tv_mov_b32 v217, 0x7fc12345\n\tv_mov_b32 v218, 0x7fc12345\n\tv_mov_b32 v219, 0x7fc12345\n\tv_mov_b32 v220, 0x7fc12345\n\tv_mov_b32 v221, 0x7fc12345\n\tv_mov_b32 v222, 0x7fc12345\n\tv_mov_b32 v223, 0x7fc12345\n\tv_mov_b32 v224, 0x7fc12345\n\tv_mov_b32 v225, 0x7fc12345\n\tv_mov_b32 v226, 0x7fc12345\n\tv_mov_b32 v227, 0x7fc12345\n\tv_mov_b32 v228, 0x7fc12345\n\tv_mov_b32 v229, 0x7fc12345\n\tv_mov_b32 v230, 0x7fc12345\n\tv_mov_b32 v231, 0x7fc12345\n\tv_mov_b32 v232, 0x7fc12345\n\tv_mov_b32 v233, 0x7fc12345\n\tv_mov_b32 v234, 0x7fc12345\n\tv_mov_b32 v235, 0x7fc12345\n\tv_mov_b32 v236, 0x7fc12345\n\tv_mov_b32 v237, 0x7fc12345\n\tv_mov_b32 v238, 0x7fc12345\n\tv_mov_b32 v239, 0x7fc12345\n\tv_mov_b32 v240, 0x7fc12345\n\tv_mov_b32 v241, 0x7fc12345\n\tv_mov_b32 v242, 0x7fc12345\n\tv_mov_b32 v243, 0x7fc12345\n\tv_mov_b32 v244, 0x7fc12345\n\tv_mov_b32 v245, 0x7fc12345\n\tv_mov_b32 v246, 0x7fc12345\n\tv_mov_b32 v247, 0x7fc12345\n\tv_mov_b32 v248, 0x7fc12345\n\tv_mov_b32 v249, 0x7fc12345\n\tv_mov_b32 v250, 0x7fc12345\n\tv_mov_b32 v251, 0x7fc12345\n\tv_mov_b32 v252, 0x7fc12345\n\tv_mov_b32 v253, 0x7fc12345\n\tv_mov_b32 v254, 0x7fc12345\n\tv_mov_b32 v255, 0x7fc12345\n\tv_accvgpr_write_b32 a0, v1\n\tv_accvgpr_write_b32 a1, v1\n\tv_accvgpr_write_b32 a2, v1\n\tv_accvgpr_write_b32 a3, v1\n\tv_accvgpr_write_b32 a4, v1\n\tv_accvgpr_write_b32 a5, v1\n\tv_accvgpr_write_b32 a6, v1\n\tv_accvgpr_write_b32 a7, v1\n\tv_accvgpr_write_b32 a8, v1\n\tv_accvgpr_write_b32 a9, v1\n\tv_accvgpr_write_b32 a10, v1\n\tv_accvgpr_write_b32 a11, v1\n\tv_accvgpr_write_b32 a12, v1\n\tv_accvgpr_write_b32 a13, v1\n\tv_accvgpr_write_b32 a14, v1\n\tv_accvgpr_write_b32 a15, v1\n\tv_accvgpr_write_b32 a16, v1\n\tv_accvgpr_write_b32 a17, v1\n\tv_accvgpr_write_b32 a18, v1\n\tv_accvgpr_write_b32 a19, v1\n\tv_accvgpr_write_b32 a20, v1\n\tv_accvgpr_write_b32 a21, v1\n\tv_accvgpr_write_b32 a22, v1\n\tv_accvgpr_write_b32 a23, v1\n\tv_accvgpr_write_b32 a24, v1\n\tv_accvgpr_write_b32 a25, v1\n\tv_accvgpr_write_b32 a26, v1\n\tv_accvgpr_write_b32 a27, v1\n\tv_accvgpr_write_b32 a28, v1\n\tv_accvgpr_write_b32 a29, v1\n\tv_accvgpr_write_b32 a30, v1\n\tv_accvgpr_write_b32 a31, v1\n\tv_accvgpr_write_b32 a32, v1\n\tv_accvgpr_write_b32 a33, v1\n\tv_accvgpr_write_b32 a34, v1\n\tv_accvgpr_write_b32 a35, v1\n\tv_accvgpr_write_b32 a36, v1\n\tv_accvgpr_write_b32 a37, v1\n\tv_accvgpr_write_b32 a38, v1\n\tv_accvgpr_write_b32 a39, v1\n\tv_accvgpr_write_b32 a40, v1\n\tv_accvgpr_write_b32 a41, v1\n\tv_accvgpr_write_b32 a42, v1\n\tv_accvgpr_write_b32 a43, v1\n\tv_accvgpr_write_b32 a44, v1\n\tv_accvgpr_write_b32 a45, v1\n\tv_accvgpr_write_b32 a46, v1\n\tv_accvgpr_write_b32 a47, v1\n\tv_accvgpr_write_b32 a48, v1\n\tv_accvgpr_write_b32 a49, v1\n\tv_accvgpr_write_b32 a50, v1\n\tv_accvgpr_write_b32 a51, v1\n\tv_accvgpr_write_b32 a52, v1\n\tv_accvgpr_write_b32 a53, v1\n\tv_accvgpr_write_b32 a54, v1\n\tv_accvgpr_write_b32 a55, v1\n\tv_accvgpr_write_b32 a56, v1\n\tv_accvgpr_write_b32 a57, v1\n\tv_accvgpr_write_b32 a58, v1\n\tv_accvgpr_write_b32 a59, v1\n\tv_accvgpr_write_b32 a60, v1\n\tv_accvgpr_write_b32 a61, v1\n\tv_accvgpr_write_b32 a62, v1\n\tv_accvgpr_write_b32 a63, v1\n\tv_accvgpr_write_b32 a64, v1\n\tv_accvgpr_write_b32 a65, v1\n\tv_accvgpr_write_b32 a66, v1\n\tv_accvgpr_write_b32 a67, v1\n\tv_accvgpr_write_b32 a68, v1\n\tv_accvgpr_write_b32 a69, v1\n\tv_accvgpr_write_b32 a70, v1\n\tv_accvgpr_write_b32 a71, v1\n\tv_accvgpr_write_b32 a72, v1\n\tv_accvgpr_write_b32 a73, v1\n\tv_accvgpr_write_b32 a74, v1\n\tv_accvgpr_write_b32 a75, v1\n\tv_accvgpr_write_b32 a76, v1\n\tv_accvgpr_write_b32 a77, v1\n\tv_accvgpr_write_b32 a78, v1\n\tv_accvgpr_write_b32 a79, v1\n\tv_accvgpr_write_b32 a80, v1\n\tv_accvgpr_write_b32 a81, v1\n\tv_accvgpr_write_b32 a82, v1\n\tv_accvgpr_write_b32 a83, v1\n\tv_accvgpr_write_b32 a84, v1\n\tv_accvgpr_write_b32 a85, v1\n\tv_accvgpr_write_b32 a86, v1\n\tv_accvgpr_write_b32 a87, v1\n\tv_accvgpr_write_b32 a88, v1\n\tv_accvgpr_write_b32 a89, v1\n\tv_accvgpr_write_b32 a90, v1\n\tv_accvgpr_write_b32 a91, v1\n\tv_accvgpr_write_b32 a92, v1\n\tv_accvgpr_write_b32 a93, v1\n\tv_accvgpr_write_b32 a94, v1\n\tv_accvgpr_write_b32 a95, v1\n\tv_accvgpr_write_b32 a96, v1\n\tv_accvgpr_write_b32 a97, v1\n\tv_accvgpr_write_b32 a98, v1\n\tv_accvgpr_write_b32 a99, v1\n\tv_accvgpr_write_b32 a100, v1\n\tv_accvgpr_write_b32 a101, v1\n\tv_accvgpr_write_b32 a102, v1\n\tv_accvgpr_write_b32 a103, v1\n\tv_accvgpr_write_b32 a104, v1\n\tv_accvgpr_write_b32 a105, v1\n\tv_accvgpr_write_b32 a106, v1\n\tv_accvgpr_write_b32 a107, v1\n\tv_accvgpr_write_b32 a108, v1\n\tv_accvgpr_write_b32 a109, v1\n\tv_accvgpr_write_b32 a110, v1\n\tv_accvgpr_write_b32 a111, v1\n\tv_accvgpr_write_b32 a112, v1\n\tv_accvgpr_write_b32 a113, v1\n\tv_accvgpr_write_b32 a114, v1\n\tv_accvgpr_write_b32 a115, v1\n\tv_accvgpr_write_b32 a116, v1\n\tv_accvgpr_write_b32 a117, v1\n\tv_accvgpr_write_b32 a118, v1\n\tv_accvgpr_write_b32 a119, v1\n\tv_accvgpr_write_b32 a120, v1\n\tv_accvgpr_write_b32 a121, v1\n\tv_accvgpr_write_b32 a122, v1\n\tv_accvgpr_write_b32 a123, v1\n\tv_accvgpr_write_b32 a124, v1\n\tv_accvgpr_write_b32 a125, v1\n\tv_accvgpr_write_b32 a126, v1\n\tv_accvgpr_write_b32 a127, v1\n\tv_accvgpr_write_b32 a128, v1\n\tv_accvgpr_write_b32 a129, v1\n\tv_accvgpr_write_b32 a130, v1\n\tv_accvgpr_write_b32 a131, v1\n\tv_accvgpr_write_b32 a132, v1\n\tv_accvgpr_write_b32 a133, v1\n\tv_accvgpr_write_b32 a134, v1\n\tv_accvgpr_write_b32 a135, v1\n\tv_accvgpr_write_b32 a136, v1\n\tv_accvgpr_write_b32 a137, v1\n\tv_accvgpr_write_b32 a138, v1\n\tv_accvgpr_write_b32 a139, v1\n\tv_accvgpr_write_b32 a140, v1\n\tv_accvgpr_write_b32 a141, v1\n\tv_accvgpr_write_b32 a142, v1\n\tv_accvgpr_write_b32 a143, v1\n\tv_accvgpr_write_b32 a144, v1\n\tv_accvgpr_write_b32 a145, v1\n\tv_accvgpr_write_b32 a146, v1\n\tv_accvgpr_write_b32 a147, v1\n\tv_accvgpr_write_b32 a148, v1\n\tv_accvgpr_write_b32 a149, v1\n\tv_accvgpr_write_b32 a150, v1\n\tv_accvgpr_write_b32 a151, v1\n\tv_accvgpr_write_b32 a152, v1\n\tv_accvgpr_write_b32 a153, v1\n\tv_accvgpr_write_b32 a154, v1\n\tv_accvgpr_write_b32 a155, v1\n\tv_accvgpr_write_b32 a156, v1\n\tv_accvgpr_write_b32 a157, v1\n\tv_accvgpr_write_b32 a158, v1\n\tv_accvgpr_write_b32 a159, v1\n\tv_accvgpr_write_b32 a160, v1\n\tv_accvgpr_write_b32 a161, v1\n\tv_accvgpr_write_b32 a162, v1\n\tv_accvgpr_write_b32 a163, v1\n\tv_accvgpr_write_b32 a164, v1\n\tv_accvgpr_write_b32 a165, v1\n\tv_accvgpr_write_b32 a166, v1\n\tv_accvgpr_write_b32 a167, v1\n\tv_accvgpr_write_b32 a168, v1\n\tv_accvgpr_write_b32 a169, v1\n\tv_accvgpr_write_b32 a170, v1\n\tv_accvgpr_write_b32 a171, v1\n\tv_accvgpr_write_b32 a172, v1\n\tv_accvgpr_write_b32 a173, v1\n\tv_accvgpr_write_b32 a174, v1\n\tv_accvgpr_write_b32 a175, v1\n\tv_accvgpr_write_b32 a176, v1\n\tv_accvgpr_write_b32 a177, v1\n\tv_accvgpr_write_b32 a178, v1\n\tv_accvgpr_write_b32 a179, v1\n\tv_accvgpr_write_b32 a180, v1\n\tv_accvgpr_write_b32 a181, v1\n\tv_accvgpr_write_b32 a182, v1\n\tv_accvgpr_write_b32 a183, v1\n\tv_accvgpr_write_b32 a184, v1\n\tv_accvgpr_write_b32 a185, v1\n\tv_accvgpr_write_b32 a186, v1\n\tv_accvgpr_write_b32 a187, v1\n\tv_accvgpr_write_b32 a188, v1\n\tv_accvgpr_write_b32 a189, v1\n\tv_accvgpr_write_b32 a190, v1\n\tv_accvgpr_write_b32 a191, v1\n\tv_accvgpr_write_b32 a192, v1\n\tv_accvgpr_write_b32 a193, v1\n\tv_accvgpr_write_b32 a194, v1\n\tv_accvgpr_write_b32 a195, v1\n\tv_accvgpr_write_b32 a196, v1\n\tv_accvgpr_write_b32 a197, v1\n\tv_accvgpr_write_b32 a198, v1\n\tv_accvgpr_write_b32 a199, v1\n\tv_accvgpr_write_b32 a200, v1\n\tv_accvgpr_write_b32 a201, v1\n\tv_accvgpr_write_b32 a202, v1\n\tv_accvgpr_write_b32 a203, v1\n\tv_accvgpr_write_b32 a204, v1\n\tv_accvgpr_write_b32 a205, v1\n\tv_accvgpr_write_b32 a206, v1\n\tv_accvgpr_write_b32 a207, v1\n\tv_accvgpr_write_b32 a208, v1\n\tv_accvgpr_write_b32 a209, v1\n\tv_accvgpr_write_b32 a210, v1\n\tv_accvgpr_write_b32 a211, v1\n\tv_accvgpr_write_b32 a212, v1\n\tv_accvgpr_write_b32 a213, v1\n\tv_accvgpr_write_b32 a214, v1\n\tv_accvgpr_write_b32 a215, v1\n\tv_accvgpr_write_b32 a216, v1\n\tv_accvgpr_write_b32 a217, v1\n\tv_accvgpr_write_b32 a218, v1\n\tv_accvgpr_write_b32 a219, v1\n\tv_accvgpr_write_b32 a220, v1\n\tv_accvgpr_write_b32 a221, v1\n\tv_accvgpr_write_b32 a222, v1\n\tv_accvgpr_write_b32 a223, v1\n\tv_accvgpr_write_b32 a224, v1\n\tv_accvgpr_write_b32 a225, v1\n\tv_accvgpr_write_b32 a226, v1\n\tv_accvgpr_write_b32 a227, v1\n\tv_accvgpr_write_b32 a228, v1\n\tv_accvgpr_write_b32 a229, v1\n\tv_accvgpr_write_b32 a230, v1\n\tv_accvgpr_write_b32 a231, v1\n\tv_accvgpr_write_b32 a232, v1\n\tv_accvgpr_write_b32 a233, v1\n\tv_accvgpr_write_b32 a234, v1\n\tv_accvgpr_write_b32 a235, v1\n\tv_accvgpr_write_b32 a236, v1\n\tv_accvgpr_write_b32 a237, v1\n\tv_accvgpr_write_b32 a238, v1\n\tv_accvgpr_write_b32 a239, v1\n\tv_accvgpr_write_b32 a240, v1\n\tv_accvgpr_write_b32 a241, v1\n\tv_accvgpr_write_b32 a242, v1\n\tv_accvgpr_write_b32 a243, v1\n\tv_accvgpr_write_b32 a244, v1\n\tv_accvgpr_write_b32 a245, v1\n\tv_accvgpr_write_b32 a246, v1\n\tv_accvgpr_write_b32 a247, v1\n\tv_accvgpr_write_b32 a248, v1\n\tv_accvgpr_write_b32 a249, v1\n\tv_accvgpr_write_b32 a250, v1\n\tv_accvgpr_write_b32 a251, v1\n\tv_accvgpr_write_b32 a252, v1\n\tv_accvgpr_write_b32 a253, v1\n\tv_accvgpr_write_b32 a254, v1\n\tv_accvgpr_write_b32 a255, v1" ::: "v1", "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "v12", "v13", "v14", "v15", "v16", "v17", "v18", "v19", "v20", "v21", "v22", "v23", "v24", "v25", "v26", "v27", "v28", "v29", "v30", "v31", "v32", "v33", "v34", "v35", "v36", "v37", "v38", "v39", "v40", "v41", "v42", "v43", "v44", "v45", "v46", "v47", "v48", "v49", "v50", "v51", "v52", "v53", "v54", "v55", "v56", "v57", "v58", "v59", "v60", "v61", "v62", "v63", "v64", "v65", "v66", "v67", "v68", "v69", "v70", "v71", "v72", "v73", "v74", "v75", "v76", "v77", "v78", "v79", "v80", "v81", "v82", "v83", "v84", "v85", "v86", "v87", "v88", "v89", "v90", "v91", "v92", "v93", "v94", "v95", "v96", "v97", "v98", "v99", "v100", "v101", "v102", "v103", "v104", "v105", "v106", "v107", "v108", "v109", "v110", "v111", "v112", "v113", "v114", "v115", "v116", "v117", "v118", "v119", "v120", "v121", "v122", "v123", "v124", "v125", "v126", "v127", "v128", "v129", "v130", "v131", "v132", "v133", "v134", "v135", "v136", "v137", "v138", "v139", "v140", "v141", "v142", "v143", "v144", "v145", "v146", "v147", "v148", "v149", "v150", "v151", "v152", "v153", "v154", "v155", "v156", "v157", "v158", "v159", "v160", "v161", "v162", "v163", "v164", "v165", "v166", "v167", "v168", "v169", "v170", "v171", "v172", "v173", "v174", "v175", "v176", "v177", "v178", "v179", "v180", "v181", "v182", "v183", "v184", "v185", "v186", "v187", "v188", "v189", "v190", "v191", "v192", "v193", "v194", "v195", "v196", "v197", "v198", "v199", "v200", "v201", "v202", "v203", "v204", "v205", "v206", "v207", "v208", "v209", "v210", "v211", "v212", "v213", "v214", "v215", "v216", "v217", "v218", "v219", "v220", "v221", "v222", "v223", "v224", "v225", "v226", "v227", "v228", "v229", "v230", "v231", "v232", "v233", "v234", "v235", "v236", "v237", "v238", "v239", "v240", "v241", "v242", "v243", "v244", "v245", "v246", "v247", "v248", "v249", "v250", "v251", "v252", "v253", "v254", "v255", "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63", "a64", "a65", "a66", "a67", "a68", "a69", "a70", "a71", "a72", "a73", "a74", "a75", "a76", "a77", "a78", "a79", "a80", "a81", "a82", "a83", "a84", "a85", "a86", "a87", "a88", "a89", "a90", "a91", "a92", "a93", "a94", "a95", "a96", "a97", "a98", "a99", "a100", "a101", "a102", "a103", "a104", "a105", "a106", "a107", "a108", "a109", "a110", "a111", "a112", "a113", "a114", "a115", "a116", "a117", "a118", "a119", "a120", "a121", "a122", "a123", "a124", "a125", "a126", "a127", "a128", "a129", "a130", "a131", "a132", "a133", "a134", "a135", "a136", "a137", "a138", "a139", "a140", "a141", "a142", "a143", "a144", "a145", "a146", "a147", "a148", "a149", "a150", "a151", "a152", "a153", "a154", "a155", "a156", "a157", "a158", "a159", "a160", "a161", "a162", "a163", "a164", "a165", "a166", "a167", "a168", "a169", "a170", "a171", "a172", "a173", "a174", "a175", "a176", "a177", "a178", "a179", "a180", "a181", "a182", "a183", "a184", "a185", "a186", "a187", "a188", "a189", "a190", "a191", "a192", "a193", "a194", "a195", "a196", "a197", "a198", "a199", "a200", "a201", "a202", "a203", "a204", "a205", "a206", "a207", "a208", "a209", "a210", "a211", "a212", "a213", "a214", "a215", "a216", "a217", "a218", "a219", "a220", "a221", "a222", "a223", "a224", "a225", "a226", "a227", "a228", "a229", "a230", "a231", "a232", "a233", "a234", "a235", "a236", "a237", "a238", "a239", "a240", "a241", "a242", "a243", "a244", "a245", "a246", "a247", "a248", "a249", "a250", "a251", "a252", "a253", "a254", "a255");
    __builtin_amdgcn_s_sleep(64);
    if (lds[(threadIdx.x * 97) % 40960] == 1u) sink[0] = 1u;       // (never true: keeps the LDS fill alive)
}
extern "C" int hz_poison_launch(void* sink, void* stream) {
    static bool once = false;
    if (!once) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return -1;
        once = true;
    }
    hipLaunchKernelGGL(poison_kernel, dim3(512), dim3(256), 160 * 1024, (hipStream_t)stream, (unsigned*)sink);
    return (int)hipGetLastError();
}
