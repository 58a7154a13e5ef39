// mfma_peak.hip -- practical bf16 MFMA ceiling of the device under the occupancy the split-bf16 conv runs at.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak ; run on the GPU box.
// Prints TFLOP/s of dependency-free v_mfma_f32_32x32x16_bf16 streams with W waves per SIMD (W = 1, 2) and
// with / without interleaved LDS fragment reads (12 ds_read_b128 per 24 MFMAs, the conv's mix).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int LDSREADS, int RANDOM_DATA, int TRIPLES = 0>
__global__ __launch_bounds__(512) void mfma_stream(float* out, int iters) {
    extern __shared__ uint4 lds[];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += blockDim.x) {
        // pseudo-random bf16 values of magnitude ~2^-6, random signs: realistic operand toggling (all-ones data
        // lets the matrix pipe run cooler, and faster, than real activations do)
        unsigned h = (unsigned)i * 2654435761u + blockIdx.x * 40503u;
        unsigned w[4];
        if (RANDOM_DATA == 2) {
            // what the conv really multiplies: bf16 heads and tails of fp32 values -- slots alternate between "weights"
            // (N(0, 0.02), head / tail) and "activations" (ReLU(N(0, 1)): half of them exact zeros, head / tail)
            const bool tail = (i >> 6) & 1, act = (i >> 7) & 1;
            for (int j = 0; j < 4; ++j) {
                unsigned pr[2];
                for (int e = 0; e < 2; ++e) {
                    float u = 0.f;                                   // ~N(0,1): sum of 4 uniforms, centred
                    for (int q = 0; q < 4; ++q) { h = h * 1664525u + 1013904223u; u += (float)(h >> 8) * (1.f / 16777216.f); }
                    float v = (u - 2.f) * 1.7320508f;
                    if (act) v = v > 0.f ? v : 0.f; else v *= 0.02f;
                    const __bf16 hi = (__bf16)v;
                    const __bf16 lo = (__bf16)(v - (float)hi);
                    const __bf16 pick = tail ? lo : hi;
                    pr[e] = (unsigned)__builtin_bit_cast(unsigned short, pick);
                }
                w[j] = pr[0] | (pr[1] << 16);
            }
            lds[i] = make_uint4(w[0], w[1], w[2], w[3]);
            continue;
        }
        for (int j = 0; j < 4; ++j) {
            h = h * 1664525u + 1013904223u;
            const unsigned lo = RANDOM_DATA ? (0x3c00u | ((h >> 9) & 0x807fu)) : 0x3f80u;
            h = h * 1664525u + 1013904223u;
            const unsigned hi = RANDOM_DATA ? (0x3c00u | ((h >> 9) & 0x807fu)) : 0x3f80u;
            w[j] = lo | (hi << 16);
        }
        lds[i] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    __syncthreads();
    f32x16 acc[8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    bf16x8 a[2][4], b[2][8];
    const bf16x8* src = reinterpret_cast<const bf16x8*>(lds) + (tid & 63);
#pragma unroll
    for (int i = 0; i < 4; ++i) a[0][i] = a[1][i] = src[i * 64];
#pragma unroll
    for (int i = 0; i < 8; ++i) b[0][i] = b[1][i] = src[(4 + i) * 64];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 8; ++t) {
            const int cb = t & 1;
            if (LDSREADS) {
                // LDSREADS == 1: the conv's mix (12 reads per 24 MFMAs); otherwise LDSREADS reads per 24 MFMAs
                constexpr int NA = LDSREADS == 1 ? 4 : (LDSREADS >= 4 ? LDSREADS / 3 : LDSREADS);
                constexpr int NB = (LDSREADS == 1 ? 12 : LDSREADS) - NA;
#pragma unroll
                for (int i = 0; i < NA; ++i) a[cb ^ 1][i % 4] = src[((t * 12 + i) & 63) * 64];
#pragma unroll
                for (int i = 0; i < NB; ++i) b[cb ^ 1][i % 8] = src[((t * 12 + 4 + i) & 63) * 64];
            }
            if (TRIPLES) {
                // three dependent MFMAs back to back on every accumulator (the naive bf16x3 order)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
#pragma unroll
                        for (int k = 0; k < 3; ++k)
                            acc[m * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][m + 2 * (k & 1)], b[cb][q + 4 * (k >> 1)],
                                                                                     acc[m * 4 + q], 0, 0, 0);
            } else {
#pragma unroll
            for (int k = 0; k < 3; ++k)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        acc[m * 4 + q] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[cb][m + 2 * (k & 1)], b[cb][q + 4 * (k >> 1)],
                                                                                 acc[m * 4 + q], 0, 0, 0);
            }
            if constexpr (LDSREADS != 0) {
                constexpr int NR = LDSREADS == 1 ? 12 : LDSREADS;
#pragma unroll
                for (int i = 0; i < NR; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 24 / NR, 0);
                }
            }
        }
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][7];
    if (s == 123.456f) out[0] = s;
}

template <int LDSREADS, int RANDOM_DATA, int TRIPLES = 0>
static void run(const char* label, int blocks, int threads, size_t ldsbytes) {
    float* out;
    hipMalloc(&out, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_stream<LDSREADS, RANDOM_DATA, TRIPLES>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL((mfma_stream<LDSREADS, RANDOM_DATA, TRIPLES>), dim3(blocks), dim3(threads), ldsbytes, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double waves = (double)blocks * threads / 64;
        const double flops = waves * iters * 8 * 24 * 2.0 * 32 * 32 * 16;
        if (rep == 3) printf("%-44s %8.3f ms  %8.1f TFLOP/s\n", label, ms, flops / (ms * 1e-3) / 1e12);
    }
    hipFree(out);
}

// `mfma_peak loop <seconds> <data: 0 ones / 1 random / 2 real split>`: the register-only stream back to back for that long
// (tools/power_record.py reads the chip's throttle accumulators around it).  Prints READY, then RESULT with the mean rate.
template <int DATA>
static void loop_for(int cus, double seconds) {
    float* out;
    hipMalloc(&out, 64);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&mfma_stream<0, DATA, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mfma_stream<0, DATA, 0>), dim3(cus), dim3(256), 150 * 1024, 0, out, iters);
    hipDeviceSynchronize();
    printf("READY\n");
    fflush(stdout);
    double total_ms = 0.0, first_ms = 0.0, last_ms = 0.0;
    long n = 0;
    while (total_ms < seconds * 1e3) {
        hipEventRecord(e0, 0);
        for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((mfma_stream<0, DATA, 0>), dim3(cus), dim3(256), 150 * 1024, 0, out, iters);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e0, e1);
        if (n == 0) first_ms = ms / 10;
        last_ms = ms / 10;
        total_ms += ms;
        n += 10;
    }
    const double flops = (double)cus * 4 * iters * 8 * 24 * 2.0 * 32 * 32 * 16;
    printf("RESULT mfma_peak loop data=%d: %ld launches in %.2f s, mean %.1f TFLOP/s executed (first ten %.1f, last ten %.1f)\n", DATA, n,
           total_ms * 1e-3, flops * n / (total_ms * 1e-3) / 1e12, flops / (first_ms * 1e-3) / 1e12, flops / (last_ms * 1e-3) / 1e12);
    hipFree(out);
}

int main(int argc, char** argv) {
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    if (argc >= 3 && std::string(argv[1]) == "loop") {
        const double seconds = atof(argv[2]);
        const int data = argc >= 4 ? atoi(argv[3]) : 2;
        if (data == 0) loop_for<0>(cus, seconds);
        else if (data == 1) loop_for<1>(cus, seconds);
        else loop_for<2>(cus, seconds);
        return 0;
    }
    printf("CUs: %d\n", cus);
    run<0, 0>("1 wave/SIMD, registers only, all-ones data", cus, 256, 150 * 1024);
    run<1, 0>("1 wave/SIMD, 12 ds_read_b128 / 24 MFMA, ones", cus, 256, 150 * 1024);
    run<0, 1>("1 wave/SIMD, registers only, random data", cus, 256, 150 * 1024);
    run<1, 1>("1 wave/SIMD, 12 ds_read_b128 / 24 MFMA, random", cus, 256, 150 * 1024);
    run<1, 1>("2 waves/SIMD, 12 ds_read_b128 / 24 MFMA, random", cus, 512, 150 * 1024);
    run<1, 2>("1 wave/SIMD, 12 reads / 24 MFMA, REAL split data", cus, 256, 150 * 1024);
    run<0, 2>("1 wave/SIMD, registers only, REAL split data", cus, 256, 150 * 1024);
    run<1, 2>("1 wave/SIMD, 12 reads / 24 MFMA, REAL split, 64 CUs", cus / 4, 256, 150 * 1024);
    run<8, 1>("1 wave/SIMD, 8 ds_read_b128 / 24 MFMA, random", cus, 256, 150 * 1024);
    run<6, 1>("1 wave/SIMD, 6 ds_read_b128 / 24 MFMA, random", cus, 256, 150 * 1024);
    run<4, 1>("1 wave/SIMD, 4 ds_read_b128 / 24 MFMA, random", cus, 256, 150 * 1024);
    run<1, 1>("1 wave/SIMD, 12 reads / 24 MFMA, random, 128 CUs", cus / 2, 256, 150 * 1024);
    run<1, 1>("1 wave/SIMD, 12 reads / 24 MFMA, random, 64 CUs", cus / 4, 256, 150 * 1024);
    run<0, 1>("1 wave/SIMD, registers only, random, 64 CUs", cus / 4, 256, 150 * 1024);
    run<0, 1, 1>("1 wave/SIMD, registers, random, dependent triples", cus, 256, 150 * 1024);
    run<0, 0, 1>("1 wave/SIMD, registers, ones, dependent triples", cus, 256, 150 * 1024);
    return 0;
}
