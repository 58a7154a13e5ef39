// conv_head.h -- the PatchGAN output layer: Conv2d(8*ndf, 1, 4, stride 1, pad 1) on a 31 x 31 map
// (reference: Module2/models/networks.py:2643), forward and weight gradient.
//
// One output channel and ~900 output pixels per image: as a GEMM this is a 1-row matrix (a 32 x 32 MFMA tile is 31/32
// padding, and the fp32 implicit-GEMM kernel needs 0.39 ms for 1.2 GFLOP); as a stream it is 63 MB of activations.
// Both kernels keep a map row in one half-wave -- lane = column, W <= 31 so that lane 31 of each half holds the zero
// padding column between the two halves -- and reach the column neighbours of the 4 taps with DPP wave shifts, so every
// input element is loaded exactly once per use (coalesced 124-byte rows) and there is no LDS traffic in the loop.
//   forward:  workgroup = (image, band of RB output rows); its 32 half-waves walk the channels 32 apart, each with the
//             band's RB accumulators in registers; one fixed-order reduction over the 32 streams through LDS.
//   wgrad:    workgroup = channel; its 8 half-waves walk the images 8 apart, 16 tap accumulators per lane, g rows and
//             x rows both in registers; lane reduction by shuffles, then a fixed-order sum over the streams.
#pragma once
#include "conv_igemm.h"

namespace apamd {

struct HeadParams {
    SrcSeg src;               // [N][C][H][W], possibly virtual
    int N, C, H, W, OH, OW;   // K = 4, pad = 1, stride 1
    const float* w;           // forward: OIHW [1][C][4][4]
    const float* bias;        // forward: [1] or null
    int act;                  // forward epilogue
    float* y;                 // forward: [N][1][OH][OW]
    const float* g;           // wgrad: [N][1][OH][OW]
    float* dw;                // wgrad: [1][C][4][4]
};

constexpr int kHeadMaxW = 31;             // a row and its zero column fit a half-wave
constexpr int kHeadMaxH = 32;             // wgrad keeps a whole plane in registers
constexpr int kHeadBand = 10;             // forward: output rows per workgroup

// lane i <- lane i-1 (lane 0 <- 0) / lane i <- lane i+1 (lane 63 <- 0)
__device__ __forceinline__ float head_shr1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float head_shl1(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}

// grid: (N, ceil(OH / RB)), 1024 threads
template <int RB>
__global__ __launch_bounds__(1024) void conv_head_fwd_kernel(const HeadParams p) {
    constexpr int NS = 32;
    __shared__ float red[NS][RB][32];
    const int n = blockIdx.x, oy0 = blockIdx.y * RB, tid = threadIdx.x;
    const int l = tid & 31, s = tid >> 5;
    const float slope = p.src.act == 1 ? 0.f : (p.src.act == 2 ? 0.2f : 1.f);
    const bool col = l < p.W;
    float acc[RB];
#pragma unroll
    for (int o = 0; o < RB; ++o) acc[o] = 0.f;
    for (int c = s; c < p.C; c += NS) {
        const float* src = p.src.data + ((long long)n * p.C + c) * p.H * p.W + l;
        float m = 0.f, r = 1.f;
        if (p.src.mean != nullptr) { m = p.src.mean[n * p.C + c]; r = p.src.rstd[n * p.C + c]; }
        float w[16];
        {
            const float4* wp = reinterpret_cast<const float4*>(p.w + c * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 t = wp[q];
                w[q * 4] = t.x; w[q * 4 + 1] = t.y; w[q * 4 + 2] = t.z; w[q * 4 + 3] = t.w;
            }
        }
        float v[RB + 3];
        bool ok[RB + 3];
#pragma unroll
        for (int i = 0; i < RB + 3; ++i) {
            const int iy = oy0 - 1 + i;
            ok[i] = col && iy >= 0 && iy < p.H;
            v[i] = ok[i] ? src[iy * p.W] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < RB + 3; ++i) {
            float x = (v[i] - m) * r;
            x = x > 0.f ? x : slope * x;
            x = ok[i] ? x : 0.f;
            const float xm = head_shr1(x), xp = head_shl1(x), xpp = head_shl1(xp);
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int o = i - ky;
                if (o >= 0 && o < RB)
                    acc[o] += (w[ky * 4] * xm + w[ky * 4 + 1] * x) + (w[ky * 4 + 2] * xp + w[ky * 4 + 3] * xpp);
            }
        }
    }
#pragma unroll
    for (int o = 0; o < RB; ++o) red[s][o][l] = acc[o];
    __syncthreads();
    if (tid < RB * 32) {
        const int o = tid >> 5, oy = oy0 + o;
        float t = 0.f;
#pragma unroll 8
        for (int q = 0; q < NS; ++q) t += red[q][o][l];
        if (oy < p.OH && l < p.OW) {
            const float b = p.bias != nullptr ? p.bias[0] : 0.f;
            p.y[((long long)n * p.OH + oy) * p.OW + l] = apply_act(t + b, p.act);
        }
    }
}

// grid: (C), 256 threads; H <= kHeadMaxH
static __global__ __launch_bounds__(256) void conv_head_wgrad_kernel(const HeadParams p) {
    constexpr int NS = 8, HM = kHeadMaxH;
    __shared__ float red[NS][16];
    const int c = blockIdx.x, tid = threadIdx.x;
    const int l = tid & 31, s = tid >> 5;
    const float slope = p.src.act == 1 ? 0.f : (p.src.act == 2 ? 0.2f : 1.f);
    const bool col = l < p.W, gcol = l < p.OW;
    float acc[16];
#pragma unroll
    for (int t = 0; t < 16; ++t) acc[t] = 0.f;
    for (int n = s; n < p.N; n += NS) {
        const float* src = p.src.data + ((long long)n * p.C + c) * p.H * p.W + l;
        const float* gp = p.g + (long long)n * p.OH * p.OW + l;
        float m = 0.f, r = 1.f;
        if (p.src.mean != nullptr) { m = p.src.mean[n * p.C + c]; r = p.src.rstd[n * p.C + c]; }
        float v[HM], g[HM];
#pragma unroll
        for (int i = 0; i < HM; ++i) {
            v[i] = (col && i < p.H) ? src[i * p.W] : 0.f;
            g[i] = (gcol && i < p.OH) ? gp[i * p.OW] : 0.f;
        }
#pragma unroll
        for (int i = 0; i < HM; ++i) {
            float x = (v[i] - m) * r;
            x = x > 0.f ? x : slope * x;
            x = (col && i < p.H) ? x : 0.f;
            const float xm = head_shr1(x), xp = head_shl1(x), xpp = head_shl1(xp);
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int oy = i - ky + 1;                   // pad = 1
                if (oy >= 0 && oy < HM) {
                    acc[ky * 4] += g[oy] * xm;
                    acc[ky * 4 + 1] += g[oy] * x;
                    acc[ky * 4 + 2] += g[oy] * xp;
                    acc[ky * 4 + 3] += g[oy] * xpp;
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < 16; ++t) {
        float a = acc[t];
#pragma unroll
        for (int sh = 1; sh < 32; sh <<= 1) a += __shfl_xor(a, sh, 64);
        if (l == 0) red[s][t] = a;
    }
    __syncthreads();
    if (tid < 16) {
        float t = 0.f;
#pragma unroll
        for (int q = 0; q < NS; ++q) t += red[q][tid];
        p.dw[c * 16 + tid] = t;
    }
}

}  // namespace apamd
