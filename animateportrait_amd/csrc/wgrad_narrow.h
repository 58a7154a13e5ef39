// wgrad_narrow.h -- weight gradient of layers with 1..2 input channels, as a memory stream.
//
//   dW[m][ci][ky][kx] = sum_{n, oy, ox} G[n][m][oy][ox] * A[n][ci][oy*S + ky - pad][ox*S + kx - pad]
//
// The landmark encoder's first layer (Conv2d(1, 8, 3), networks.py:1284) and the PatchGAN's first layer
// (Conv2d(1..2, 64, 4, 2, 1), networks.py:2620): Q = Cin*K*K is 9 .. 32 columns and the pixel sum runs over millions of
// positions, so as a GEMM the tile is almost empty (1.2 ms / 0.37 ms on the matrix kernel, padded operand copies
// included) while the data is one pass over G.  Here a lane owns an output pixel: it reads the Cin x K x K window of A
// (L1-shared with its neighbours) and COB gradient channels, and keeps COB x Q sums in registers; workgroups own row
// ranges and COB channels, block sums go to partial[P][M][Q] and are added in fixed order by wgrad_reduce_kernel.
// No operand copies: padding, InstanceNorm + activation of the source are applied on the way.
#pragma once
#include "conv_igemm.h"

namespace apamd {

// The COB x Q accumulators of a lane summed over the 64 lanes of its wave, into out[c * Q + q] (common.h: wave_sums_to_lds).
template <int COB, int Q>
__device__ __forceinline__ void wave_sums_to_lds(const float (&acc)[COB][Q], int tid, float* out) {
    float v[COB * Q];
#pragma unroll
    for (int i = 0; i < COB * Q; ++i) v[i] = acc[i / Q][i % Q];
    wave_sums_to_lds<COB * Q>(v, tid, out);
}

struct WgradNarrowParams {
    SrcSeg src;               // A: [N][CIN][H][W], possibly virtual
    const float* g;           // G: [N][M][GH][GW], plain
    int N, M, GH, GW, H, W, pad, pad_mode;
    int rows_per_block;       // rows of (n, oy) per workgroup
    int gwc, gwc_shift, rpi;  // threads along x (power of two <= 256, and its log2), rows per iteration (256 / gwc)
    float* partial;           // [gridDim.x][M][Q]
};

// grid: (P, ceil(M / COB)), 256 threads
template <int K, int S, int CIN, int COB>
__global__ __launch_bounds__(256) void wgrad_narrow_kernel(const WgradNarrowParams p) {
    constexpr int Q = CIN * K * K;
    __shared__ float red[4][COB * Q];
    const int tid = threadIdx.x, tx = tid & (p.gwc - 1), ty = tid >> p.gwc_shift;
    const int co0 = blockIdx.y * COB;
    const int total_rows = p.N * p.GH;
    const int r0 = blockIdx.x * p.rows_per_block;
    int r1 = r0 + p.rows_per_block;
    if (r1 > total_rows) r1 = total_rows;
    const int HW = p.H * p.W, GHW = p.GH * p.GW;
    const float slope = p.src.act == 1 ? 0.f : (p.src.act == 2 ? 0.2f : 1.f);
    float acc[COB][Q];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[c][q] = 0.f;

    for (int row = r0 + ty; row < r1; row += p.rpi) {
        const int n = row / p.GH, oy = row - n * p.GH;
        float m[CIN], rs[CIN];
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            m[ci] = 0.f;
            rs[ci] = 1.f;
            if (p.src.mean != nullptr) { m[ci] = p.src.mean[n * CIN + ci]; rs[ci] = p.src.rstd[n * CIN + ci]; }
        }
        for (int ox = tx; ox < p.GW; ox += p.gwc) {
            float gv[COB];
#pragma unroll
            for (int c = 0; c < COB; ++c)
                gv[c] = co0 + c < p.M ? p.g[((long long)n * p.M + co0 + c) * GHW + oy * p.GW + ox] : 0.f;
            float xw[Q];
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int ky = 0; ky < K; ++ky)
#pragma unroll
                    for (int kx = 0; kx < K; ++kx) {
                        int iy = oy * S + ky - p.pad, ix = ox * S + kx - p.pad;
                        bool ok = true;
                        if (p.pad_mode == 1) {
                            iy = reflect_clamp(iy, p.H);
                            ix = reflect_clamp(ix, p.W);
                        } else {
                            ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                        }
                        float v = p.src.data[((long long)n * CIN + ci) * HW + (ok ? iy * p.W + ix : 0)];
                        v = (v - m[ci]) * rs[ci];
                        v = v > 0.f ? v : slope * v;
                        xw[(ci * K + ky) * K + kx] = ok ? v : 0.f;
                    }
#pragma unroll
            for (int c = 0; c < COB; ++c)
#pragma unroll
                for (int q = 0; q < Q; ++q) acc[c][q] += gv[c] * xw[q];
        }
    }
    // block sums (fixed order: lanes by halving exchanges, then the four waves)
    wave_sums_to_lds<COB, Q>(acc, tid, red[tid >> 6]);
    __syncthreads();
    if (tid < COB * Q) {
        const int c = tid / Q, q = tid - c * Q;
        if (co0 + c < p.M)
            p.partial[((long long)blockIdx.x * p.M + co0 + c) * Q + q] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

// The PatchGAN first layer (K = 4, stride 2, zero pad 1, GW a power of two <= 256 = the x extent of a workgroup): the
// general kernel above spends ~10 VALU instructions of address / padding logic on each of the 16 CIN taps of a pixel --
// 3.5x its 128 FMAs (230 us per launch on 48 x 2 x 256 x 256).  Here the input rows of an iteration are staged once per
// workgroup in LDS with InstanceNorm / activation applied and the zero border in place, and a tap row is two 8-byte LDS
// reads at a constant offset.  The kernel is latency-bound (4 waves per SIMD at best: 128 accumulators per thread, one
// barrier per iteration), so an iteration covers PPT * rpi output rows -- PPT pixels per thread -- and the global loads
// of iteration i + 1 (its RIN = 2 PPT rpi + 2 staged rows and gradient values) fly while iteration i multiplies.
// grid: (P, ceil(M / COB)), 256 threads; dynamic LDS: 2 * CIN * RIN * (W + 2) floats
template <int CIN, int COB, int PPT>
__global__ __launch_bounds__(256) void wgrad_narrow_s2k4_kernel(const WgradNarrowParams p) {
    constexpr int K = 4, Q = CIN * K * K;
    extern __shared__ float xs[];
    __shared__ float red[4][COB * Q];
    const int tid = threadIdx.x, tx = tid & (p.gwc - 1), ty = tid >> p.gwc_shift;
    const int co0 = blockIdx.y * COB;
    const int total_rows = p.N * p.GH;
    const int r0 = blockIdx.x * p.rows_per_block;
    int r1 = r0 + p.rows_per_block;
    if (r1 > total_rows) r1 = total_rows;
    const int HW = p.H * p.W, GHW = p.GH * p.GW;
    const int RPI = p.rpi * PPT;                                   // output rows per iteration
    const int RIN = 2 * RPI + 2, WP = p.W + 2, BUF = CIN * RIN * WP;
    const int W4 = p.W >> 2;                                       // float4 groups per input row
    const float slope = p.src.act == 1 ? 0.f : (p.src.act == 2 ? 0.2f : 1.f);
    float acc[COB][Q];
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int q = 0; q < Q; ++q) acc[c][q] = 0.f;
    // staged float4 groups per thread: CIN RIN W4 / 256 = CIN (PPT rpi + 1) gwc / 256 = CIN (PPT + 1 / rpi)
    constexpr int MAXS = CIN * (PPT + 1);
    float4 sv[MAXS];
    float gn[PPT][COB];
    // what a thread stages does not change with the iteration: (channel, row j of the RIN, four columns) per group --
    // worked out once (integer divisions by run-time values cost ~40 VALU instructions each)
    int s_src[MAXS], s_dst[MAXS], s_j[MAXS], s_ci[MAXS], s_edge[MAXS];   // s_edge: 1 first / 2 last group of a row
#pragma unroll
    for (int k = 0; k < MAXS; ++k) {
        const int e = tid + k * 256;
        const int rr = e / W4, x4 = (e - rr * W4) * 4;             // staged row (ci, j), first of four columns
        const int ci = rr / RIN, j = rr - ci * RIN;
        const bool on = e < CIN * RIN * W4;
        s_ci[k] = on ? ci : 0;
        s_j[k] = on ? j : -(1 << 20);                              // (an inactive group never passes the row test)
        s_src[k] = ci * HW + (j - 1) * p.W + x4;                   // + (n CIN HW + 2 oy0 W)
        s_dst[k] = on ? rr * WP + 1 + x4 : -1;
        s_edge[k] = (x4 == 0 ? 1 : 0) | (x4 == p.W - 4 ? 2 : 0);
    }
    // fetch = raw loads only, from clamped addresses (round 6): with the normalisation inside the row test and the gradient values
    // behind a select, every one of the ~8 loads of an iteration was waited for before the next was issued -- eight serial round
    // trips against the 128 FMAs they were meant to hide under (118 us per launch at 1.07 TB/s, profiles/r05_train_hbm_per_kernel.md).
    // The row test, InstanceNorm, activation and the zeroing of inactive values happen in stage() / at the copy of the gradient.
    float mraw[CIN], rraw[CIN];
    auto fetch = [&](int rowb) __attribute__((always_inline)) {
        const int n = rowb / p.GH, oy0 = rowb - n * p.GH;
        const float* base = p.src.data + (long long)n * CIN * HW + 2 * oy0 * p.W;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            const int iy = oy0 * 2 - 1 + s_j[k];
            sv[k] = *reinterpret_cast<const float4*>((iy >= 0 && iy < p.H) ? base + s_src[k] : p.src.data);
        }
#pragma unroll
        for (int ci = 0; ci < CIN; ++ci) {
            mraw[ci] = p.src.mean != nullptr ? p.src.mean[n * CIN + ci] : 0.f;
            rraw[ci] = p.src.mean != nullptr ? p.src.rstd[n * CIN + ci] : 1.f;
        }
#pragma unroll
        for (int u = 0; u < PPT; ++u) {
            const int row = rowb + u * p.rpi + ty;
#pragma unroll
            for (int c = 0; c < COB; ++c)
                gn[u][c] = p.g[(row < r1 && co0 + c < p.M) ? ((long long)n * p.M + co0 + c) * GHW + (oy0 + u * p.rpi + ty) * p.GW + tx : 0];
        }
    };
    auto stage = [&](int buf, int rowb) __attribute__((always_inline)) {
        float* xb = xs + buf * BUF;
        const int oy0 = rowb % p.GH;
#pragma unroll
        for (int k = 0; k < MAXS; ++k) {
            if (s_dst[k] >= 0) {
                const int iy = oy0 * 2 - 1 + s_j[k];
                const bool ok = iy >= 0 && iy < p.H;
                float m = mraw[0], r = rraw[0];
#pragma unroll
                for (int ci = 1; ci < CIN; ++ci) { m = s_ci[k] == ci ? mraw[ci] : m; r = s_ci[k] == ci ? rraw[ci] : r; }
                float4 v = sv[k];
                v.x = (v.x - m) * r; v.y = (v.y - m) * r; v.z = (v.z - m) * r; v.w = (v.w - m) * r;
                v.x = v.x > 0.f ? v.x : slope * v.x; v.y = v.y > 0.f ? v.y : slope * v.y;
                v.z = v.z > 0.f ? v.z : slope * v.z; v.w = v.w > 0.f ? v.w : slope * v.w;
                float* d = xb + s_dst[k];                          // staged column = image column + 1
                d[0] = ok ? v.x : 0.f; d[1] = ok ? v.y : 0.f; d[2] = ok ? v.z : 0.f; d[3] = ok ? v.w : 0.f;
                if (s_edge[k] & 1) d[-1] = 0.f;
                if (s_edge[k] & 2) d[4] = 0.f;
            }
        }
    };
    int buf = 0;
    if (r0 < r1) {
        fetch(r0);
        stage(0, r0);
    }
    __syncthreads();
    for (int rowb = r0; rowb < r1; rowb += RPI, buf ^= 1) {
        // rows rowb .. rowb + RPI - 1 belong to one image (RPI divides GH, r0 is a multiple of RPI)
        float gv[PPT][COB];
#pragma unroll
        for (int u = 0; u < PPT; ++u)
#pragma unroll
            for (int c = 0; c < COB; ++c) gv[u][c] = (rowb + u * p.rpi + ty < r1 && co0 + c < p.M) ? gn[u][c] : 0.f;
        const bool more = rowb + RPI < r1;
        if (more) fetch(rowb + RPI);
        const float* xb = xs + buf * BUF;
#pragma unroll
        for (int u = 0; u < PPT; ++u)
#pragma unroll
            for (int ci = 0; ci < CIN; ++ci)
#pragma unroll
                for (int ky = 0; ky < K; ++ky) {
                    const float2* rp = reinterpret_cast<const float2*>(xb + (ci * RIN + (u * p.rpi + ty) * 2 + ky) * WP + tx * 2);
                    const float2 a = rp[0], b = rp[1];
                    const float xw[4] = {a.x, a.y, b.x, b.y};
#pragma unroll
                    for (int kx = 0; kx < K; ++kx)
#pragma unroll
                        for (int c = 0; c < COB; ++c) acc[c][(ci * K + ky) * K + kx] += gv[u][c] * xw[kx];
                }
        if (more) stage(buf ^ 1, rowb + RPI);
        __syncthreads();                                           // next buffer written; this one free for the one after
    }
    // block sums (fixed order: lanes by halving exchanges, then the four waves)
    wave_sums_to_lds<COB, Q>(acc, tid, red[tid >> 6]);
    __syncthreads();
    if (tid < COB * Q) {
        const int c = tid / Q, q = tid - c * Q;
        if (co0 + c < p.M)
            p.partial[((long long)blockIdx.x * p.M + co0 + c) * Q + q] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

}  // namespace apamd
