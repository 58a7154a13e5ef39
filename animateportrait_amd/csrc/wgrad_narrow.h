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
    // block sums (fixed order: lanes by butterfly, then the four waves)
#pragma unroll
    for (int c = 0; c < COB; ++c)
#pragma unroll
        for (int q = 0; q < Q; ++q) {
            float a = acc[c][q];
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) a += __shfl_xor(a, sh, 64);
            if ((tid & 63) == 0) red[tid >> 6][c * Q + q] = a;
        }
    __syncthreads();
    if (tid < COB * Q) {
        const int c = tid / Q, q = tid - c * Q;
        if (co0 + c < p.M)
            p.partial[((long long)blockIdx.x * p.M + co0 + c) * Q + q] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
    }
}

}  // namespace apamd
