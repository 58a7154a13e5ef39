// conv_tsmall.h -- ConvTranspose2d(C, 1..4, 4, stride 2, pad 1) as a memory stream.
//
// This is the data gradient of the PatchGAN's first layer (Conv2d(1..2, 64, 4, 2, 1), networks.py:2620) w.r.t. the
// generated frame -- needed in backward_G for every discriminator (geomgm_ifw_fore_model.py:677-780).  With 1..2
// output channels an MFMA tile is 30/32 padding and the four sub-pixel phases of the general kernel each re-read the
// 64-channel gradient (4 x 81 us per layer at B=32); the arithmetic is 1 GMAC.  Here a lane owns one INPUT-resolution
// pixel (q, r): it reads the 3 x 3 neighbourhood of that pixel once per channel (coalesced rows, L1-shared with its
// neighbours) and accumulates the 2 x 2 output quad (2q + py, 2r + px) of every output channel in registers:
//     y[co][2q + py][2r + px] = sum_ci sum_{(ky,dy) in Y(py)} sum_{(kx,dx) in Y(px)} x[ci][q + dy][r + dx] * w[ci][co][ky][kx]
//     Y(0) = {(1, 0), (3, -1)},  Y(1) = {(0, +1), (2, 0)}            (oy = 2 iy - 1 + ky)
// Weights are wave-uniform scalars; the quad rows are written as 256-byte contiguous wave stores.
#pragma once
#include "conv_igemm.h"

namespace apamd {

struct TSmallParams {
    SrcSeg src;               // [N][C][H][W], possibly virtual (InstanceNorm + activation of the producer)
    int N, C, H, W, Cout;
    const float* w;           // IOHW [C][Cout][4][4]
    const float* bias;        // [Cout] or null
    int act;
    float* y;                 // [N][Cout][2H][2W]
};

// grid: (ceil(W / 32), ceil(H / 8), N), 256 threads
template <int COUT>
__global__ __launch_bounds__(256) void conv_tsmall_f32(const TSmallParams p) {
    const int r = blockIdx.x * 32 + (threadIdx.x & 31), q = blockIdx.y * 8 + (threadIdx.x >> 5), n = blockIdx.z;
    const int H = p.H, W = p.W, HW = H * W;
    const float slope = p.src.act == 1 ? 0.f : (p.src.act == 2 ? 0.2f : 1.f);
    // neighbourhood offsets and validity (zero padding), clamped so that every load is legal
    int off[3][3];
    bool ok[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            const int yy = q + a - 1, xx = r + b - 1;
            ok[a][b] = yy >= 0 && yy < H && xx >= 0 && xx < W;
            off[a][b] = ok[a][b] ? yy * W + xx : 0;
        }
    float acc[COUT][2][2];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co][0][0] = acc[co][0][1] = acc[co][1][0] = acc[co][1][1] = 0.f;
    typedef const float __attribute__((address_space(4))) cfloat;    // uniform -> scalar loads
    cfloat* const wc = (cfloat*)(uintptr_t)p.w;
    const float* src = p.src.data + (long long)n * p.C * HW;
    // the neighbourhood of channel ci + 1 is in flight while channel ci multiplies (a run-time loop: left alone, every
    // channel waited for its own nine loads)
    float nx[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) nx[a][b] = src[off[a][b]];
    for (int ci = 0; ci < p.C; ++ci) {
        float m = 0.f, rs = 1.f;
        if (p.src.mean != nullptr) { m = p.src.mean[n * p.C + ci]; rs = p.src.rstd[n * p.C + ci]; }
        float v[3][3];
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) {
                float t = (nx[a][b] - m) * rs;
                t = t > 0.f ? t : slope * t;
                v[a][b] = ok[a][b] ? t : 0.f;
            }
        if (ci + 1 < p.C) {
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b) nx[a][b] = src[(long long)(ci + 1) * HW + off[a][b]];
        }
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
            if (co >= p.Cout) continue;                          // Cout = 3 runs the 4-channel instantiation
            cfloat* w = wc + ((long long)ci * p.Cout + co) * 16;
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    // (k, neighbourhood index) pairs of the phase: Y(0) = {(1, 1), (3, 0)}, Y(1) = {(0, 2), (2, 1)}
                    const int ky0 = py ? 0 : 1, ay0 = py ? 2 : 1, ky1 = py ? 2 : 3, ay1 = py ? 1 : 0;
                    const int kx0 = px ? 0 : 1, bx0 = px ? 2 : 1, kx1 = px ? 2 : 3, bx1 = px ? 1 : 0;
                    acc[co][py][px] += (v[ay0][bx0] * w[ky0 * 4 + kx0] + v[ay0][bx1] * w[ky0 * 4 + kx1]) +
                                       (v[ay1][bx0] * w[ky1 * 4 + kx0] + v[ay1][bx1] * w[ky1 * 4 + kx1]);
                }
        }
    }
    if (q >= H || r >= W) return;
    const int OW = 2 * W;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        if (co >= p.Cout) break;
        const float bv = p.bias != nullptr ? p.bias[co] : 0.f;
        float* dst = p.y + (((long long)n * p.Cout + co) * 2 * H + 2 * q) * OW + 2 * r;
#pragma unroll
        for (int py = 0; py < 2; ++py)
            *reinterpret_cast<float2*>(dst + py * OW) =
                make_float2(apply_act(acc[co][py][0] + bv, p.act), apply_act(acc[co][py][1] + bv, p.act));
    }
}

}  // namespace apamd
