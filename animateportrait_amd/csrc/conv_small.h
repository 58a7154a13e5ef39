// conv_small.h -- direct (vector-ALU) 3x3 convolution for narrow layers (Cin <= 16, Cout = 8 or 16).
//
// The landmark encoder (reference: Module2/models/networks.py:1284-1296, model_landmark_trans: 1 -> 8 -> 16 -> 16
// channels at 256^2 .. 64^2, stride 1 / 2 / 2) has so few channels that an MFMA tile is mostly padding and the
// layers are pure memory streams: 1.2 GFLOP against 100 MB of activations.  One lane = one output pixel and ALL
// output channels; weights sit in LDS as [ci][tap][cout] and are read as broadcast ds_read_b128; the input taps are
// plain cached global loads (neighbouring lanes share them) with the producer's InstanceNorm + activation and the
// zero / reflection padding applied on the fly.  Epilogue as everywhere: bias + activation, or the InstanceNorm
// partial statistics of the raw output.
#pragma once
#include "conv_igemm.h"

namespace apamd {

struct SmallKParams {
    SrcSeg src;            // one source segment (C = Cin)
    int N, H, W, Cin, Cout, OH, OW, pad, pad_mode;
    float* y;
    const float* w;        // OIHW [Cout][Cin][3][3]
    const float* bias;
    int act;
    float* stats;          // [N][Cout][stat_tiles][2] or null
    int stat_tiles, tiles_x, tiles_y;
};

template <int S_, int COUT_>
struct SmallCfg {
    static constexpr int S = S_, COUT = COUT_, TH = 8, TW = 32, MAXCIN = 16;
};

template <class C>
__global__ __launch_bounds__(256) void conv_small_f32(const SmallKParams p) {
    constexpr int S = C::S, COUT = C::COUT;
    __shared__ __attribute__((aligned(16))) float wl[C::MAXCIN * 9 * COUT];
    __shared__ float red[8][COUT][2];
    const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;
    int b = blockIdx.x;
    const int tix = b % p.tiles_x; b /= p.tiles_x;
    const int tiy = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy = tiy * C::TH + ly, ox = tix * C::TW + lx;
    const int H = p.H, W = p.W, HW = H * W, Cin = p.Cin;
    for (int i = tid; i < Cin * 9 * COUT; i += 256) {
        const int co = i % COUT, t = (i / COUT) % 9, ci = i / (9 * COUT);
        wl[i] = co < p.Cout ? p.w[(co * Cin + ci) * 9 + t] : 0.f;
    }
    __syncthreads();
    const bool live = oy < p.OH && ox < p.OW;
    // tap offsets inside a channel plane (-1: zero padding)
    int off[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        int iy = oy * S + t / 3 - p.pad, ix = ox * S + t % 3 - p.pad;
        bool ok = live;
        if (p.pad_mode == 1) {
            iy = reflect_clamp(iy, H);
            ix = reflect_clamp(ix, W);
        } else {
            ok = ok && iy >= 0 && iy < H && ix >= 0 && ix < W;
        }
        off[t] = ok ? iy * W + ix : -1;
    }
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int ci = 0; ci < Cin; ++ci) {
        const float* plane = p.src.data + ((long long)n * Cin + ci) * HW;
        float m = 0.f, r = 1.f;
        if (p.src.mean != nullptr) { m = p.src.mean[n * Cin + ci]; r = p.src.rstd[n * Cin + ci]; }
        float x[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = off[t] >= 0 ? (plane[off[t]] - m) * r : 0.f;
            if (off[t] >= 0) v = p.src.act == 1 ? fmaxf(v, 0.f) : (p.src.act == 2 ? (v > 0.f ? v : 0.2f * v) : v);
            x[t] = v;
        }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4* wv = reinterpret_cast<const float4*>(wl + (ci * 9 + t) * COUT);
#pragma unroll
            for (int q = 0; q < COUT / 4; ++q) {
                const float4 w4 = wv[q];
                acc[q * 4 + 0] += x[t] * w4.x;
                acc[q * 4 + 1] += x[t] * w4.y;
                acc[q * 4 + 2] += x[t] * w4.z;
                acc[q * 4 + 3] += x[t] * w4.w;
            }
        }
    }
    const bool want_stats = p.stats != nullptr;
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        float v = acc[co] + ((p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f);
        if (live && co < p.Cout)
            p.y[((long long)n * p.Cout + co) * p.OH * p.OW + oy * p.OW + ox] = apply_act(v, p.act);
        if (want_stats) {
            float s = live ? v : 0.f, q = live ? v * v : 0.f;
#pragma unroll
            for (int sh = 1; sh < 32; sh <<= 1) {
                s += __shfl_xor(s, sh, 64);
                q += __shfl_xor(q, sh, 64);
            }
            if (lx == 0) { red[ly][co][0] = s; red[ly][co][1] = q; }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid < COUT && tid < p.Cout) {
            float s = 0.f, q = 0.f;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) { s += red[r8][tid][0]; q += red[r8][tid][1]; }
            float* d = p.stats + (((long long)n * p.Cout + tid) * p.stat_tiles + tiy * p.tiles_x + tix) * 2;
            d[0] = s;
            d[1] = q;
        }
    }
}

}  // namespace apamd
