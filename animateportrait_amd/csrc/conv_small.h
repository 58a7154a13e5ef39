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

// Sums of V values per lane over the 32 lanes of a half-wave (lane bits 4..0) by halving: at every step a lane keeps half
// of its values and receives the partner's sums of those, so V = 16 costs 8 + 4 + 2 + 1 + 1 exchanges instead of 16 x 5.
// Returns the total of value index `idx_out` (set per lane); valid in every lane.
template <int V>
__device__ __forceinline__ float halfwave_sums(float (&v)[V], int l32, int& idx_out) {
    int idx = 0, cnt = V;
#pragma unroll
    for (int bit = 16; bit >= 1; bit >>= 1) {
        if (cnt > 1) {
            const int h = cnt >> 1;
            const bool up = (l32 & bit) != 0;
#pragma unroll
            for (int j = 0; j < V / 2; ++j) {
                if (j < h) {
                    const float keep = up ? v[h + j] : v[j], give = up ? v[j] : v[h + j];
                    v[j] = keep + __shfl_xor(give, bit, 64);
                }
            }
            idx += up ? h : 0;
            cnt = h;
        } else {
            v[0] += __shfl_xor(v[0], bit, 64);
        }
    }
    idx_out = idx;
    return v[0];
}

// The gather form (one lane per output pixel, taps straight from global memory): kept for the stride-2 layers, where the
// LDS-staged form below measured slower (84 vs 68 us on the 8 -> 16 layer at B = 32: eight short tiles per CU leave the
// staging latency exposed; profiles/r03b_gen_kernel_stats.md).
template <class C>
__global__ __launch_bounds__(256) void conv_small_gather_f32(const SmallKParams p) {
    constexpr int S = C::S, COUT = C::COUT;
    __shared__ __attribute__((aligned(16))) float wl[C::MAXCIN * 9 * COUT];
    __shared__ float red[8][COUT][2];
    const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;
    int b = (int)xcd_logical_block(gridDim.x, blockIdx.x);   // neighbouring tiles (shared halo rows) on one XCD's L2
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
    // the nine taps of channel ci + 1 are in flight while channel ci multiplies (the loop is a run-time loop: left alone,
    // every channel waited for its own gathers)
    float raw[9];
    {
        const float* plane0 = p.src.data + (long long)n * Cin * HW;
#pragma unroll
        for (int t = 0; t < 9; ++t) raw[t] = plane0[off[t] >= 0 ? off[t] : 0];
    }
    for (int ci = 0; ci < Cin; ++ci) {
        float m = 0.f, r = 1.f;
        if (p.src.mean != nullptr) { m = p.src.mean[n * Cin + ci]; r = p.src.rstd[n * Cin + ci]; }
        float x[9];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = (raw[t] - m) * r;
            v = p.src.act == 1 ? fmaxf(v, 0.f) : (p.src.act == 2 ? (v > 0.f ? v : 0.2f * v) : v);
            x[t] = off[t] >= 0 ? v : 0.f;
        }
        if (ci + 1 < Cin) {
            const float* plane = p.src.data + ((long long)n * Cin + ci + 1) * HW;
#pragma unroll
            for (int t = 0; t < 9; ++t) raw[t] = plane[off[t] >= 0 ? off[t] : 0];
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
    float sv[COUT], qv[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float v = acc[co] + ((p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f);
        if (live && co < p.Cout)
            p.y[((long long)n * p.Cout + co) * p.OH * p.OW + oy * p.OW + ox] = apply_act(v, p.act);
        sv[co] = live ? v : 0.f;
        qv[co] = live ? v * v : 0.f;
    }
    if (want_stats) {
        // row sums by halving (16 + 16 exchanges instead of 16 x 5 x 2 butterfly steps)
        int co_s, co_q;
        const float s = halfwave_sums<COUT>(sv, lx, co_s);
        const float q = halfwave_sums<COUT>(qv, lx, co_q);
        constexpr int GROUP = 32 / COUT < 1 ? 1 : 32 / COUT;       // lanes per output channel
        if ((lx & (GROUP - 1)) == 0) { red[ly][co_s][0] = s; red[ly][co_q][1] = q; }
        __syncthreads();
        if (tid < COUT && tid < p.Cout) {
            float s2 = 0.f, q2 = 0.f;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) { s2 += red[r8][tid][0]; q2 += red[r8][tid][1]; }
            float* d = p.stats + (((long long)n * p.Cout + tid) * p.stat_tiles + tiy * p.tiles_x + tix) * 2;
            d[0] = s2;
            d[1] = q2;
        }
    }
}


// The input tile (with halo) of CC channels at a time goes through LDS: coalesced row loads, the producer's InstanceNorm +
// activation and the padding rule applied ONCE per element (the first form gathered every tap of every lane from global
// memory -- 72 stride-2 gathers per output pixel in the 8 -> 16 layer -- and ran at 0.17-0.22 of the HBM rate).
template <class C>
__global__ __launch_bounds__(256) void conv_small_f32(const SmallKParams p) {
    constexpr int S = C::S, COUT = C::COUT, TH = C::TH, TW = C::TW;
    constexpr int IH = (TH - 1) * S + 3, IW = (TW - 1) * S + 3, IWP = IW + (IW % 2 == 0 ? 1 : 0), CC = 8;
    __shared__ __attribute__((aligned(16))) float wl[C::MAXCIN * 9 * COUT];
    __shared__ float xt[CC * IH * IWP];
    __shared__ float red[8][COUT][2];
    const int tid = threadIdx.x, lx = tid & 31, ly = tid >> 5;
    int b = (int)xcd_logical_block(gridDim.x, blockIdx.x);   // neighbouring tiles (shared halo rows) on one XCD's L2
    const int tix = b % p.tiles_x; b /= p.tiles_x;
    const int tiy = b % p.tiles_y;
    const int n = b / p.tiles_y;
    const int oy = tiy * TH + ly, ox = tix * TW + lx;
    const int H = p.H, W = p.W, HW = H * W, Cin = p.Cin;
    for (int i = tid; i < Cin * 9 * COUT; i += 256) {
        const int co = i % COUT, t = (i / COUT) % 9, ci = i / (9 * COUT);
        wl[i] = co < p.Cout ? p.w[(co * Cin + ci) * 9 + t] : 0.f;
    }
    const bool live = oy < p.OH && ox < p.OW;
    const int iy0 = tiy * TH * S - p.pad, ix0 = tix * TW * S - p.pad;
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    for (int c0 = 0; c0 < Cin; c0 += CC) {
        const int cc = Cin - c0 < CC ? Cin - c0 : CC;
        __syncthreads();                                   // weights written / previous chunk consumed
        for (int e = tid; e < cc * IH * IW; e += 256) {
            const int c = e / (IH * IW), r = e - c * (IH * IW);
            const int ty = r / IW, tx = r - ty * IW;
            int iy = iy0 + ty, ix = ix0 + tx;
            bool ok = true;
            if (p.pad_mode == 1) {
                iy = reflect_clamp(iy, H);
                ix = reflect_clamp(ix, W);
            } else {
                ok = iy >= 0 && iy < H && ix >= 0 && ix < W;
            }
            float v = 0.f;
            if (ok) {
                const int ch = n * Cin + c0 + c;
                v = p.src.data[(long long)ch * HW + iy * W + ix];
                if (p.src.mean != nullptr) v = (v - p.src.mean[ch]) * p.src.rstd[ch];
                v = p.src.act == 1 ? fmaxf(v, 0.f) : (p.src.act == 2 ? (v > 0.f ? v : 0.2f * v) : v);
            }
            xt[(c * IH + ty) * IWP + tx] = v;
        }
        __syncthreads();
        for (int c = 0; c < cc; ++c) {
            const float* xp = xt + (c * IH + ly * S) * IWP + lx * S;
            float x[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) x[t] = xp[(t / 3) * IWP + t % 3];
#pragma unroll
            for (int t = 0; t < 9; ++t) {
                const float4* wv = reinterpret_cast<const float4*>(wl + ((c0 + c) * 9 + t) * COUT);
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
    }
    const bool want_stats = p.stats != nullptr;
    float sv[COUT], qv[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
        const float v = acc[co] + ((p.bias != nullptr && co < p.Cout) ? p.bias[co] : 0.f);
        if (live && co < p.Cout)
            p.y[((long long)n * p.Cout + co) * p.OH * p.OW + oy * p.OW + ox] = apply_act(v, p.act);
        sv[co] = live ? v : 0.f;
        qv[co] = live ? v * v : 0.f;
    }
    if (want_stats) {
        int co_s, co_q;
        const float s = halfwave_sums<COUT>(sv, lx, co_s);
        const float q = halfwave_sums<COUT>(qv, lx, co_q);
        // lanes that share the kept index hold the same total: the lowest lane of each group writes it
        constexpr int GROUP = 32 / COUT < 1 ? 1 : 32 / COUT;       // lanes per output channel
        if ((lx & (GROUP - 1)) == 0) { red[ly][co_s][0] = s; red[ly][co_q][1] = q; }
        __syncthreads();
        if (tid < COUT && tid < p.Cout) {
            float s2 = 0.f, q2 = 0.f;
#pragma unroll
            for (int r8 = 0; r8 < 8; ++r8) { s2 += red[r8][tid][0]; q2 += red[r8][tid][1]; }
            float* d = p.stats + (((long long)n * p.Cout + tid) * p.stat_tiles + tiy * p.tiles_x + tix) * 2;
            d[0] = s2;
            d[1] = q2;
        }
    }
}

}  // namespace apamd
