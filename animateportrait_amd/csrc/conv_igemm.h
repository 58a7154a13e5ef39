// conv_igemm.h -- im2col-free implicit-GEMM convolution for gfx950 (CDNA4), exact fp32.
//
// One kernel template covers every convolution-shaped layer of the Module2 generator and
// PatchGAN discriminators (reference: Module2/models/networks.py:1218-1282, 2329-2421,
// 2620-2643): Conv2d k3/k4/k7, stride 1/2, zero or reflection padding, ConvTranspose2d
// (as four sub-pixel phases), and -- with swapped weight roles -- their data gradients.
//
// GEMM view per workgroup:  D[co, pix] = sum_{tap, ci} Wp[tap][ci][co] * X[ci][pix + tap]
//   * A operand (weights) and B operand (activations) are both read from LDS with one
//     ds_read_b32 per lane per MFMA operand; v_mfma_f32_32x32x2_f32 consumes the channel
//     pair (ci, ci+1) held by the two half-waves (exact fp32, 64 FLOP/clk/SIMD).
//   * The activation tile is staged ONCE per channel chunk with its halo; the KHxKW taps
//     are shifted reads of that tile (no im2col expansion anywhere).
//   * The loader fuses: torch.cat of up to 3 sources, InstanceNorm + ReLU/LeakyReLU of the
//     producer layer ((x-mean)*rstd, act), zero / reflection padding.
//   * The epilogue fuses: bias, activation (LeakyReLU/tanh), and the per-(n,cout) partial
//     sum / sum-of-squares that InstanceNorm of THIS layer needs (deterministic, no atomics).
//
// Wave layout: 256 threads = 4 waves arranged WCO x WPX; each wave owns MT x NT MFMA tiles
// of 32(cout) x 32(pixels of one output row).  Workgroup tile = (WCO*MT*32) couts x
// (WPX*NT) rows x 32 columns.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "common.h"

namespace apamd {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kMaxTaps = 49;
constexpr int kMaxSeg = 3;

struct SrcSeg {
    const float* data;
    const float* mean;
    const float* rstd;
    int C;            // real channels
    int act;          // 0 none, 1 relu, 2 lrelu(0.2)
    int chunk_begin;  // first chunk index of this segment
    int pad_;
};

struct ConvKParams {
    SrcSeg seg[kMaxSeg];
    int nseg;
    int N, H, W;          // input dims
    int Cout;
    int OH, OW;           // output grid iterated by this launch (per-phase grid for transposed)
    int dy0, dx0;         // input coordinate of LDS-tile origin for output (0,0): iy = oy*S + dy0 + ly
    int pad_mode;         // 0 zero, 1 reflect
    float* y;
    long long o_nstride;  // output strides (elements)
    long long o_cstride;
    int o_rstride;
    int osy, osx, oy_off, ox_off;   // output pixel = (oy*osy + oy_off, ox*osx + ox_off)
    const float* wp;      // packed weights of this launch: [co_tile][chunk][tap][ci][CO_TILE]
    const float* bias;    // Cout or null
    int act;              // epilogue: 0 none, 1 relu, 2 lrelu, 3 tanh
    float* stats;         // [N][Cout][stat_tiles][2] or null
    int stat_tiles;
    int stat_tile_off;
    int ntaps;
    int nchunks;
    int tiles_x, tiles_y, co_tiles;
    int cin_pad;          // nchunks * CI
    int wfloats;          // floats of one packed (co_tile, chunk) weight block, padded to 256
    int ablate;           // debugging/ablation only (APAMD_ABLATE): 1 no refill, 2 no barrier, 4 MFMA-only, 8 no epilogue
    unsigned tap_bits;    // K == 0 kernels only (sub-pixel phases, <= 4 taps): bit 2t = ly, bit 2t+1 = lx of tap t
    // conv_bf16x3 only: the four sub-pixel phases of a stride-2 transposed convolution as ONE launch.  The cout tile
    // index runs over nphase * co_tiles_phase "virtual" tiles (phase-major); phase ph overrides dy0 / dx0 / oy_off /
    // ox_off / stat_tile_off with its table entry.  nphase <= 1: plain launch, the scalar fields above apply.
    int nphase, co_tiles_phase;
    int ph_dy0[4], ph_dx0[4], ph_oy[4], ph_ox[4], ph_stat[4];
    int s2d_div;              // > 0: space-to-depth form of a 3x3 stride-2 layer -- chunk c belongs to input phase c / s2d_div,
    unsigned s2d_mask[4];     // whose existing taps are s2d_mask[phase] (run-time-tap split-bf16 kernels skip the others)
    unsigned ph_tapmask[4];   // bit t: tap t of the 2 x 2 window exists in that phase (split-bf16 run-time-tap kernels skip
                              // the MFMAs of the others -- their weights are zero: 7 of the 16 taps of a fused 3x3 up-convolution)
    int o_octet;              // conv_bf16x3 run-time-tap / row families: output as y[n][Cout/8][OH*OW][8] (ap_conv2d_fwd_octet)
    // conv_bf16x3<..., FNORM = 1> only (ap_conv2d_fwd_norm): InstanceNorm of the layer's own output inside the epilogue
    int fn_act;               // activation after the normalisation
    float fn_eps;
    double fn_inv_count;      // 1 / (OH * OW)
    const float* fn_res_oct;  // residual added after the normalisation: channel-octet fp32, or
    const float* fn_res_nchw; // ... plain NCHW fp32, or neither
    float* fn_y_oct;          // fp32 result, channel-octet layout (or null)
    void* fn_xs;              // split-bf16 copy of the result (or null)
    float* fn_mean;           // [N * Cout] finished statistics
    float* fn_rstd;
    int fn_debug;             // APAMD_FNORM_DEBUG (timing experiments, WRONG results): 1 no waiting for the group, 2 no output traffic
    unsigned* fn_counters;    // [N * co_tiles * 2], zero at launch: arrivals of a group's workgroups (stats is the exchange buffer)
    // experiment build only (-DAPAMD_ABLATION, tools/cycle_account.py): s_memtime stamps of every wave, kept in `stamp_words`
    // dwords of LDS per wave behind the stage buffers (byte offset stamp_lds_off) and copied to stamps[workgroup][wave][...] at exit
    unsigned* stamps;
    int stamp_lds_off, stamp_words;
};

// K_ > 0: dense K x K taps at compile-time offsets (tap t = ky*K + kx).  K_ == 0: up to four taps inside a
// 2 x 2 window given at run time (the sub-pixel phases of a stride-2 transposed convolution).
template <int CI_, int S_, int K_, int WCO_, int MT_, int WPX_, int NT_>
struct ConvCfg {
    static constexpr int CI = CI_, S = S_, K = K_, WCO = WCO_, MT = MT_, WPX = WPX_, NT = NT_;
    static constexpr int EXT = K_ > 0 ? K_ - 1 : 1;
    static constexpr int TH = WPX * NT;
    static constexpr int CO_TILE = WCO * MT * 32;
    static constexpr int IH = (TH - 1) * S + EXT + 1;
    static constexpr int IW = 31 * S + EXT + 1;
    static constexpr int PLANE = IH * IW;
    static constexpr int XE = CI * PLANE;
    static constexpr int NE = (XE + 255) / 256;
    // waves per SIMD the register allocator must leave room for (3 x 4-wave workgroups per CU when the
    // accumulator tile is 4 x 16 registers; LDS is sized on the host to match)
    static constexpr int MINW = (MT * NT <= 4 && NE <= 6) ? 3 : (NE <= 24 ? 2 : 1);
    static_assert(WCO * WPX == 4, "4 waves per workgroup");
    static_assert(CI % 2 == 0, "channel chunk must hold whole (ci, ci+1) pairs");
    // dynamic LDS floats for `ntaps` taps, `nbuf` (1 or 2) pipeline buffers, cin_pad channels
    static int wfloats(int ntaps) { return (ntaps * CI * CO_TILE + 255) / 256 * 256; }
    static size_t lds_floats(int ntaps, int nbuf, int cin_pad) {
        size_t f = (size_t)nbuf * ((size_t)wfloats(ntaps) + XE) + 2 * (size_t)cin_pad;
        size_t red = (size_t)WPX * CO_TILE * 2;
        return f > red ? f : red;
    }
};

__device__ __forceinline__ int reflect_clamp(int i, int n) {
    i = i < 0 ? -i : i;
    i = i >= n ? 2 * (n - 1) - i : i;
    // masked-out rows/cols of ragged tiles can land further out; keep the address legal
    i = i < 0 ? 0 : (i >= n ? n - 1 : i);
    return i;
}

__device__ __forceinline__ void glds16(const float* gsrc, float* ldst) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)ldst, 16, 0, 0);
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : 0.2f * v;
    if (act == 3) return tanhf(v);
    return v;
}

template <class C>
__global__ __launch_bounds__(256, C::MINW) void conv_igemm_f32(const ConvKParams p) {
    constexpr int CI = C::CI, S = C::S, MT = C::MT, NT = C::NT, WCO = C::WCO;
    constexpr int IW = C::IW, PLANE = C::PLANE, XE = C::XE, NE = C::NE, CO_TILE = C::CO_TILE;
    extern __shared__ __attribute__((aligned(16))) float smem[];

    const int tid = threadIdx.x;
    const int wave = tid >> 6;
    const int lane = tid & 63;
    const int half = lane >> 5;
    const int l32 = lane & 31;
    const int wco = wave % WCO;
    const int wpx = wave / WCO;

    // ---- workgroup -> tile.  Consecutive logical ids share the activation tile (different
    // cout tiles) and are kept on one XCD (observed dispatch: block b -> XCD b % 8), so the
    // second reader hits that XCD's L2.  Pure speed: any placement is correct.
    int logical;
    {
        const int nblk = gridDim.x, b = blockIdx.x;
        const int q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int cot = logical % p.co_tiles;
    int t_ = logical / p.co_tiles;
    const int tx = t_ % p.tiles_x;
    t_ /= p.tiles_x;
    const int ty = t_ % p.tiles_y;
    const int n = t_ / p.tiles_y;

    const int oy0 = ty * C::TH, ox0 = tx * 32;
    const int iy0 = oy0 * S + p.dy0, ix0 = ox0 * S + p.dx0;
    const int H = p.H, W = p.W, HW = H * W;

    const int wfloats = p.wfloats;
    const int nbuf = p.nchunks > 1 ? 2 : 1;
    float* const wbuf = smem;
    float* const xbuf = smem + nbuf * wfloats;
    float* const s_mean = xbuf + nbuf * XE;
    float* const s_rstd = s_mean + p.cin_pad;

    // ---- per-(n, channel) normalisation constants of the producer layers -> LDS (once)
    for (int c = tid; c < p.cin_pad; c += 256) {
        const int chunk = c / CI;
        int s = 0;
        if (p.nseg > 1 && chunk >= p.seg[1].chunk_begin) s = 1;
        if (p.nseg > 2 && chunk >= p.seg[2].chunk_begin) s = 2;
        const int cs = c - p.seg[s].chunk_begin * CI;
        float m = 0.f, r = 1.f;
        if (p.seg[s].mean != nullptr && cs < p.seg[s].C) {
            m = p.seg[s].mean[n * p.seg[s].C + cs];
            r = p.seg[s].rstd[n * p.seg[s].C + cs];
        }
        s_mean[c] = m;
        s_rstd[c] = r;
    }

    // ---- loader geometry (identical for every chunk): element e of the [CI][IH][IW] tile
    int goff[NE];
#pragma unroll
    for (int k = 0; k < NE; ++k) {
        const int e = tid + k * 256;
        const int ci = e / PLANE;
        const int r = e - ci * PLANE;
        const int ly = r / IW;
        const int lx = r - ly * IW;
        int gy = iy0 + ly, gx = ix0 + lx;
        bool ok = e < XE;
        if (p.pad_mode == 1) {
            gy = reflect_clamp(gy, H);
            gx = reflect_clamp(gx, W);
        } else {
            ok = ok && gy >= 0 && gy < H && gx >= 0 && gx < W;
        }
        goff[k] = ok ? ci * HW + gy * W + gx : -1;
    }

    float xr[NE];
    auto seg_of = [&](int chunk) {
        int s = 0;
        if (p.nseg > 1 && chunk >= p.seg[1].chunk_begin) s = 1;
        if (p.nseg > 2 && chunk >= p.seg[2].chunk_begin) s = 2;
        return s;
    };
    auto issue_x = [&](int chunk) {
        const int s = seg_of(chunk);
        const int cbase = (chunk - p.seg[s].chunk_begin) * CI;
        const float* base = p.seg[s].data + ((long long)n * p.seg[s].C + cbase) * HW;
        const int cleft = p.seg[s].C - cbase;  // valid channels in this chunk (may exceed CI)
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = tid + k * 256;
            const int ci = e / PLANE;
            const bool ok = goff[k] >= 0 && ci < cleft;
            xr[k] = ok ? base[goff[k]] : 0.f;
        }
    };
    auto commit_x = [&](int chunk, float* dst) {
        const int s = seg_of(chunk);
        const int cbase = (chunk - p.seg[s].chunk_begin) * CI;
        const int cleft = p.seg[s].C - cbase;
        const int act = p.seg[s].act;
        const bool norm = p.seg[s].mean != nullptr;
#pragma unroll
        for (int k = 0; k < NE; ++k) {
            const int e = tid + k * 256;
            if (e < XE) {
                const int ci = e / PLANE;
                float v = xr[k];
                if (norm) v = (v - s_mean[chunk * CI + ci]) * s_rstd[chunk * CI + ci];
                if (act == 1) v = v > 0.f ? v : 0.f;
                else if (act == 2) v = v > 0.f ? v : 0.2f * v;
                // padding zeros are zeros of the NORMALISED tensor
                if (!(goff[k] >= 0 && ci < cleft)) v = 0.f;
                dst[e] = v;
            }
        }
    };
    // weights of one chunk are one contiguous block in the packed buffer: async copy to LDS
    const float* wsrc0 = p.wp + (long long)cot * p.nchunks * wfloats;
    // (global_load_lds: each wave instruction moves 64 lanes x 16 B to a wave-uniform LDS base)
    auto issue_w = [&](int chunk, float* dst) {
        const float* src = wsrc0 + (long long)chunk * wfloats;
        for (int j = wave; j < (wfloats >> 8); j += 4) glds16(src + j * 256 + lane * 4, dst + j * 256);
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int q = 0; q < NT; ++q)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][q][r] = 0.f;

    // ---- prologue
    __syncthreads();  // s_mean/s_rstd visible
    issue_x(0);
    issue_w(0, wbuf);
    commit_x(0, xbuf);
    __syncthreads();

    const int a_lane = half * CO_TILE + wco * MT * 32 + l32;
    const int b_lane = half * PLANE + (wpx * NT) * S * IW + l32 * S;

    for (int chunk = 0; chunk < p.nchunks; ++chunk) {
        const int cur = chunk & 1;
        const bool more = chunk + 1 < p.nchunks;
        // Refill of the other pipeline buffer.  On the dense path both halves are placed INSIDE the MFMA
        // stream (loads right after the first step, normalise + ds_write three quarters through), so
        // their VALU/VMEM/DS issue slots hide under the 64-cycle MFMAs instead of serialising with them.
        const bool refill = more && !AP_ABLATE(p, 1);
        if (C::K == 0 && refill) {
            issue_x(chunk + 1);
            issue_w(chunk + 1, wbuf + (cur ^ 1) * wfloats);
        }
        const float* Wc = wbuf + cur * wfloats + a_lane;
        const float* Xc = xbuf + cur * XE + b_lane;

        if (AP_ABLATE(p, 4)) {
            float a0 = Wc[0], b0 = Xc[0];
            for (int s = 0; s < p.ntaps * (CI / 2); ++s) {
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int q = 0; q < NT; ++q)
                        acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[m][q], 0, 0, 0);
            }
        } else if constexpr (C::K > 0) {
            // Dense taps: fully unrolled, LDS offsets are instruction immediates.  Operands of step s+1
            // are fetched into the other register set BEFORE the MFMAs of step s are issued, and the
            // interleave {MT+NT ds_reads, MT*NT MFMAs} is pinned, so that one wave alone keeps its
            // SIMD's matrix pipe busy (fp32 MFMA: 64 cycles each, LDS latency < one step).
            // Large kernels (7x7) keep the row loop rolled: one row of taps = K * CI/2 pipelined steps.
            constexpr int ROWS_UNROLLED = (C::K <= 4) ? C::K : 1;
            constexpr int NS = ROWS_UNROLLED * C::K * (CI / 2);
            for (int ky0 = 0; ky0 < C::K; ky0 += ROWS_UNROLLED) {
                const float* Wr = Wc + ky0 * (C::K * CI * CO_TILE);
                const float* Xr = Xc + ky0 * IW;
                float a[2][MT], b[2][NT];
                auto fetch = [&](int s, float* av, float* bv) {
                    const int t = s / (CI / 2), cp = s % (CI / 2);
                    const int toff = (t / C::K) * IW + (t % C::K);
#pragma unroll
                    for (int m = 0; m < MT; ++m) av[m] = Wr[(t * CI + cp * 2) * CO_TILE + m * 32];
#pragma unroll
                    for (int q = 0; q < NT; ++q) bv[q] = Xr[toff + cp * 2 * PLANE + q * S * IW];
                };
                fetch(0, a[0], b[0]);
#pragma unroll
                for (int s = 0; s < NS; ++s) {
                    if (s + 1 < NS) fetch(s + 1, a[(s + 1) & 1], b[(s + 1) & 1]);
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s & 1][m], b[s & 1][q], acc[m][q], 0, 0, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, MT + NT, 0);   // DS reads of step s+1
                    __builtin_amdgcn_sched_group_barrier(0x008, MT * NT, 0);   // MFMAs of step s
                    if (s == 0 && ky0 == 0 && refill) {
                        issue_x(chunk + 1);
                        issue_w(chunk + 1, wbuf + (cur ^ 1) * wfloats);
                    }
                    if (s == (C::K <= 4 ? (NS * 3) / 4 : NS - 1) && ky0 == (C::K <= 4 ? 0 : C::K - 2) && refill)
                        commit_x(chunk + 1, xbuf + (cur ^ 1) * XE);
                }
            }
        } else {
            for (int t = 0; t < p.ntaps; ++t) {
                const unsigned tb = p.tap_bits >> (2 * t);
                const int toff = (int)(tb & 1u) * IW + (int)((tb >> 1) & 1u);
                const float* wt = Wc + t * (CI * CO_TILE);
#pragma unroll
                for (int cp = 0; cp < CI / 2; ++cp) {
                    float a[MT], b[NT];
#pragma unroll
                    for (int m = 0; m < MT; ++m) a[m] = wt[cp * 2 * CO_TILE + m * 32];
#pragma unroll
                    for (int q = 0; q < NT; ++q) b[q] = Xc[toff + cp * 2 * PLANE + q * S * IW];
#pragma unroll
                    for (int m = 0; m < MT; ++m)
#pragma unroll
                        for (int q = 0; q < NT; ++q)
                            acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], b[q], acc[m][q], 0, 0, 0);
                }
            }
        }
        if (C::K == 0 && refill) commit_x(chunk + 1, xbuf + (cur ^ 1) * XE);
        if (!AP_ABLATE(p, 2)) __syncthreads();
    }

    if (AP_ABLATE(p, 8)) {
        if (acc[0][0][0] == 123.456f) p.y[0] = 1.f;   // keep the accumulators live
        return;
    }
    // ---- epilogue: bias, statistics, activation, store.
    // MFMA 32x32 C/D layout: column j = lane & 31, row i = (r & 3) + 8 * (r >> 2) + 4 * half.
    float* sred = smem;  // [WPX][CO_TILE][2], safe: all waves passed the final barrier
    const int co_base = cot * CO_TILE + wco * MT * 32;
    const int ox = ox0 + l32;
    const bool want_stats = p.stats != nullptr;
#pragma unroll
    for (int m = 0; m < MT; ++m) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = (r & 3) + 8 * (r >> 2) + 4 * half;
            const int co = co_base + m * 32 + i;
            const bool cok = co < p.Cout;
            const float bv = (p.bias != nullptr && cok) ? p.bias[co] : 0.f;
            float s = 0.f, q2 = 0.f;
#pragma unroll
            for (int q = 0; q < NT; ++q) {
                const int oy = oy0 + wpx * NT + q;
                const float v = acc[m][q][r] + bv;
                if (cok && oy < p.OH && ox < p.OW) {
                    s += v;
                    q2 += v * v;
                    p.y[(long long)n * p.o_nstride + (long long)co * p.o_cstride +
                        (long long)(oy * p.osy + p.oy_off) * p.o_rstride + (ox * p.osx + p.ox_off)] =
                        apply_act(v, p.act);
                }
            }
            if (want_stats) {
#pragma unroll
                for (int sh = 1; sh < 32; sh <<= 1) {
                    s += __shfl_xor(s, sh, 64);
                    q2 += __shfl_xor(q2, sh, 64);
                }
                if (l32 == 0) {
                    float* d = sred + ((wpx * CO_TILE) + wco * MT * 32 + m * 32 + i) * 2;
                    d[0] = s;
                    d[1] = q2;
                }
            }
        }
    }
    if (want_stats) {
        __syncthreads();
        if (tid < CO_TILE) {
            const int co = cot * CO_TILE + tid;
            if (co < p.Cout) {
                float s = 0.f, q2 = 0.f;
#pragma unroll
                for (int w = 0; w < C::WPX; ++w) {
                    s += sred[(w * CO_TILE + tid) * 2];
                    q2 += sred[(w * CO_TILE + tid) * 2 + 1];
                }
                float* d = p.stats + (((long long)n * p.Cout + co) * p.stat_tiles + p.stat_tile_off +
                                      ty * p.tiles_x + tx) * 2;
                d[0] = s;
                d[1] = q2;
            }
        }
    }
}

typedef void (*ConvKernelFn)(const ConvKParams);

struct ConvKernelInfo {
    int CI, S, K, EXT, WCO, MT, WPX, NT;
    int TH, CO_TILE, XE;
    const void* fn;
    size_t (*lds_floats)(int, int, int);
    int (*wfloats)(int);
};

template <class C>
ConvKernelInfo make_info() {
    ConvKernelInfo k;
    k.CI = C::CI; k.S = C::S; k.K = C::K; k.EXT = C::EXT; k.WCO = C::WCO; k.MT = C::MT; k.WPX = C::WPX; k.NT = C::NT;
    k.TH = C::TH; k.CO_TILE = C::CO_TILE; k.XE = C::XE;
    k.fn = reinterpret_cast<const void*>(&conv_igemm_f32<C>);
    k.lds_floats = &C::lds_floats;
    k.wfloats = &C::wfloats;
    return k;
}

}  // namespace apamd
