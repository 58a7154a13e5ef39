// conv_prepass.h -- the streaming passes around the split-bf16 convolutions (conv_bf16x3.h): the norm / residual / split pass between
// two convolutions, the row and space-to-depth split copies, and the weight packer.  Included by conv_host.hip only (the
// convolution kernels themselves are instantiated one tile family per translation unit, conv_bf3_inst_*.hip).
#pragma once
#include "conv_bf16x3.h"

namespace apamd {

// ---- activation pre-pass: XS[n][part][cg][HW + 1] (16-byte slots of 8 bf16) = split(act((x - mean) * rstd));
// slot HW of every plane is all-zero (the source of every out-of-image tap: a fixed offset from the plane, so the
// convolution's per-lane DMA offsets are constants of the tile).  grid: (ceil(HW/256), C/8, N).  HBM-bound: reads
// C*HW*4 B and writes the same amount per sample.
//
// The same pass is the generator's whole "between two convolutions" step (norm_split_kernel):
//   v = act((x - mean) * rstd) [+ (res - res_mean) * res_rstd]        InstanceNorm + activation + residual add
//   y  = v   (fp32, optional)       the materialised feature (ResnetBlock output, networks.py:2358-2360)
//   xs = split(v) (optional)        what the next split-bf16 convolution stages
// and, when the producing convolution's per-tile (sum, sum of squares) are passed instead of finished statistics,
// it finalises mean / rstd itself (fp64, as instnorm_finalize_kernel) and stores them for later consumers -- so a
// Conv -> IN -> ReLU -> Conv link costs one streaming pass, not finalize + apply + split.
struct NormSplitParams {
    const float* x;
    const float* mean;        // finished statistics [N*C], or null
    const float* rstd;
    const float* partials;    // or the conv epilogue's [N*C][tiles][2] partial sums (then mean/rstd above are null)
    int tiles;
    double inv_count;
    float eps;
    float* mean_out;          // where the finalised statistics go (partials mode)
    float* rstd_out;
    int act;
    const float* res;         // residual [N, C, HW] or null
    const float* res_mean;    // its statistics or null (plain residual)
    const float* res_rstd;
    const uint4* res_xs;      // or: the residual as its split copy (head + tail planes; inference keeps the residual stream of
                              // the ResNet trunk only in that form, 2^-17 relative per block) -- then res is null
    float* y;                 // fp32 output or null
    uint4* xs;                // split output or null
    int heads_only;           // 1: only the head planes of xs are written (consumers in AP_PRECISION_BF16 never read tails)
    int xs_relu;              // 1: the split copy holds relu(v) while y holds v -- the next layer's `activation -> conv` of a
                              // pre-activation residual stream (FlowUnet_v2's ResidualBlock) without a second pass
    int N, C, HW;
};

// grid: (ceil(HW / (256 * VEC)), C/8, N); VEC pixels per thread (4 when HW % 4 == 0)
// XB16: x holds bf16 values (a raw convolution output stored by ap_conv2d_fwd_bf16out): same element offsets, 2-byte elements
// RES: 0 = no residual, 1 = fp32 residual (p.res, optionally with statistics), 2 = the residual as its split copy (p.res_xs) -- a
// template parameter so that the load phase has no branch around its loads
template <int VEC, bool XB16 = false, int RES = 0>
__global__ __launch_bounds__(256) void norm_split_kernel(const NormSplitParams p) {
    __shared__ float s_m[8], s_r[8];
    __shared__ int s_bad[8];
    __shared__ double s_red[2][4];
    const int cg = blockIdx.y, n = blockIdx.z, C = p.C, HW = p.HW, CG = C >> 3;
    const int tid = threadIdx.x;
    const bool normed = p.partials != nullptr || p.mean != nullptr;
    // ---- every global load of the thread is issued BEFORE the statistics are finalised (round 6): the block used to wait for the
    // partial sums, reduce them, synchronise and only then ask for its data -- two dependent HBM round trips per block, and a
    // workgroup lives for little more than that (profiles/r06_stream_readers.md).  The first partial of wave 0 goes out ahead of
    // the data, so the in-order vmcnt wait of the reduction does not wait for the data as well.
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // (scalar: the branches on it do not touch EXEC)
    // (no branch around it -- every thread loads 8 bytes from a clamped address, x itself when there are no partials: a load under
    // a branch came back through a phi copy, and the copy was waited for on the spot)
    const float2 part0 = *(p.partials != nullptr
        ? reinterpret_cast<const float2*>(p.partials) + ((long long)n * C + cg * 8 + ((tid >> 3) & 7)) * p.tiles + ((tid & 7) < p.tiles ? (tid & 7) : 0)
        : reinterpret_cast<const float2*>(p.x));
    const int pix = (blockIdx.x * 256 + tid) * VEC;
    const bool live = pix < HW;
    // The load phase holds nothing but loads: raw words into registers, addresses clamped instead of branched on (a dead thread
    // reads pixel 0 of its plane: a cache hit), conversions and arithmetic after the
    // statistics.  With the bf16 -> fp32 shifts (or the residual statistics' loads) inside this loop the compiler waited for
    // every channel's data before it asked for the next one's: 8 serial HBM round trips per thread (the 4.4 TB/s of round 5).
    typedef unsigned XRaw __attribute__((ext_vector_type(XB16 ? (VEC == 4 ? 2 : 1) : VEC)));
    typedef float RRaw __attribute__((ext_vector_type(VEC)));
    XRaw xraw[8];
    RRaw rraw[8];
    constexpr bool has_res = RES == 1, has_res_xs = RES == 2;
    {
        const int lpix = live ? pix : 0;
        const long long off0 = ((long long)n * C + cg * 8) * HW + lpix;
        const unsigned char* xb = reinterpret_cast<const unsigned char*>(p.x);
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (XB16 && VEC == 1) {
                xraw[c][0] = *reinterpret_cast<const unsigned short*>(xb + (off0 + (long long)c * HW) * 2);
            } else {
                xraw[c] = *reinterpret_cast<const XRaw*>(xb + (off0 + (long long)c * HW) * (XB16 ? 2 : 4));
            }
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (has_res) rraw[c] = *reinterpret_cast<const RRaw*>(p.res + off0 + (long long)c * HW);
            else rraw[c] = RRaw(0.f);
        }
    }
    uint4 hq[VEC], lq[VEC];
    if constexpr (has_res_xs) {
        const uint4* rh = p.res_xs + ((long long)(n * 2 + 0) * CG + cg) * (HW + 1) + (live ? pix : 0);
        const uint4* rl = p.res_xs + ((long long)(n * 2 + 1) * CG + cg) * (HW + 1) + (live ? pix : 0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) { hq[j] = rh[j]; lq[j] = rl[j]; }
    }
    // statistics of the residual (a virtual feature): eight threads fetch them into LDS, everybody reads them after the barrier
    __shared__ float s_rm[8], s_rr[8];
    if (RES == 1 && wave == 1) {
        const int c = tid & 7;
        const bool rs = has_res && p.res_mean != nullptr;
        const float m_ = rs ? p.res_mean[n * C + cg * 8 + c] : 0.f, r_ = rs ? p.res_rstd[n * C + cg * 8 + c] : 1.f;
        if (tid < 72) { s_rm[c] = m_; s_rr[c] = r_; }
    }
    if (p.partials != nullptr) {
        // wave 0: lane = (channel, 8-way tile split); fp64 sums of fp32 partials are exact, so the grouping
        // does not change the result
        if (wave == 0) {
            const int c = tid >> 3, sub = tid & 7;
            const float2* pp = reinterpret_cast<const float2*>(p.partials) + ((long long)n * C + cg * 8 + c) * p.tiles;
            double s = sub < p.tiles ? (double)part0.x : 0.0, q = sub < p.tiles ? (double)part0.y : 0.0;
            for (int t = sub + 8; t < p.tiles; t += 8) {
                const float2 v = pp[t];
                s += (double)v.x;
                q += (double)v.y;
            }
#pragma unroll
            for (int sh = 1; sh < 8; sh <<= 1) {
                s += __shfl_xor(s, sh, 64);
                q += __shfl_xor(q, sh, 64);
            }
            if (sub == 0) {
                const double m = s * p.inv_count;
                double var = q * p.inv_count - m * m;
                var = var > 0.0 ? var : 0.0;
                s_m[c] = (float)m;
                s_r[c] = (float)(1.0 / sqrt(var + (double)p.eps));
                s_bad[c] = (m * m > (double)kInstNormRefineRatio * var) ? 1 : 0;
            }
        }
        __syncthreads();
        // ill-conditioned planes (|mean| >> std: E[x^2] - E[x]^2 of fp32 sums is rounding noise there) are recomputed
        // from the data with the shifted two-pass formula, as instnorm_finalize_kernel does.  Every pixel block of the
        // plane repeats the same deterministic sum (rare path: read amplification only for such planes).
        for (int c = 0; c < 8; ++c) {
            if (!s_bad[c]) continue;                           // block-uniform
            const float m0 = s_m[c];
            const float* px = p.x + ((long long)n * C + cg * 8 + c) * HW;
            double s = 0.0, q = 0.0;
            for (int k = tid; k < HW; k += 256) {
                const float d = (XB16 ? (float)reinterpret_cast<const __bf16*>(p.x)[((long long)n * C + cg * 8 + c) * HW + k] : px[k]) - m0;
                s += (double)d;
                q += (double)d * (double)d;
            }
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                s += __shfl_xor(s, sh, 64);
                q += __shfl_xor(q, sh, 64);
            }
            if ((tid & 63) == 0) { s_red[0][tid >> 6] = s; s_red[1][tid >> 6] = q; }
            __syncthreads();
            if (tid == 0) {
                const double S = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
                const double Q = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
                const double dm = S * p.inv_count;
                double var = Q * p.inv_count - dm * dm;
                var = var > 0.0 ? var : 0.0;
                s_m[c] = (float)((double)m0 + dm);
                s_r[c] = (float)(1.0 / sqrt(var + (double)p.eps));
            }
            __syncthreads();
        }
        if (blockIdx.x == 0 && tid < 8) {
            p.mean_out[n * C + cg * 8 + tid] = s_m[tid];
            p.rstd_out[n * C + cg * 8 + tid] = s_r[tid];
        }
    } else {
        if (p.mean != nullptr && tid < 8) {
            s_m[tid] = p.mean[n * C + cg * 8 + tid];
            s_r[tid] = p.rstd[n * C + cg * 8 + tid];
        }
        __syncthreads();
    }
    // ---- now the data: raw words -> fp32
    float v[8][VEC], rv[8][VEC];
    float rm[8], rr[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        rm[c] = RES == 1 ? s_rm[c] : 0.f;
        rr[c] = RES == 1 ? s_rr[c] : 1.f;
        if constexpr (XB16 && VEC == 4) {          // a bf16 is the upper half of the fp32 of the same value
            v[c][0] = __uint_as_float(xraw[c][0] << 16); v[c][1] = __uint_as_float(xraw[c][0] & 0xffff0000u);
            v[c][2] = __uint_as_float(xraw[c][1] << 16); v[c][3] = __uint_as_float(xraw[c][1] & 0xffff0000u);
        } else if constexpr (XB16) {
            v[c][0] = __uint_as_float(xraw[c][0] << 16);
        } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v[c][j] = __uint_as_float(xraw[c][j]);
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) rv[c][j] = rraw[c][j];
    }
    if constexpr (has_res_xs) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const unsigned hw_[4] = {hq[j].x, hq[j].y, hq[j].z, hq[j].w}, lw_[4] = {lq[j].x, lq[j].y, lq[j].z, lq[j].w};
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const unsigned hb = (c & 1) ? (hw_[c >> 1] & 0xffff0000u) : (hw_[c >> 1] << 16);
                const unsigned lb = (c & 1) ? (lw_[c >> 1] & 0xffff0000u) : (lw_[c >> 1] << 16);
                rv[c][j] = __uint_as_float(hb) + __uint_as_float(lb);
            }
        }
    }
    if (VEC == 1 && !live) return;
    uint4* ph = nullptr;
    uint4* pl = nullptr;
    if (p.xs != nullptr) {
        ph = p.xs + ((long long)(n * 2 + 0) * CG + cg) * (HW + 1);
        pl = p.xs + ((long long)(n * 2 + 1) * CG + cg) * (HW + 1);
        if (blockIdx.x == 0 && tid == 0) ph[HW] = pl[HW] = make_uint4(0u, 0u, 0u, 0u);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        if (!live) continue;
        const long long off = ((long long)n * C + cg * 8 + c) * HW + pix;
        const float m = normed ? s_m[c] : 0.f, r = normed ? s_r[c] : 1.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float t = normed ? (v[c][j] - m) * r : v[c][j];
            t = p.act == 1 ? fmaxf(t, 0.f) : (p.act == 2 ? (t > 0.f ? t : 0.2f * t) : t);
            if (has_res || has_res_xs) t += (rv[c][j] - rm[c]) * rr[c];
            v[c][j] = t;
        }
        if (p.y != nullptr) {
            if constexpr (VEC == 4) *reinterpret_cast<float4*>(p.y + off) = make_float4(v[c][0], v[c][1], v[c][2], v[c][3]);
            else p.y[off] = v[c][0];
        }
    }
    if (p.xs == nullptr) return;
    if constexpr (VEC == 1) {
        bf16x8 hv, lv;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            __bf16 h, l;
            split_bf16(p.xs_relu ? fmaxf(v[c][0], 0.f) : v[c][0], h, l);
            hv[c] = h;
            lv[c] = l;
        }
        *reinterpret_cast<bf16x8*>(ph + pix) = hv;
        if (!p.heads_only) *reinterpret_cast<bf16x8*>(pl + pix) = lv;
    } else {
        // A thread owns VEC consecutive pixels (16-byte loads), but a 16-byte store per lane at a 64-byte lane
        // stride writes every cache line in four partial pieces.  Transpose the block's slots through LDS so
        // that store k of lane T lands on slot k*256 + T of the block's span: 1 KiB contiguous per wave.
        // Staging position of (thread i, pixel j) = j * (256 + 4) + i: conflict-free both ways.
        constexpr int LP = 256 + 4;
        __shared__ uint4 stage[2][VEC * LP];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            bf16x8 hv, lv;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                __bf16 h, l;
                split_bf16(live ? (p.xs_relu ? fmaxf(v[c][j], 0.f) : v[c][j]) : 0.f, h, l);
                hv[c] = h;
                lv[c] = l;
            }
            *reinterpret_cast<bf16x8*>(&stage[0][j * LP + tid]) = hv;
            *reinterpret_cast<bf16x8*>(&stage[1][j * LP + tid]) = lv;
        }
        __syncthreads();
        const int base = blockIdx.x * 256 * VEC;
#pragma unroll
        for (int k = 0; k < VEC; ++k) {
            const int g = k * 256 + tid;                       // slot inside the block's span = pixel base + g
            const int i = g / VEC, j = g % VEC;                // owner thread and its pixel
            if (base + g < HW) {
                ph[base + g] = stage[0][j * LP + i];
                if (!p.heads_only) pl[base + g] = stage[1][j * LP + i];
            }
        }
    }
}

// ---- norm_split on a CHANNEL-OCTET raw output (round 6): x is [N][C/8][HW][8] fp32 as ap_conv2d_fwd_octet writes it -- straight from
// the MFMA accumulator layout, no LDS transposition in the convolution's epilogue -- and the only output is the split copy.  A lane
// pair owns a pixel (lane = 2 * pixel + half: four channels each), so a wave's loads are 1 KiB contiguous; the 8-channel slot of a
// pixel is put together with two quad-permute exchanges: the even lane stores the head slot, the odd lane the tail slot -- 512
// contiguous bytes per plane and wave-instruction, no LDS staging.  Residual: none (RES = 0) or the previous block's split copy
// (RES = 2: the even lane loads the head slot, the odd lane the tail slot, halves exchanged the same way).  The inference trunk:
// convolution -> this pass -> convolution (networks.py:2329-2361), 15 of its 24 passes.   grid: (ceil(HW / 512), C / 8, N)
template <int RES>
__global__ __launch_bounds__(256) void norm_split_oct_kernel(const NormSplitParams p) {
    __shared__ float s_m[8], s_r[8];
    __shared__ int s_bad[8];
    __shared__ double s_red[2][4];
    const int cg = blockIdx.y, n = blockIdx.z, C = p.C, HW = p.HW, CG = C >> 3;
    const int tid = threadIdx.x, half = tid & 1;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const float4* xo = reinterpret_cast<const float4*>(p.x) + ((long long)n * CG + cg) * HW * 2;
    // ---- load phase: nothing but loads (section "load-phase rule" of DESIGN.md)
    const float2 part0 = *(p.partials != nullptr
        ? reinterpret_cast<const float2*>(p.partials) + ((long long)n * C + cg * 8 + ((tid >> 3) & 7)) * p.tiles + ((tid & 7) < p.tiles ? (tid & 7) : 0)
        : reinterpret_cast<const float2*>(p.x));
    constexpr int VP = 4;
    float4 xv[VP];
    uint4 rq[VP];
    const int pbase = blockIdx.x * (128 * VP) + (tid >> 1);
#pragma unroll
    for (int j = 0; j < VP; ++j) {
        const int px = pbase + j * 128;
        xv[j] = xo[(long long)(px < HW ? px : 0) * 2 + half];
    }
    if constexpr (RES == 2) {
        const uint4* rp = p.res_xs + ((long long)(n * 2 + half) * CG + cg) * (HW + 1);        // even lane: head plane, odd lane: tail plane
#pragma unroll
        for (int j = 0; j < VP; ++j) {
            const int px = pbase + j * 128;
            rq[j] = rp[px < HW ? px : 0];
        }
    }
    // ---- statistics from the convolution's partial tiles (as norm_split_kernel; ill-conditioned planes from the data)
    if (p.partials != nullptr) {
        if (wave == 0) {
            const int c = tid >> 3, sub = tid & 7;
            const float2* pp = reinterpret_cast<const float2*>(p.partials) + ((long long)n * C + cg * 8 + c) * p.tiles;
            double s = sub < p.tiles ? (double)part0.x : 0.0, q = sub < p.tiles ? (double)part0.y : 0.0;
            for (int t = sub + 8; t < p.tiles; t += 8) {
                const float2 v = pp[t];
                s += (double)v.x;
                q += (double)v.y;
            }
#pragma unroll
            for (int sh = 1; sh < 8; sh <<= 1) {
                s += __shfl_xor(s, sh, 64);
                q += __shfl_xor(q, sh, 64);
            }
            if (sub == 0) {
                const double m = s * p.inv_count;
                double var = q * p.inv_count - m * m;
                var = var > 0.0 ? var : 0.0;
                s_m[c] = (float)m;
                s_r[c] = (float)(1.0 / sqrt(var + (double)p.eps));
                s_bad[c] = (m * m > (double)kInstNormRefineRatio * var) ? 1 : 0;
            }
        }
        __syncthreads();
        for (int c = 0; c < 8; ++c) {
            if (!s_bad[c]) continue;                           // block-uniform
            const float m0 = s_m[c];
            const float* px = p.x + ((long long)n * CG + cg) * HW * 8 + c;
            double s = 0.0, q = 0.0;
            for (int k = tid; k < HW; k += 256) {
                const float d = px[(long long)k * 8] - m0;
                s += (double)d;
                q += (double)d * (double)d;
            }
#pragma unroll
            for (int sh = 1; sh < 64; sh <<= 1) {
                s += __shfl_xor(s, sh, 64);
                q += __shfl_xor(q, sh, 64);
            }
            if ((tid & 63) == 0) { s_red[0][tid >> 6] = s; s_red[1][tid >> 6] = q; }
            __syncthreads();
            if (tid == 0) {
                const double S = (s_red[0][0] + s_red[0][1]) + (s_red[0][2] + s_red[0][3]);
                const double Q = (s_red[1][0] + s_red[1][1]) + (s_red[1][2] + s_red[1][3]);
                const double dm = S * p.inv_count;
                double var = Q * p.inv_count - dm * dm;
                var = var > 0.0 ? var : 0.0;
                s_m[c] = (float)((double)m0 + dm);
                s_r[c] = (float)(1.0 / sqrt(var + (double)p.eps));
            }
            __syncthreads();
        }
        if (blockIdx.x == 0 && tid < 8) {
            p.mean_out[n * C + cg * 8 + tid] = s_m[tid];
            p.rstd_out[n * C + cg * 8 + tid] = s_r[tid];
        }
    } else {
        if (tid < 8) {
            s_m[tid] = p.mean != nullptr ? p.mean[n * C + cg * 8 + tid] : 0.f;
            s_r[tid] = p.mean != nullptr ? p.rstd[n * C + cg * 8 + tid] : 1.f;
        }
        __syncthreads();
    }
    uint4* const ph = p.xs + ((long long)(n * 2 + 0) * CG + cg) * (HW + 1);
    uint4* const pl = p.xs + ((long long)(n * 2 + 1) * CG + cg) * (HW + 1);
    if (blockIdx.x == 0 && tid == 0) ph[HW] = pl[HW] = make_uint4(0u, 0u, 0u, 0u);
    float m4[4], r4[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) { m4[i] = s_m[half * 4 + i]; r4[i] = s_r[half * 4 + i]; }
    const float slope = p.act == 1 ? 0.f : (p.act == 2 ? 0.2f : 1.f);
    auto swap1 = [](unsigned v) { return (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, false); };   // quad_perm [1,0,3,2]: the partner lane's value
#pragma unroll
    for (int j = 0; j < VP; ++j) {
        const int px = pbase + j * 128;
        float v[4] = {xv[j].x, xv[j].y, xv[j].z, xv[j].w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float t = (v[i] - m4[i]) * r4[i];
            v[i] = fmaxf(t, slope * t);
        }
        if constexpr (RES == 2) {
            // even lane holds the 8 heads (dwords x, y = channels 0-3; z, w = 4-7), odd lane the 8 tails: each needs the head AND tail
            // dwords of ITS four channels -- the even lane gives away its heads of channels 4-7, the odd lane its tails of channels 0-3
            const unsigned g0 = swap1(half ? rq[j].x : rq[j].z), g1 = swap1(half ? rq[j].y : rq[j].w);
            const unsigned hd0 = half ? g0 : rq[j].x, hd1 = half ? g1 : rq[j].y;
            const unsigned tl0 = half ? rq[j].z : g0, tl1 = half ? rq[j].w : g1;
            v[0] += __uint_as_float(hd0 << 16) + __uint_as_float(tl0 << 16);
            v[1] += __uint_as_float(hd0 & 0xffff0000u) + __uint_as_float(tl0 & 0xffff0000u);
            v[2] += __uint_as_float(hd1 << 16) + __uint_as_float(tl1 << 16);
            v[3] += __uint_as_float(hd1 & 0xffff0000u) + __uint_as_float(tl1 & 0xffff0000u);
        }
        unsigned hw_[2], tw_[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            __bf16 h0, l0, h1, l1;
            split_bf16(p.xs_relu ? fmaxf(v[2 * i], 0.f) : v[2 * i], h0, l0);
            split_bf16(p.xs_relu ? fmaxf(v[2 * i + 1], 0.f) : v[2 * i + 1], h1, l1);
            hw_[i] = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
            tw_[i] = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
        }
        // the even lane stores the pixel's head slot (its channels 0-3 + the partner's 4-7), the odd lane the tail slot
        const unsigned o0 = swap1(half ? hw_[0] : tw_[0]), o1 = swap1(half ? hw_[1] : tw_[1]);
        if (px < HW) {
            if (!half) ph[px] = make_uint4(hw_[0], hw_[1], o0, o1);
            else if (!p.heads_only) pl[px] = make_uint4(o0, o1, tw_[0], tw_[1]);
        }
    }
}

// ---- row expansion for the K x K stems with <= 4 input channels (Bf3Cfg ROW mode): the split tensor of the
// 32-channel map  R[ky * C + c][y][x] = act(IN(x))[c][y + ky - pad][x]   (vertical zero / reflection padding applied
// here, channels >= K * C zero); the horizontal taps and padding are the 1 x K convolution's.  One lane per pixel;
// every (part, channel group) plane is written as consecutive 16-byte slots.  grid: (ceil(HW / 256), N)
struct SplitRowsParams {
    const float* x;
    const float* mean;
    const float* rstd;
    int act;
    int N, C, H, W, K, pad, pad_mode;
    uint4* out;               // XS[n][part][4][HW + 1]
};

template <int K, int CIN>
__global__ __launch_bounds__(256) void split_rows_kernel(const SplitRowsParams p) {
    static_assert(K * CIN <= 32, "row channels must fit two 16-channel chunks");
    const int n = blockIdx.y, HW = p.H * p.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (blockIdx.x == 0 && threadIdx.x < 8)   // the all-zero slot closing each of the 2 x 4 planes
        p.out[((long long)(n * 2 + (threadIdx.x >> 2)) * 4 + (threadIdx.x & 3)) * (HW + 1) + HW] = make_uint4(0u, 0u, 0u, 0u);
    if (pix >= HW) return;
    const int y = pix / p.W, x = pix - y * p.W;
    float v[32];
#pragma unroll
    for (int i = 0; i < 32; ++i) v[i] = 0.f;
    // branch-free (see split_s2d_kernel): constants, then all K * CIN reads in flight from clamped addresses, then the math
    float m[CIN], rs[CIN];
#pragma unroll
    for (int c = 0; c < CIN; ++c) {
        m[c] = 0.f; rs[c] = 1.f;
        if (p.mean != nullptr) { m[c] = p.mean[n * CIN + c]; rs[c] = p.rstd[n * CIN + c]; }
    }
    const float slope = p.act == 1 ? 0.f : (p.act == 2 ? 0.2f : 1.f);
    bool okr[K];
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        int sy = y + ky - p.pad;
        okr[ky] = true;
        if (p.pad_mode == 1) sy = reflect_clamp(sy, p.H);
        else okr[ky] = sy >= 0 && sy < p.H;
        const float* row = p.x + (long long)n * CIN * HW + (okr[ky] ? sy * p.W + x : 0);
#pragma unroll
        for (int c = 0; c < CIN; ++c) v[ky * CIN + c] = row[(long long)c * HW];
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int c = 0; c < CIN; ++c) {
            float t = (v[ky * CIN + c] - m[c]) * rs[c];
            t = fmaxf(t, slope * t);
            v[ky * CIN + c] = okr[ky] ? t : 0.f;
        }
#pragma unroll
    for (int cg = 0; cg < 4; ++cg) {
        bf16x8 hv, lv;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            __bf16 h, l;
            split_bf16(v[cg * 8 + j], h, l);
            hv[j] = h;
            lv[j] = l;
        }
        *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 0) * 4 + cg) * (HW + 1) + pix) = hv;
        *reinterpret_cast<bf16x8*>(p.out + ((long long)(n * 2 + 1) * 4 + cg) * (HW + 1) + pix) = lv;
    }
}

// ---- space-to-depth split copy for the 4x4 stride-2 pad-1 layers of the PatchGAN (networks.py:2620-2636):
//   X'[(ry*2 + rx)*C + c][qy][qx] = pad1(act(IN(x)))[c][2 qy + ry][2 qx + rx]        (H/2 + 1) x (W/2 + 1), 4C channels
// so that the layer is the stride-1 2 x 2 convolution  y[co][oy][ox] = sum W'[co][c'][ty][tx] X'[c'][oy + ty][ox + tx]
// with W'[co][(ry*2+rx)*C + c][ty][tx] = W[co][c][2 ty + ry][2 tx + rx], which the run-time-tap kernel computes.
struct SplitS2dParams {
    const float* x;
    const float* mean;
    const float* rstd;
    int act;
    int N, C, H, W;           // source
    uint4* out;               // XS[n][part][4C/8][H'W' + 1]
};

// grid: (ceil(H'W' / 256), 4C/8, N)
static __global__ __launch_bounds__(256) void split_s2d_kernel(const SplitS2dParams p) {
    const int n = blockIdx.z, g2 = blockIdx.y, G = p.C >> 3;
    const int r = g2 / G, g = g2 - r * G, ry = r >> 1, rx = r & 1;
    const int H2 = p.H / 2 + 1, W2 = p.W / 2 + 1, HW2 = H2 * W2, HW = p.H * p.W;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    uint4* const hi = p.out + ((long long)(n * 2 + 0) * (4 * G) + g2) * (HW2 + 1);
    uint4* const lo = p.out + ((long long)(n * 2 + 1) * (4 * G) + g2) * (HW2 + 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) hi[HW2] = lo[HW2] = make_uint4(0u, 0u, 0u, 0u);
    if (pix >= HW2) return;
    const int qy = pix / W2, qx = pix - qy * W2;
    const int sy = 2 * qy + ry - 1, sx = 2 * qx + rx - 1;
    bf16x8 hv, lv;
    const bool ok = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
    // branch-free: the norm constants of the 8 channels first, then the 8 plane reads in flight together (from a clamped,
    // always legal address), then the arithmetic.  (Written as a per-channel `if (ok) { load; norm; act }` the compiler
    // emitted one memory round trip per channel -- data, then mean / rstd as vector loads behind it -- and a branch per
    // activation: eight serial round trips per thread, 4.0 TB/s.)
    float m[8], rs[8], t[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        m[j] = 0.f; rs[j] = 1.f;
        if (p.mean != nullptr) { m[j] = p.mean[n * p.C + g * 8 + j]; rs[j] = p.rstd[n * p.C + g * 8 + j]; }
    }
    const float* const x0 = p.x + ((long long)n * p.C + g * 8) * HW + (ok ? sy * p.W + sx : 0);
#pragma unroll
    for (int j = 0; j < 8; ++j) t[j] = x0[(long long)j * HW];
    const float slope = p.act == 1 ? 0.f : (p.act == 2 ? 0.2f : 1.f);        // act(v) = max(v, slope * v)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = (t[j] - m[j]) * rs[j];
        v = fmaxf(v, slope * v);
        v = ok ? v : 0.f;
        __bf16 h, l;
        split_bf16(v, h, l);
        hv[j] = h;
        lv[j] = l;
    }
    *reinterpret_cast<bf16x8*>(hi + pix) = hv;
    *reinterpret_cast<bf16x8*>(lo + pix) = lv;
}

// ---- weight packer: out = LDS image per (cout tile, chunk): [part][tap][kgroup][CO_TILE][8] bf16
// The source is addressed through element strides, so that the operand of a data-gradient operator -- a channel slice, a
// transposed-tap view of the layer's parameter -- or of the derived forms below needs no contiguous temporary:
//   view == 0: the dense OIHW / IOHW tensor described by layout / Cin / Cout / K (strides derived here);
//   view == 1: element (co, cin, ky, kx) of the OPERATOR at  w[co * s_co + cin * s_ci + ky * s_ky + kx * s_kx];
//   s2d_c  > 0: the operator is the 2 x 2 space-to-depth form over 4 * s2d_c channels of a ksrc x ksrc stride-2 layer:
//               operator (cin = r * s2d_c + c, tap (ty, tx)) = source (c, 2 ty + (r >> 1), 2 tx + (r & 1)), zero beyond ksrc;
//   rows_c > 0: the operator is the 1 x K row form of a K x K stem over rows_c channels: operator cin = ky * rows_c + c.
struct PackBf3Params {
    const float* w;
    unsigned short* out;
    int Cin, Cout, K, layout, flip;
    int KH;                                             // 0: square K x K weights; 1: 1 x K
    int nseg, segC[kMaxSeg], chunk_begin[kMaxSeg];
    int CO_TILE, nchunks, co_tiles;
    int ntaps, tap_ky[kMaxTaps], tap_kx[kMaxTaps];      // source tap of packed tap t (already flipped if needed)
    int view, s2d_c, rows_c, ksrc;
    long long s_co, s_ci, s_ky, s_kx;
};

// One 16-byte slot (8 input channels of one cout, tap and k-group) per thread and step, head and tail images together:
// the source value is read once for both parts and leaves as two 16-byte stores (a thread per bf16 element measured 424 us
// for the generator's table, 0.45 TB/s).
__device__ __forceinline__ void pack_bf16x3_body(const PackBf3Params& p, long long first, long long step) {
    const int T = p.ntaps;
    const unsigned slots_blk = (unsigned)T * 2u * (unsigned)p.CO_TILE;          // slots of one part of a (cout tile, chunk) block
    const long long total = (long long)p.co_tiles * p.nchunks * slots_blk;
    const int KH = p.KH > 0 ? p.KH : p.K;                                       // KH != K: a 1 x K row kernel
    for (long long sidx = first; sidx < total; sidx += step) {
        const unsigned blk = (unsigned)(sidx / slots_blk);
        unsigned r = (unsigned)(sidx - (long long)blk * slots_blk);
        const int chunk = (int)(blk % (unsigned)p.nchunks), cot = (int)(blk / (unsigned)p.nchunks);
        const int col = (int)(r % (unsigned)p.CO_TILE); r /= (unsigned)p.CO_TILE;
        const int kg = (int)(r & 1u);
        const int t = (int)(r >> 1);
        const int co = cot * p.CO_TILE + col;
        int s = 0;
        if (p.nseg > 1 && chunk >= p.chunk_begin[1]) s = 1;
        if (p.nseg > 2 && chunk >= p.chunk_begin[2]) s = 2;
        int seg0 = 0;
        for (int j = 0; j < s; ++j) seg0 += p.segC[j];
        const int cs0 = (chunk - p.chunk_begin[s]) * 16 + kg * 8;
        const bool row_ok = co < p.Cout && p.tap_ky[t] >= 0;       // tap_ky < 0: a window position this phase does not have
        bf16x8 hv, lv;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const int cs = cs0 + c;
            float v = 0.f;
            if (row_ok && cs < p.segC[s]) {
                int cin = cs + seg0;
                int ky = p.tap_ky[t], kx = p.tap_kx[t];
                bool ok = true;
                if (p.s2d_c > 0) {
                    const int rr = cin / p.s2d_c;
                    cin -= rr * p.s2d_c;
                    ky = 2 * ky + (rr >> 1);
                    kx = 2 * kx + (rr & 1);
                    ok = ky < p.ksrc && kx < p.ksrc;
                } else if (p.rows_c > 0) {
                    ky = cin / p.rows_c;
                    cin -= ky * p.rows_c;
                    ok = ky < p.ksrc;
                }
                if (ok) {
                    long long off;
                    if (p.view) {
                        off = co * p.s_co + cin * p.s_ci + ky * p.s_ky + kx * p.s_kx;
                    } else {
                        off = p.layout == 0 ? (((long long)co * p.Cin + cin) * KH + ky) * p.K + kx
                                            : (((long long)cin * p.Cout + co) * KH + ky) * p.K + kx;
                    }
                    v = p.w[off];
                }
            }
            __bf16 h, l;
            split_bf16(v, h, l);
            hv[c] = h;
            lv[c] = l;
        }
        // image of the block: [part][tap][kgroup][CO_TILE] slots
        bf16x8* const img = reinterpret_cast<bf16x8*>(p.out) + (long long)blk * 2 * slots_blk;
        const unsigned slot = ((unsigned)t * 2u + (unsigned)kg) * (unsigned)p.CO_TILE + (unsigned)col;
        img[slot] = hv;
        img[slots_blk + slot] = lv;
    }
}

static __global__ void pack_bf16x3_kernel(const PackBf3Params p) {
    pack_bf16x3_body(p, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

// every (layer, variant) of a network in ONE launch: entry blockIdx.y of a device-resident table (built once: the parameters
// are views of the optimiser's flat buffer and the packed images are persistent, so the pointers do not change)
static __global__ void pack_bf16x3_table_kernel(const PackBf3Params* __restrict__ table) {
    __shared__ PackBf3Params p;
    const int* src = reinterpret_cast<const int*>(table + blockIdx.y);
    int* dst = reinterpret_cast<int*>(&p);
    for (int i = threadIdx.x; i < (int)(sizeof(PackBf3Params) / sizeof(int)); i += blockDim.x) dst[i] = src[i];
    __syncthreads();
    pack_bf16x3_body(p, (long long)blockIdx.x * blockDim.x + threadIdx.x, (long long)gridDim.x * blockDim.x);
}

}  // namespace apamd
