// backward_elementwise.hip -- the HBM-bound pieces of the backward pass.
//
// What autograd would launch for the reference's layers (Module2/models/networks.py:1218-1282,
// 2329-2421, 2620-2643) as separate ATen kernels is fused here into four streaming kernels:
//   * instnorm_bwd_reduce / instnorm_bwd_apply: backward of  a = act(IN(y))  w.r.t. y, where the
//     incoming gradient may be (i) the data-gradient of a reflection-padded convolution, still in
//     PADDED coordinates (the fold of nn.ReflectionPad2d's backward is done while reading), plus
//     (ii) a second plain gradient (residual branch);
//   * act_bwd: backward of a plain activation (LeakyReLU / tanh) for layers without normalisation;
//   * bias_grad: per-channel sum of a gradient (layers whose bias is live: no InstanceNorm after them);
//   * grad_fold_add: out = fold(a) + b  (residual stream accumulation).
#include "common.h"
#include <type_traits>

#include <cstdlib>
#include <cstring>

namespace apamd {

// gradient read with optional reflection fold: g has spatial (H+2p) x (W+2p)
struct FoldReader {
    const float* g;   // plane base
    int H, W, p, Wp;
    __device__ __forceinline__ float at(int y, int x) const {
        if (p == 0) return g[y * W + x];
        int ys[3], xs[3], ny = 0, nx = 0;
        ys[ny++] = y + p;
        if (y >= 1 && y <= p) ys[ny++] = p - y;
        if (y >= H - 1 - p && y <= H - 2) ys[ny++] = 2 * (H - 1) - y + p;
        xs[nx++] = x + p;
        if (x >= 1 && x <= p) xs[nx++] = p - x;
        if (x >= W - 1 - p && x <= W - 2) xs[nx++] = 2 * (W - 1) - x + p;
        float s = 0.f;
        for (int i = 0; i < ny; ++i)
            for (int j = 0; j < nx; ++j) s += g[ys[i] * Wp + xs[j]];
        return s;
    }
};

// Four consecutive pixels (x4 .. x4 + 3, x4 % 4 == 0, W % 4 == 0, H >= 3) of the fold of a pad-1 reflection-padded
// gradient plane gp[(H + 2) x (W + 2)]: one 16-byte load of the padded row (4-byte aligned: the hardware takes it)
// plus, on the first / last group of a row and on rows 1 and H - 2, the reflected border terms.
typedef float float4u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ float4 fold1_row4(const float* __restrict__ gp, int W, int py, int x4) {
    const float* rp = gp + py * (W + 2);
    const float4u t = *reinterpret_cast<const float4u*>(rp + x4 + 1);
    float4 v = make_float4(t.x, t.y, t.z, t.w);
    if (x4 == 0) v.y += rp[0];                       // padded column 0 is the reflection of column 1
    if (x4 == W - 4) v.z += rp[W + 1];               // padded column W + 1 is the reflection of column W - 2
    return v;
}

// ... on bf16 gradients (2-byte elements, any 2-byte alignment: the padded rows start at odd elements)
__device__ __forceinline__ float bf16_val(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
// four bf16 at an 8-byte-aligned address: one 8-byte load
__device__ __forceinline__ float4 ld4_bf16(const unsigned short* __restrict__ q) {
    const uint2 t = *reinterpret_cast<const uint2*>(q);
    return make_float4(__uint_as_float(t.x << 16), __uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u));
}
// four bf16 at an ODD element offset of a 4-byte-aligned row (the interior of a pad-1 row: element x + 1, x % 4 == 0, even row
// pitch): ONE 12-byte load of the three dwords around them -- q[-1] and q[4] lie inside the same padded row.  (As four 2-byte
// loads -- what a 2-byte-aligned vector type compiles to -- this read made instnorm_bwd_split<.., G16> half as fast as its
// fp32 form: 110-117 us against 56 for half the bytes.)
__device__ __forceinline__ float4 ld4_bf16_odd(const unsigned short* __restrict__ q) {
    typedef unsigned u32x3 __attribute__((ext_vector_type(3), aligned(4)));
    const u32x3 t = *reinterpret_cast<const u32x3*>(q - 1);
    return make_float4(__uint_as_float(t.x & 0xffff0000u), __uint_as_float(t.y << 16), __uint_as_float(t.y & 0xffff0000u), __uint_as_float(t.z << 16));
}
__device__ __forceinline__ float4 fold1_row4_bf16(const unsigned short* __restrict__ gp, int W, int py, int x4) {
    const unsigned short* rp = gp + py * (W + 2);
    float4 v = ld4_bf16_odd(rp + x4 + 1);
    if (x4 == 0) v.y += bf16_val(rp[0]);
    if (x4 == W - 4) v.z += bf16_val(rp[W + 1]);
    return v;
}

__device__ __forceinline__ float4 fold1_at4(const float* __restrict__ gp, int H, int W, int y, int x4) {
    float4 a = fold1_row4(gp, W, y + 1, x4);
    if (y == 1) { const float4 t = fold1_row4(gp, W, 0, x4); a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    if (y == H - 2) { const float4 t = fold1_row4(gp, W, H + 1, x4); a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    return a;
}

// The same for pad p = 2, 3 (the 7x7 layers: p = 3) with W % 4 == 0, W >= 4 + 2 p, H >= 2 p + 2: only the first and the last
// group of a row have reflected column partners (pixel x in [1, p] <- padded column p - x; x in [W - 1 - p, W - 2] <- padded
// column 2 (W - 1) - x + p), and only rows [1, p] / [H - 1 - p, H - 2] a partner row: no per-pixel loops (FoldReader::at walks
// up to 3 x 3 taps per pixel with data-dependent trip counts -- one memory round trip each).
__device__ __forceinline__ float4 foldp_row4(const float* __restrict__ gp, int W, int p, int py, int x4) {
    const float* rp = gp + py * (W + 2 * p);
    const float4u t = *reinterpret_cast<const float4u*>(rp + x4 + p);
    float4 v = make_float4(t.x, t.y, t.z, t.w);
    if (x4 == 0) {                                   // pixels 1 .. 3 <- padded columns p - 1, p - 2, p - 3
        v.y += rp[p - 1];
        if (p >= 2) v.z += rp[p - 2];
        if (p >= 3) v.w += rp[p - 3];
    }
    if (x4 == W - 4) {                               // pixels W - 4 .. W - 2 <- padded columns W + p + 2, W + p + 1, W + p
        v.z += rp[W + p];
        if (p >= 2) v.y += rp[W + p + 1];
        if (p >= 3) v.x += rp[W + p + 2];
    }
    return v;
}

__device__ __forceinline__ float4 foldp_at4(const float* __restrict__ gp, int H, int W, int p, int y, int x4) {
    float4 a = foldp_row4(gp, W, p, y + p, x4);
    int py2 = -1;
    if (y >= 1 && y <= p) py2 = p - y;
    else if (y >= H - 1 - p && y <= H - 2) py2 = 2 * (H - 1) - y + p;
    if (py2 >= 0) { const float4 t = foldp_row4(gp, W, p, py2, x4); a.x += t.x; a.y += t.y; a.z += t.z; a.w += t.w; }
    return a;
}

__device__ __forceinline__ float act_grad_from_xhat(float xh, int act) {
    if (act == 1) return xh > 0.f ? 1.f : 0.f;
    if (act == 2) return xh > 0.f ? 1.f : 0.2f;
    return 1.f;
}

__device__ __forceinline__ float block_sum(float v, float* red) {
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) v += __shfl_xor(v, sh, 64);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[w] = v;
    __syncthreads();
    float s = 0.f;
    for (int i = 0; i < (int)(blockDim.x >> 6); ++i) s += red[i];
    __syncthreads();
    return s;
}

// grid: (N*C).  sums[nc] = (sum g', sum g' * xhat), g' = (fold(g1) + g2) * act'(xhat)
__global__ __launch_bounds__(256) void instnorm_bwd_reduce_kernel(const float* __restrict__ g1, int p1,
                                                                  const float* __restrict__ g2,
                                                                  const float* __restrict__ y,
                                                                  const float* __restrict__ mean,
                                                                  const float* __restrict__ rstd, int act, int H, int W,
                                                                  float* __restrict__ sums) {
    __shared__ float red[8];
    const int nc = blockIdx.x;
    const int HW = H * W;
    const float m = mean[nc], r = rstd[nc];
    FoldReader fr{g1 + (long long)nc * (H + 2 * p1) * (W + 2 * p1), H, W, p1, W + 2 * p1};
    const float* yp = y + (long long)nc * HW;
    const float* g2p = g2 ? g2 + (long long)nc * HW : nullptr;
    float s1 = 0.f, s2 = 0.f;
    if (p1 == 0 && (HW & 3) == 0) {
        // unfolded gradient: 16-byte lanes (same per-thread summation order is not required: block_sum is a tree)
        const float4* y4 = reinterpret_cast<const float4*>(yp);
        const float4* ga = reinterpret_cast<const float4*>(g1 + (long long)nc * HW);
        const float4* gb = g2p ? reinterpret_cast<const float4*>(g2p) : nullptr;
        for (int i = threadIdx.x; i < HW / 4; i += 256) {
            const float4 yv = y4[i];
            float4 gv = ga[i];
            if (gb) { const float4 t = gb[i]; gv.x += t.x; gv.y += t.y; gv.z += t.z; gv.w += t.w; }
            const float xh0 = (yv.x - m) * r, xh1 = (yv.y - m) * r, xh2 = (yv.z - m) * r, xh3 = (yv.w - m) * r;
            const float a0 = gv.x * act_grad_from_xhat(xh0, act), a1 = gv.y * act_grad_from_xhat(xh1, act);
            const float a2 = gv.z * act_grad_from_xhat(xh2, act), a3 = gv.w * act_grad_from_xhat(xh3, act);
            s1 += (a0 + a1) + (a2 + a3);
            s2 += (a0 * xh0 + a1 * xh1) + (a2 * xh2 + a3 * xh3);
        }
    } else
    for (int i = threadIdx.x; i < HW; i += 256) {
        const int yy = i / W, xx = i - yy * W;
        const float xh = (yp[i] - m) * r;
        float g = fr.at(yy, xx);
        if (g2p) g += g2p[i];
        g *= act_grad_from_xhat(xh, act);
        s1 += g;
        s2 += g * xh;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    if (threadIdx.x == 0) {
        sums[nc * 2] = s1;
        sums[nc * 2 + 1] = s2;
    }
}

// grid: (ceil(HW/256/4), N*C).  dy = rstd * (g' - S1/HW - xhat * S2/HW)
__global__ __launch_bounds__(256) void instnorm_bwd_apply_kernel(const float* __restrict__ g1, int p1,
                                                                 const float* __restrict__ g2,
                                                                 const float* __restrict__ y,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ rstd, int act, int H, int W,
                                                                 const float* __restrict__ sums,
                                                                 float* __restrict__ dy) {
    const int nc = blockIdx.y;
    const int HW = H * W;
    const float m = mean[nc], r = rstd[nc];
    const float inv = 1.f / (float)HW;
    const float a1 = sums[nc * 2] * inv, a2 = sums[nc * 2 + 1] * inv;
    FoldReader fr{g1 + (long long)nc * (H + 2 * p1) * (W + 2 * p1), H, W, p1, W + 2 * p1};
    const float* yp = y + (long long)nc * HW;
    const float* g2p = g2 ? g2 + (long long)nc * HW : nullptr;
    float* out = dy + (long long)nc * HW;
    if (p1 == 0 && (HW & 3) == 0) {
        const float4* y4 = reinterpret_cast<const float4*>(yp);
        const float4* ga = reinterpret_cast<const float4*>(g1 + (long long)nc * HW);
        const float4* gb = g2p ? reinterpret_cast<const float4*>(g2p) : nullptr;
        float4* o4 = reinterpret_cast<float4*>(out);
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
            const float4 yv = y4[i];
            float4 gv = ga[i];
            if (gb) { const float4 t = gb[i]; gv.x += t.x; gv.y += t.y; gv.z += t.z; gv.w += t.w; }
            float4 o;
            float xh = (yv.x - m) * r; o.x = r * (gv.x * act_grad_from_xhat(xh, act) - a1 - xh * a2);
            xh = (yv.y - m) * r; o.y = r * (gv.y * act_grad_from_xhat(xh, act) - a1 - xh * a2);
            xh = (yv.z - m) * r; o.z = r * (gv.z * act_grad_from_xhat(xh, act) - a1 - xh * a2);
            xh = (yv.w - m) * r; o.w = r * (gv.w * act_grad_from_xhat(xh, act) - a1 - xh * a2);
            o4[i] = o;
        }
        return;
    }
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW; i += gridDim.x * 256) {
        const int yy = i / W, xx = i - yy * W;
        const float xh = (yp[i] - m) * r;
        float g = fr.at(yy, xx);
        if (g2p) g += g2p[i];
        g *= act_grad_from_xhat(xh, act);
        out[i] = r * (g - a1 - xh * a2);
    }
}

// One pass for planes that fit the registers of a workgroup (H*W <= NT * EPT): the gradient and the normalised
// activation of one (n, c) plane are loaded once, reduced in the block, and dy is written from the registers --
// 2 reads + 1 write per element instead of the 4 + 1 of the reduce / apply pair.  grid: (N*C)
template <int NT, int EPT>
__global__ __launch_bounds__(NT) void instnorm_bwd_fused_kernel(const float* __restrict__ g1, int p1,
                                                                const float* __restrict__ g2,
                                                                const float* __restrict__ y,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ rstd, int act, int H, int W,
                                                                float* __restrict__ dy) {
    __shared__ float red[NT / 64];
    const int nc = blockIdx.x, tid = threadIdx.x;
    const int HW = H * W;
    const float m = mean[nc], r = rstd[nc];
    const float* yp = y + (long long)nc * HW;
    const float* g2p = g2 ? g2 + (long long)nc * HW : nullptr;
    float* out = dy + (long long)nc * HW;
    float gv[EPT], xh[EPT];
    float s1 = 0.f, s2 = 0.f;
    const bool fold1 = p1 == 1 && (W & 3) == 0 && H >= 3;
    const bool vec = (p1 == 0 && (HW & 3) == 0) || fold1;
    if (vec) {
        const float4* y4 = reinterpret_cast<const float4*>(yp);
        const float4* ga = reinterpret_cast<const float4*>(g1 + (long long)nc * HW);
        const float* gp = g1 + (long long)nc * (H + 2) * (W + 2);
        const float4* gb = g2p ? reinterpret_cast<const float4*>(g2p) : nullptr;
        const int W4 = W >> 2;
#pragma unroll
        for (int k = 0; k < EPT / 4; ++k) {
            const int i = k * NT + tid;
            float4 yv = make_float4(m, m, m, m), gq = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < HW / 4) {
                yv = y4[i];
                if (fold1) { const int yy = i / W4; gq = fold1_at4(gp, H, W, yy, (i - yy * W4) * 4); }
                else gq = ga[i];
                if (gb) { const float4 t = gb[i]; gq.x += t.x; gq.y += t.y; gq.z += t.z; gq.w += t.w; }
            }
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, gg[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = (yy[j] - m) * r;
                const float g = gg[j] * act_grad_from_xhat(x, act);
                xh[k * 4 + j] = x;
                gv[k * 4 + j] = g;
                s1 += g;
                s2 += g * x;
            }
        }
    } else {
        FoldReader fr{g1 + (long long)nc * (H + 2 * p1) * (W + 2 * p1), H, W, p1, W + 2 * p1};
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = k * NT + tid;
            float x = 0.f, g = 0.f;
            if (i < HW) {
                const int yy = i / W, xx = i - yy * W;
                x = (yp[i] - m) * r;
                g = fr.at(yy, xx);
                if (g2p) g += g2p[i];
                g *= act_grad_from_xhat(x, act);
            }
            xh[k] = x;
            gv[k] = g;
            s1 += g;
            s2 += g * x;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float inv = 1.f / (float)HW;
    const float a1 = s1 * inv, a2 = s2 * inv;
    if (vec) {
        float4* o4 = reinterpret_cast<float4*>(out);
#pragma unroll
        for (int k = 0; k < EPT / 4; ++k) {
            const int i = k * NT + tid;
            if (i < HW / 4)
                o4[i] = make_float4(r * (gv[k * 4] - a1 - xh[k * 4] * a2), r * (gv[k * 4 + 1] - a1 - xh[k * 4 + 1] * a2),
                                    r * (gv[k * 4 + 2] - a1 - xh[k * 4 + 2] * a2), r * (gv[k * 4 + 3] - a1 - xh[k * 4 + 3] * a2));
        }
    } else {
#pragma unroll
        for (int k = 0; k < EPT; ++k) {
            const int i = k * NT + tid;
            if (i < HW) out[i] = r * (gv[k] - a1 - xh[k] * a2);
        }
    }
}

// The same for an unfolded gradient with H W % 4 == 0 (every launch of the train step that is not on the operand-writing route),
// with the load-phase rule applied (section 3.5 of DESIGN.md): all 16-byte loads of a thread -- y, g1 and, as a template parameter,
// g2 -- are issued from clamped indices before any arithmetic.  (In the general kernel above the `i < HW / 4` test, the fold test and
// the `g2 != nullptr` test sit between the loads: 8-12 dependent round trips per thread, 3.9 TB/s.)
template <int NT, int EPT, bool G2>
__global__ __launch_bounds__(NT) void instnorm_bwd_fused_vec_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                                    const float* __restrict__ y, const float* __restrict__ mean,
                                                                    const float* __restrict__ rstd, int act, int HW,
                                                                    float* __restrict__ dy) {
    __shared__ float red[NT / 64];
    constexpr int NG = EPT / 4;
    const int nc = blockIdx.x, tid = threadIdx.x, Q = HW >> 2;
    const float4* y4 = reinterpret_cast<const float4*>(y + (long long)nc * HW);
    const float4* ga = reinterpret_cast<const float4*>(g1 + (long long)nc * HW);
    const float4* gb = reinterpret_cast<const float4*>((G2 ? g2 : g1) + (long long)nc * HW);
    float4 yv[NG], gq[NG], gr[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int i = k * NT + tid, ic = i < Q ? i : Q - 1;
        yv[k] = y4[ic];
        gq[k] = ga[ic];
        if constexpr (G2) gr[k] = gb[ic];
    }
    const float m = mean[nc], r = rstd[nc];
    float gv[EPT], xh[EPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const bool in = k * NT + tid < Q;
        const float yy[4] = {yv[k].x, yv[k].y, yv[k].z, yv[k].w};
        float gg[4] = {gq[k].x, gq[k].y, gq[k].z, gq[k].w};
        if constexpr (G2) { gg[0] += gr[k].x; gg[1] += gr[k].y; gg[2] += gr[k].z; gg[3] += gr[k].w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = in ? (yy[j] - m) * r : 0.f;
            const float g = in ? gg[j] * act_grad_from_xhat(x, act) : 0.f;
            xh[k * 4 + j] = x;
            gv[k * 4 + j] = g;
            s1 += g;
            s2 += g * x;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float inv = 1.f / (float)HW;
    const float a1 = s1 * inv, a2 = s2 * inv;
    float4* o4 = reinterpret_cast<float4*>(dy + (long long)nc * HW);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int i = k * NT + tid;
        if (i < Q)
            o4[i] = make_float4(r * (gv[k * 4] - a1 - xh[k * 4] * a2), r * (gv[k * 4 + 1] - a1 - xh[k * 4 + 1] * a2),
                                r * (gv[k * 4 + 2] - a1 - xh[k * 4 + 2] * a2), r * (gv[k * 4 + 3] - a1 - xh[k * 4 + 3] * a2));
    }
}

// ... and for the gradient of a pad-1 reflection-padded consumer (g1: (H + 2) x (W + 2) planes, W % 4 == 0, H >= 3): the padded row's
// 16 bytes and ONE edge dword (padded column 0 or W + 1, whichever this group could need) are loaded unconditionally with y and g2; the
// reflected ROWS 0 and H + 1 belong to the groups of rows 1 and H - 2 only and are added in a second phase that a wave enters only if
// it holds such a group (as instnorm_bwd_split_kernel does).
template <int NT, int EPT, bool G2>
__global__ __launch_bounds__(NT) void instnorm_bwd_fused_fold1_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                                      const float* __restrict__ y, const float* __restrict__ mean,
                                                                      const float* __restrict__ rstd, int act, int H, int W,
                                                                      float* __restrict__ dy) {
    __shared__ float red[NT / 64];
    constexpr int NG = EPT / 4;
    const int nc = blockIdx.x, tid = threadIdx.x, HW = H * W, Q = HW >> 2, W4 = W >> 2;
    const float4* y4 = reinterpret_cast<const float4*>(y + (long long)nc * HW);
    const float* gp = g1 + (long long)nc * (H + 2) * (W + 2);
    const float4* gb = reinterpret_cast<const float4*>((G2 ? g2 : y) + (long long)nc * HW);
    float4 yv[NG], gr[NG];
    float4u gm[NG];
    float ge[NG];
    int gy[NG], gx[NG];
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int i = k * NT + tid, ic = i < Q ? i : Q - 1;
        gy[k] = ic / W4;
        gx[k] = (ic - gy[k] * W4) * 4;
        const float* rp = gp + (gy[k] + 1) * (W + 2);
        gm[k] = *reinterpret_cast<const float4u*>(rp + gx[k] + 1);
        ge[k] = rp[gx[k] == 0 ? 0 : W + 1];
        yv[k] = y4[ic];
        if constexpr (G2) gr[k] = gb[ic];
    }
    const float m = mean[nc], r = rstd[nc];
    float4 gq[NG];
    bool border = false;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        gq[k] = make_float4(gm[k].x, gm[k].y, gm[k].z, gm[k].w);
        if (gx[k] == 0) gq[k].y += ge[k];                 // padded column 0 is the reflection of column 1
        if (gx[k] == W - 4) gq[k].z += ge[k];             // padded column W + 1 is the reflection of column W - 2
        border = border || gy[k] == 1 || gy[k] == H - 2;
    }
    if (__builtin_amdgcn_ballot_w64(border) != 0) {       // wave-uniform: this wave holds groups of row 1 or row H - 2
#pragma unroll
        for (int k = 0; k < NG; ++k) {
            if (gy[k] == 1) { const float4 t = fold1_row4(gp, W, 0, gx[k]); gq[k].x += t.x; gq[k].y += t.y; gq[k].z += t.z; gq[k].w += t.w; }
            if (gy[k] == H - 2) { const float4 t = fold1_row4(gp, W, H + 1, gx[k]); gq[k].x += t.x; gq[k].y += t.y; gq[k].z += t.z; gq[k].w += t.w; }
        }
    }
    float gv[EPT], xh[EPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const bool in = k * NT + tid < Q;
        const float yy[4] = {yv[k].x, yv[k].y, yv[k].z, yv[k].w};
        float gg[4] = {gq[k].x, gq[k].y, gq[k].z, gq[k].w};
        if constexpr (G2) { gg[0] += gr[k].x; gg[1] += gr[k].y; gg[2] += gr[k].z; gg[3] += gr[k].w; }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float x = in ? (yy[j] - m) * r : 0.f;
            const float g = in ? gg[j] * act_grad_from_xhat(x, act) : 0.f;
            xh[k * 4 + j] = x;
            gv[k * 4 + j] = g;
            s1 += g;
            s2 += g * x;
        }
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float inv = 1.f / (float)HW;
    const float a1 = s1 * inv, a2 = s2 * inv;
    float4* o4 = reinterpret_cast<float4*>(dy + (long long)nc * HW);
#pragma unroll
    for (int k = 0; k < NG; ++k) {
        const int i = k * NT + tid;
        if (i < Q)
            o4[i] = make_float4(r * (gv[k * 4] - a1 - xh[k * 4] * a2), r * (gv[k * 4 + 1] - a1 - xh[k * 4 + 1] * a2),
                                r * (gv[k * 4 + 2] - a1 - xh[k * 4 + 2] * a2), r * (gv[k * 4 + 3] - a1 - xh[k * 4 + 3] * a2));
    }
}

// ... and for small planes of any size with an unfolded gradient (the PatchGAN's 31 x 31 maps: H W = 961 is odd, so the general
// kernel took its element-wise path -- 16 predicated iterations with the loads behind their tests): four elements per thread, every
// load from a clamped index first.
template <bool G2>
__global__ __launch_bounds__(256) void instnorm_bwd_fused_small_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                                       const float* __restrict__ y, const float* __restrict__ mean,
                                                                       const float* __restrict__ rstd, int act, int HW,
                                                                       float* __restrict__ dy) {
    __shared__ float red[4];
    constexpr int EPT = 4;
    const int nc = blockIdx.x, tid = threadIdx.x;
    const float* yp = y + (long long)nc * HW;
    const float* ga = g1 + (long long)nc * HW;
    const float* gb = (G2 ? g2 : g1) + (long long)nc * HW;
    float yv[EPT], gq[EPT], gr[EPT];
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = k * 256 + tid, ic = i < HW ? i : HW - 1;
        yv[k] = yp[ic];
        gq[k] = ga[ic];
        if constexpr (G2) gr[k] = gb[ic];
    }
    const float m = mean[nc], r = rstd[nc];
    float gv[EPT], xh[EPT];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const bool in = k * 256 + tid < HW;
        const float x = in ? (yv[k] - m) * r : 0.f;
        float g = gq[k];
        if constexpr (G2) g += gr[k];
        g = in ? g * act_grad_from_xhat(x, act) : 0.f;
        xh[k] = x;
        gv[k] = g;
        s1 += g;
        s2 += g * x;
    }
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float inv = 1.f / (float)HW;
    const float a1 = s1 * inv, a2 = s2 * inv;
    float* out = dy + (long long)nc * HW;
#pragma unroll
    for (int k = 0; k < EPT; ++k) {
        const int i = k * 256 + tid;
        if (i < HW) out[i] = r * (gv[k] - a1 - xh[k] * a2);
    }
}

// ---- instnorm_bwd_split: the InstanceNorm backward of a layer whose gradient goes straight into the bf16 matrix kernels.
// The fp32 gradient dy of such a layer was written once (4 B / element) and read twice -- by the split pass in front of the
// data-gradient convolution and by the transposition in front of the weight gradient -- to be rounded to bf16 head (+ tail)
// planes both times.  Here a workgroup owns the EIGHT channels of a channel octet of one image (a thread: 4 consecutive pixels
// of each), so after the plane reductions it holds whole slots of both operand layouts and writes them itself:
//   xs  [n][part][C / 8][HW + 1][8 channels]      the split copy conv_bf16x3 stages (conv_bf16x3.h), zero slot included;
//   gt  [n][part][GHp * GX8][Mp][8 pixels]        the M-role operand of wgrad_bf16x3 (wgrad_bf16x3.h): a lane pair holds the two
//                                                 quads of a pixel octet and swaps channel halves (8 shuffles per part);
//   strip [n][c][2][H]  fp32                      columns W - 2, W - 1 transposed: the operand of ops.conv2d_dgrad_strip.
// dy itself is written only on request.  2 reads + (1/2 + 1/2 | 1 + 1) writes per element instead of 4 + (2 | 3).
// grid: min(N * C / 8, resident workgroups), persistent; NT threads >= H * W / 4; needs W % 8 == 0, H >= 3.
struct InBwdSplitParams {
    const float* g1;
    const float* g2;
    const float* y;
    const float* mean;
    const float* rstd;
    int p1, act, N, C, H, W;
    uint4* xs;
    uint4* gt;
    int GHp, GX8, Mp;
    float* strip;
    float* dy;
    int heads_only;
};

__device__ __forceinline__ unsigned pack_bf16x2(float a, float b) {
    const __bf16 ha = (__bf16)a, hb = (__bf16)b;
    return (unsigned)__builtin_bit_cast(unsigned short, ha) | ((unsigned)__builtin_bit_cast(unsigned short, hb) << 16);
}
__device__ __forceinline__ float bf16_tail(float v) { return v - (float)(__bf16)v; }

// YB16: y holds bf16 values (the raw output ap_conv2d_fwd_bf16out stored); G16: g1 holds bf16 values (a data gradient stored by
// ap_conv2d_fwd_view_bf16out) -- same element offsets, 2-byte elements
// FOLD: g1 is the gradient of a reflection-padded (pad 1) layer, still in padded coordinates (p.p1 == 1)
template <int NT, bool YB16 = false, bool G16 = false, bool FOLD = false>
__global__ __launch_bounds__(NT) void instnorm_bwd_split_kernel(const InBwdSplitParams p) {
    constexpr int NW = NT / 64;
    __shared__ float red[16][NW];
    __shared__ float tot[16];
    const int tid0 = threadIdx.x;
    const int H = p.H, W = p.W, HW = H * W, W4 = W >> 2;
    const bool live = tid0 < HW / 4;
    const int CG = p.C >> 3, items = p.N * CG;
    // A workgroup holds one (image, channel octet) in registers -- a CU has room for one such workgroup -- so it walks several of
    // them: the loads of the next item are issued right behind the stores of the current one (which are not waited for), so reads
    // and writes of a CU overlap and there is no drain + dispatch gap between items (first version, one item per workgroup:
    // 2.7 TB/s, profiles/r04x_train_bf16_bygrid.md; an explicit prefetch in front of the stores spilled 150-300 registers).
    // Also tried: the slots staged through LDS so that every store instruction writes whole lines (as norm_split_kernel does for
    // its 64-byte lane stride) -- 84.6 -> 95.2 us per launch (gpurun_out/r04ag_call.log): with ONE workgroup per CU the two extra
    // barriers and 130 KB of LDS traffic cost more than the four-piece line writes, which the L2 merges anyway.
    for (int item = blockIdx.x; item < items; item += (int)gridDim.x) {
    const int n = item / CG, cg = item - n * CG, c0 = cg * 8;
    // (per-lane addresses are recomputed in every iteration from an opaque copy of the lane id: hoisted out of the loop they
    // occupied registers across it)
    int tid = tid0;
    asm volatile("" : "+v"(tid));
    const int row = tid / W4, x4 = (tid - row * W4) * 4;
    float xh[32], gv[32];
    float rvl;                   // lane c (< 8) holds rstd of channel c0 + c
    {
        // ---- load phase (round 6): NOTHING but loads -- raw words into registers, addresses clamped instead of branched on, the
        // fold's border terms taken from a 6-element window of the padded row (its two outer elements ARE the reflected columns of
        // the first / last group) and, in the two waves that hold rows 1 and H - 2, from a second window of padded row 0 / H + 1.
        // Before, every channel's loads were followed by their conversion (bf16 -> fp32 shifts) or sat in the arms of the
        // run-time `fold1` branch, and the compiler waited for each channel before it asked for the next: 16 serial memory round
        // trips per item plus one per border term, 2.8-3.9 TB/s (profiles/r06_stream_readers.md).
        typedef unsigned u32x3 __attribute__((ext_vector_type(3), aligned(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        using YRaw = std::conditional_t<YB16, u32x2, float4>;
        using GRaw = std::conditional_t<G16, std::conditional_t<FOLD, u32x3, u32x2>, std::conditional_t<FOLD, float4u, float4>>;
        const int ltid = live ? tid : 0;
        const int lrow = ltid / W4, lx4 = (ltid - lrow * W4) * 4;
        const long long nc0 = (long long)n * p.C + c0;
        const int PW = W + 2, PHW = (H + 2) * PW;
        // lanes of rows 1 / H - 2 fold padded row 0 / H + 1 in; everybody else reads its own row again (a cache hit), masked out later
        const bool brow = FOLD && (lrow == 1 || lrow == H - 2);
        const bool wave_brow = FOLD && __builtin_amdgcn_readfirstlane((int)(__ballot(brow && live) != 0ull));
        const int prow = lrow == 1 ? 0 : (lrow == H - 2 ? H + 1 : lrow + 1);
        const int ecol = lx4 == 0 ? 0 : (lx4 == W - 4 ? W + 1 : lx4 + 1);          // fp32 fold: the reflected column of the group, if any
        YRaw yr[8];
        GRaw gr[8], br[8];
        float ge[8], be[8];
        const float mvl = p.mean[nc0 + (tid & 7)];
        rvl = p.rstd[nc0 + (tid & 7)];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (YB16) yr[c] = reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p.y) + (nc0 + c) * HW)[ltid];
            else yr[c] = reinterpret_cast<const float4*>(p.y + (nc0 + c) * HW)[ltid];
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            if constexpr (G16 && FOLD) {
                gr[c] = *reinterpret_cast<const u32x3*>(reinterpret_cast<const unsigned short*>(p.g1) + (nc0 + c) * PHW + (lrow + 1) * PW + lx4);
            } else if constexpr (G16) {
                gr[c] = reinterpret_cast<const u32x2*>(reinterpret_cast<const unsigned short*>(p.g1) + (nc0 + c) * HW)[ltid];
            } else if constexpr (FOLD) {
                const float* rp = p.g1 + (nc0 + c) * PHW + (lrow + 1) * PW;
                gr[c] = *reinterpret_cast<const float4u*>(rp + lx4 + 1);
                ge[c] = rp[ecol];
            } else {
                gr[c] = reinterpret_cast<const float4*>(p.g1 + (nc0 + c) * HW)[ltid];
            }
        }
        constexpr bool BR_EARLY = false;      // (with the border windows in the first phase the bf16 form spilled 11 registers between its loads: a second phase for the two waves that hold rows 1 and H - 2)
        auto load_border = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                if constexpr (!FOLD) {
                } else if constexpr (G16) {
                    br[c] = *reinterpret_cast<const u32x3*>(reinterpret_cast<const unsigned short*>(p.g1) + (nc0 + c) * PHW + prow * PW + lx4);
                } else {
                    const float* rp = p.g1 + (nc0 + c) * PHW + prow * PW;
                    br[c] = *reinterpret_cast<const float4u*>(rp + lx4 + 1);
                    be[c] = rp[ecol];
                }
            }
        };
        if constexpr (BR_EARLY) {
            if (wave_brow) load_border();
        }
        // ---- conversion and arithmetic
        const bool first = lx4 == 0, last = lx4 == W - 4;
        auto window = [&](const GRaw& w, float e) -> float4 {          // the folded group of a padded row from its raw words
            float4 v;
            if constexpr (G16) {
                const float e0 = __uint_as_float(w[0] << 16), e5 = __uint_as_float(w[2] & 0xffff0000u);
                v = make_float4(__uint_as_float(w[0] & 0xffff0000u), __uint_as_float(w[1] << 16), __uint_as_float(w[1] & 0xffff0000u),
                                __uint_as_float(w[2] << 16));
                v.y += first ? e0 : 0.f;
                v.z += last ? e5 : 0.f;
            } else {
                v = make_float4(w[0], w[1], w[2], w[3]);
                v.y += first ? e : 0.f;
                v.z += last ? e : 0.f;
            }
            return v;
        };
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float m = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(mvl), c)), r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rvl), c));
            float4 yv, gq;
            if constexpr (YB16) yv = make_float4(__uint_as_float(yr[c][0] << 16), __uint_as_float(yr[c][0] & 0xffff0000u),
                                                 __uint_as_float(yr[c][1] << 16), __uint_as_float(yr[c][1] & 0xffff0000u));
            else yv = yr[c];
            if constexpr (FOLD) {
                gq = window(gr[c], G16 ? 0.f : ge[c]);
                if constexpr (BR_EARLY) {
                    if (wave_brow) {
                        const float4 t = window(br[c], 0.f);
                        gq.x += brow ? t.x : 0.f; gq.y += brow ? t.y : 0.f; gq.z += brow ? t.z : 0.f; gq.w += brow ? t.w : 0.f;
                    }
                }
            } else if constexpr (G16) {
                gq = make_float4(__uint_as_float(gr[c][0] << 16), __uint_as_float(gr[c][0] & 0xffff0000u),
                                 __uint_as_float(gr[c][1] << 16), __uint_as_float(gr[c][1] & 0xffff0000u));
            } else {
                gq = gr[c];
            }
            const float yy[4] = {yv.x, yv.y, yv.z, yv.w}, gg[4] = {gq.x, gq.y, gq.z, gq.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float x = live ? (yy[j] - m) * r : 0.f;
                xh[c * 4 + j] = x;
                gv[c * 4 + j] = live ? gg[j] * act_grad_from_xhat(x, p.act) : 0.f;
            }
        }
        // ---- second phases (their own round trip, taken by few launches / few waves): the border rows where they did not fit the
        // first phase, and a second gradient contribution
        if constexpr (FOLD && !BR_EARLY) {
            if (wave_brow) {
                load_border();
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const float4 t = window(br[c], G16 ? 0.f : be[c]);
                    const float tt[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
                    for (int j = 0; j < 4; ++j) gv[c * 4 + j] += (brow && live) ? tt[j] * act_grad_from_xhat(xh[c * 4 + j], p.act) : 0.f;
                }
            }
        }
        if (p.g2 != nullptr) {
            float4 g2r[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) g2r[c] = reinterpret_cast<const float4*>(p.g2 + (nc0 + c) * HW)[ltid];
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float tt[4] = {g2r[c].x, g2r[c].y, g2r[c].z, g2r[c].w};
#pragma unroll
                for (int j = 0; j < 4; ++j) gv[c * 4 + j] += live ? tt[j] * act_grad_from_xhat(xh[c * 4 + j], p.act) : 0.f;
            }
        }
    }
    // plane sums of g and g * xhat, per channel: wave butterflies, then a fixed-order sum over the waves
    {
        float s[16];
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            s[c] = (gv[c * 4] + gv[c * 4 + 1]) + (gv[c * 4 + 2] + gv[c * 4 + 3]);
            s[8 + c] = (gv[c * 4] * xh[c * 4] + gv[c * 4 + 1] * xh[c * 4 + 1]) + (gv[c * 4 + 2] * xh[c * 4 + 2] + gv[c * 4 + 3] * xh[c * 4 + 3]);
        }
        // 16 values per lane -> wave totals by halving (common.h: wave_sums_to_lds): at lane bits 32, 16, 8, 4 a lane keeps one half
        // of its values and receives the partner's sums of that half, then two plain steps: 17 exchanges instead of 96
        const int lane = tid & 63;
        int cnt = 16;
#pragma unroll
        for (int bit = 32; bit >= 4; bit >>= 1) {
            const int hcnt = cnt >> 1;
            const bool up = (lane & bit) != 0;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (j < hcnt) {
                    const float keep = up ? s[hcnt + j] : s[j], give = up ? s[j] : s[hcnt + j];
                    s[j] = keep + __shfl_xor(give, bit, 64);
                }
            cnt = hcnt;
        }
        s[0] += __shfl_xor(s[0], 2, 64);
        s[0] += __shfl_xor(s[0], 1, 64);
        if ((lane & 3) == 0) red[((lane >> 5) & 1) * 8 + ((lane >> 4) & 1) * 4 + ((lane >> 3) & 1) * 2 + ((lane >> 2) & 1)][tid >> 6] = s[0];
        __syncthreads();
        if (tid < 16) {
            float t = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) t += red[tid][w];
            tot[tid] = t * (1.f / (float)HW);
        }
        __syncthreads();
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float r = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(rvl), c)), a1 = tot[c], a2 = tot[8 + c];
#pragma unroll
        for (int j = 0; j < 4; ++j) gv[c * 4 + j] = r * (gv[c * 4 + j] - a1 - xh[c * 4 + j] * a2);
    }
    const int nparts = p.heads_only ? 1 : 2;
    if (p.dy != nullptr && live) {
#pragma unroll
        for (int c = 0; c < 8; ++c)
            reinterpret_cast<float4*>(p.dy + ((long long)n * p.C + c0 + c) * HW)[tid] =
                make_float4(gv[c * 4], gv[c * 4 + 1], gv[c * 4 + 2], gv[c * 4 + 3]);
    }
    if (p.strip != nullptr && live && x4 == W - 4) {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            float* d = p.strip + (((long long)n * p.C + c0 + c) * 2) * H + row;
            d[0] = gv[c * 4 + 2];
            d[H] = gv[c * 4 + 3];
        }
    }
    if (p.xs != nullptr) {
        const int CG = p.C >> 3;
        for (int part = 0; part < nparts; ++part) {
            uint4* plane = p.xs + ((long long)(n * 2 + part) * CG + cg) * (HW + 1);
            if (live) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float v[8];
#pragma unroll
                    for (int c = 0; c < 8; ++c) v[c] = part ? bf16_tail(gv[c * 4 + j]) : gv[c * 4 + j];
                    plane[tid * 4 + j] = make_uint4(pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]),
                                                    pack_bf16x2(v[6], v[7]));
                }
            }
            if (tid == 0) plane[HW] = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    if (p.gt != nullptr) {
        const bool odd = tid & 1;                  // the high quad of the pixel octet (W % 8 == 0: the pair shares a row)
        const long long noct = (long long)p.GHp * p.GX8;
        const long long oct = (long long)row * p.GX8 + (x4 >> 3);
        for (int part = 0; part < nparts; ++part) {
            unsigned lo[8], hi[8];                 // per channel: pixels (0, 1), (2, 3) of this lane's quad
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = part ? bf16_tail(gv[c * 4 + j]) : gv[c * 4 + j];
                lo[c] = pack_bf16x2(v[0], v[1]);
                hi[c] = pack_bf16x2(v[2], v[3]);
            }
            uint4* dst = p.gt + ((long long)(n * 2 + part) * noct + oct) * p.Mp + c0 + (odd ? 4 : 0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                // the even lane keeps channels 0..3 and gives 4..7, the odd lane the other way round
                const unsigned give_lo = odd ? lo[k] : lo[4 + k], give_hi = odd ? hi[k] : hi[4 + k];
                const unsigned keep_lo = odd ? lo[4 + k] : lo[k], keep_hi = odd ? hi[4 + k] : hi[k];
                const unsigned got_lo = (unsigned)__shfl_xor((int)give_lo, 1, 64), got_hi = (unsigned)__shfl_xor((int)give_hi, 1, 64);
                if (live) dst[k] = odd ? make_uint4(got_lo, got_hi, keep_lo, keep_hi) : make_uint4(keep_lo, keep_hi, got_lo, got_hi);
            }
        }
    }
    }
}

// The same for 256 x 256 planes (unfolded gradient, HW % 4 == 0, HW <= 65536): 1024 threads x 16 float4 groups.  The
// activated gradient of the first 8 groups stays in registers, the other 8 are parked in LDS (128 KiB); the normalised
// activation is recomputed from a second read of y, which the plane's first pass left in the memory-side cache --
// 2 + (1 cached) reads + 1 write per element instead of the 4 + 1 of the reduce / apply pair.  grid: (N*C)
// OB16: dy is stored as bf16 values (its one reader is the stems' weight gradient on the bf16 matrix pipe, wgrad_k7.h, which rounds
// it to bf16 anyway: half the bytes written here and read there)
template <bool FOLD, bool OB16 = false>
__global__ __launch_bounds__(1024) void instnorm_bwd_fused_big_kernel(const float* __restrict__ g1, int p1,
                                                                     const float* __restrict__ g2,
                                                                     const float* __restrict__ y,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd, int act, int H, int W,
                                                                     float* __restrict__ dy) {
    extern __shared__ float4 park[];                 // [8][1024]
    __shared__ float red[16];
    const int nc = blockIdx.x, tid = threadIdx.x, HW = H * W, Q = HW >> 2, W4 = W >> 2;
    // p1 > 0: the gradient of a reflection-padded convolution, still in padded coordinates (W % 4 == 0: a group of four
    // pixels stays in one row)
    const FoldReader fr{g1 + (long long)nc * (H + 2 * p1) * (W + 2 * p1), H, W, p1, W + 2 * p1};
    const float m = mean[nc], r = rstd[nc];
    const float4* y4 = reinterpret_cast<const float4*>(y + (long long)nc * HW);
    const float4* ga = reinterpret_cast<const float4*>(g1 + (long long)nc * HW);
    const float4* gb = g2 ? reinterpret_cast<const float4*>(g2 + (long long)nc * HW) : nullptr;
    float4* o4 = reinterpret_cast<float4*>(dy + (long long)nc * HW);
    float4 keep[8];
    float s1 = 0.f, s2 = 0.f;
    auto take = [&](int k) -> float4 {
        const int i = k * 1024 + tid;
        float4 yv = make_float4(m, m, m, m), gq = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < Q) {
            yv = y4[i];
            if constexpr (!FOLD) {
                gq = ga[i];
            } else {
                const int yy = i / W4, xx = (i - yy * W4) * 4;
                if (p1 == 1 && H >= 3) gq = fold1_at4(fr.g, H, W, yy, xx);
                else if (p1 <= 3 && W >= 4 + 2 * p1 && H >= 2 * p1 + 2) gq = foldp_at4(fr.g, H, W, p1, yy, xx);
                else gq = make_float4(fr.at(yy, xx), fr.at(yy, xx + 1), fr.at(yy, xx + 2), fr.at(yy, xx + 3));
            }
            if (gb) { const float4 t = gb[i]; gq.x += t.x; gq.y += t.y; gq.z += t.z; gq.w += t.w; }
        }
        const float x0 = (yv.x - m) * r, x1 = (yv.y - m) * r, x2 = (yv.z - m) * r, x3 = (yv.w - m) * r;
        gq.x *= act_grad_from_xhat(x0, act); gq.y *= act_grad_from_xhat(x1, act);
        gq.z *= act_grad_from_xhat(x2, act); gq.w *= act_grad_from_xhat(x3, act);
        s1 += (gq.x + gq.y) + (gq.z + gq.w);
        s2 += (gq.x * x0 + gq.y * x1) + (gq.z * x2 + gq.w * x3);
        return gq;
    };
    // (at most four groups' loads in flight: hoisted all at once they would take the registers `keep` needs)
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        keep[k] = take(k);
        if ((k & 3) == 3) asm volatile("" ::: "memory");
    }
#pragma unroll 4
    for (int k = 8; k < 16; ++k) park[(k - 8) * 1024 + tid] = take(k);
    s1 = block_sum(s1, red);
    s2 = block_sum(s2, red);
    const float inv = 1.f / (float)HW;
    const float a1 = s1 * inv, a2 = s2 * inv;
    auto put = [&](int k, float4 g) {
        const int i = k * 1024 + tid;
        if (i < Q) {
            const float4 yv = y4[i];
            const float4 o = make_float4(r * (g.x - a1 - (yv.x - m) * r * a2), r * (g.y - a1 - (yv.y - m) * r * a2),
                                         r * (g.z - a1 - (yv.z - m) * r * a2), r * (g.w - a1 - (yv.w - m) * r * a2));
            if constexpr (OB16) reinterpret_cast<uint2*>(reinterpret_cast<unsigned short*>(dy) + (long long)nc * HW)[i] = make_uint2(pack_bf16x2(o.x, o.y), pack_bf16x2(o.z, o.w));
            else o4[i] = o;
        }
    };
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        put(k, keep[k]);
        if ((k & 3) == 3) asm volatile("" ::: "memory");
    }
#pragma unroll 4
    for (int k = 8; k < 16; ++k) put(k, park[(k - 8) * 1024 + tid]);
}

// dy = (fold(g1) + g2) * act'(out): act 1 relu / 2 lrelu (sign of the ACTIVATED output) / 3 tanh (1 - out^2)
// SUMS: the block's sum of dy goes to sums[(c * N + n) * gridDim.x + blockIdx.x] -- the partials bias_grad_final_kernel adds in fixed
// order: the bias gradient of a layer without InstanceNorm (the PatchGAN's first layer: 64 x 128 x 128 per image) costs no second pass
// over dy (round 6; bias_grad_partial_kernel re-read 1.1 GB per train step)
template <bool SUMS>
__global__ __launch_bounds__(256) void act_bwd_kernel(const float* __restrict__ g1, int p1, const float* __restrict__ g2,
                                                      const float* __restrict__ outv, int act, int H, int W,
                                                      float* __restrict__ dy, float* __restrict__ sums = nullptr, int C = 1) {
    __shared__ float sred[8];
    float ssum = 0.f;
    const int nc = blockIdx.y;
    const int HW = H * W;
    FoldReader fr{g1 + (long long)nc * (H + 2 * p1) * (W + 2 * p1), H, W, p1, W + 2 * p1};
    const float* op = outv ? outv + (long long)nc * HW : nullptr;
    const float* g2p = g2 ? g2 + (long long)nc * HW : nullptr;
    float* out = dy + (long long)nc * HW;
    if (p1 == 0 && (HW & 3) == 0) {
        const float4* ga = reinterpret_cast<const float4*>(g1 + (long long)nc * HW);
        const float4* gb = g2p ? reinterpret_cast<const float4*>(g2p) : nullptr;
        const float4* o4 = op ? reinterpret_cast<const float4*>(op) : nullptr;
        for (int i = blockIdx.x * 256 + threadIdx.x; i < HW / 4; i += gridDim.x * 256) {
            float4 gv = ga[i];
            if (gb) { const float4 t = gb[i]; gv.x += t.x; gv.y += t.y; gv.z += t.z; gv.w += t.w; }
            if (act != 0) {
                const float4 ov = o4[i];
                if (act == 1) {
                    gv.x = ov.x > 0.f ? gv.x : 0.f; gv.y = ov.y > 0.f ? gv.y : 0.f;
                    gv.z = ov.z > 0.f ? gv.z : 0.f; gv.w = ov.w > 0.f ? gv.w : 0.f;
                } else if (act == 2) {
                    gv.x = ov.x > 0.f ? gv.x : 0.2f * gv.x; gv.y = ov.y > 0.f ? gv.y : 0.2f * gv.y;
                    gv.z = ov.z > 0.f ? gv.z : 0.2f * gv.z; gv.w = ov.w > 0.f ? gv.w : 0.2f * gv.w;
                } else if (act == 3) {
                    gv.x *= 1.f - ov.x * ov.x; gv.y *= 1.f - ov.y * ov.y;
                    gv.z *= 1.f - ov.z * ov.z; gv.w *= 1.f - ov.w * ov.w;
                }
            }
            reinterpret_cast<float4*>(out)[i] = gv;
            if constexpr (SUMS) ssum += (gv.x + gv.y) + (gv.z + gv.w);
        }
    } else {
    // four elements per thread and trip, every load before the first store (one at a time this loop was four serial
    // memory round trips per thread: 150 us on the residual stream's 32 x 256 x 64 x 64 fold, 2.7 TB/s)
    for (int base = blockIdx.x * 1024; base < HW; base += gridDim.x * 1024) {
        float g[4], o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + k * 256 + threadIdx.x;
            g[k] = 0.f; o[k] = 0.f;
            if (i < HW) {
                const int yy = i / W, xx = i - yy * W;
                g[k] = fr.at(yy, xx);
                if (g2p) g[k] += g2p[i];
                if (act != 0) o[k] = op[i];
            }
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int i = base + k * 256 + threadIdx.x;
            if (i < HW) {
                float v = g[k];
                if (act == 1) v = o[k] > 0.f ? v : 0.f;
                else if (act == 2) v = o[k] > 0.f ? v : 0.2f * v;
                else if (act == 3) v *= 1.f - o[k] * o[k];
                out[i] = v;
                if constexpr (SUMS) ssum += v;
            }
        }
    }
    }
    if constexpr (SUMS) {
        ssum = block_sum(ssum, sred);
        if (threadIdx.x == 0) {
            const int n = nc / C, c = nc - n * C, N = gridDim.y / C;
            sums[((long long)c * N + n) * gridDim.x + blockIdx.x] = ssum;
        }
    }
}

// act_bwd for the gradient of a pad-1 reflection-padded convolution (the residual stream's fold + add), W % 4 == 0:
// a thread produces four consecutive pixels of a row (fold1_at4).
// grid: (ceil(H * W / 4 / 512), N*C)
__global__ __launch_bounds__(256) void act_bwd_fold1_kernel(const float* __restrict__ g1, const float* __restrict__ g2,
                                                            const float* __restrict__ outv, int act, int H, int W,
                                                            float* __restrict__ dy) {
    const int nc = blockIdx.y;
    const int HW = H * W, W4 = W >> 2;
    const float* gp = g1 + (long long)nc * (H + 2) * (W + 2);
    const float4* gb = g2 ? reinterpret_cast<const float4*>(g2 + (long long)nc * HW) : nullptr;
    const float4* o4 = outv ? reinterpret_cast<const float4*>(outv + (long long)nc * HW) : nullptr;
    float4* out = reinterpret_cast<float4*>(dy + (long long)nc * HW);
    float4 acc[2], add[2], ov[2];
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = (blockIdx.x * 2 + k) * 256 + threadIdx.x;
        acc[k] = add[k] = ov[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < HW / 4) {
            const int y = i / W4;
            acc[k] = fold1_at4(gp, H, W, y, (i - y * W4) * 4);
            if (gb) add[k] = gb[i];
            if (act != 0) ov[k] = o4[i];
        }
    }
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const int i = (blockIdx.x * 2 + k) * 256 + threadIdx.x;
        if (i < HW / 4) {
            float4 gv = make_float4(acc[k].x + add[k].x, acc[k].y + add[k].y, acc[k].z + add[k].z, acc[k].w + add[k].w);
            const float4 o = ov[k];
            if (act == 1) {
                gv.x = o.x > 0.f ? gv.x : 0.f; gv.y = o.y > 0.f ? gv.y : 0.f;
                gv.z = o.z > 0.f ? gv.z : 0.f; gv.w = o.w > 0.f ? gv.w : 0.f;
            } else if (act == 2) {
                gv.x = o.x > 0.f ? gv.x : 0.2f * gv.x; gv.y = o.y > 0.f ? gv.y : 0.2f * gv.y;
                gv.z = o.z > 0.f ? gv.z : 0.2f * gv.z; gv.w = o.w > 0.f ? gv.w : 0.2f * gv.w;
            } else if (act == 3) {
                gv.x *= 1.f - o.x * o.x; gv.y *= 1.f - o.y * o.y; gv.z *= 1.f - o.z * o.z; gv.w *= 1.f - o.w * o.w;
            }
            out[i] = gv;
        }
    }
}

// grid: (C).  db[c] = sum_{n, pix} dy[n, c, pix]
__global__ __launch_bounds__(256) void bias_grad_kernel(const float* __restrict__ dy, int N, int C, int HW,
                                                        float* __restrict__ db) {
    __shared__ float red[8];
    const int c = blockIdx.x;
    float s = 0.f;
    for (int n = 0; n < N; ++n) {
        const float* p = dy + ((long long)n * C + c) * HW;
        for (int i = threadIdx.x; i < HW; i += 256) s += p[i];
    }
    s = block_sum(s, red);
    if (threadIdx.x == 0) db[c] = s;
}

// Two-stage form for few channels (C = 1: the final layer of the generator and of every PatchGAN): stage 1 reduces
// SPLIT slices of every (n, c) plane in parallel, stage 2 adds the N * SPLIT partials of a channel in a fixed order.
__global__ __launch_bounds__(256) void bias_grad_partial_kernel(const float* __restrict__ dy, int C, int HW, int split,
                                                                float* __restrict__ ws) {
    __shared__ float red[8];
    const int c = blockIdx.x, n = blockIdx.y, sp = blockIdx.z;
    const int per = (HW + split - 1) / split;
    const int lo = sp * per, hi = lo + per < HW ? lo + per : HW;
    const float* p = dy + ((long long)n * C + c) * HW;
    float s = 0.f;
    for (int i = lo + threadIdx.x; i < hi; i += 256) s += p[i];
    s = block_sum(s, red);
    if (threadIdx.x == 0) ws[((long long)c * gridDim.y + n) * split + sp] = s;
}

__global__ __launch_bounds__(64) void bias_grad_final_kernel(const float* __restrict__ ws, int count, float* __restrict__ db) {
    const int c = blockIdx.x;
    if (threadIdx.x != 0) return;
    float s = 0.f;
    for (int i = 0; i < count; ++i) s += ws[(long long)c * count + i];
    db[c] = s;
}

// db[c] = sum of the count block sums act_bwd_kernel<true> left for channel c: lane t adds elements t, t + 64, ..., then the
// lanes are added as a fixed tree
__global__ __launch_bounds__(64) void bias_sums_final_kernel(const float* __restrict__ ws, int count, float* __restrict__ db) {
    const int c = blockIdx.x;
    float s = 0.f;
    for (int i = threadIdx.x; i < count; i += 64) s += ws[(long long)c * count + i];
#pragma unroll
    for (int sh = 32; sh >= 1; sh >>= 1) s += __shfl_xor(s, sh, 64);
    if (threadIdx.x == 0) db[c] = s;
}

}  // namespace apamd

using namespace apamd;

static int fold_args_ok(const float* g1, int p1, int H, int W, const char* who) {
    if (!g1) return fail(AP_ERR_INVALID, "%s: null gradient", who);
    if (p1 < 0 || (p1 > 0 && (p1 >= H || p1 >= W))) return fail(AP_ERR_INVALID, "%s: fold pad %d vs %dx%d", who, p1, H, W);
    return AP_OK;
}

extern "C" {

int ap_instnorm_bwd(const float* g1, int32_t g1_pad, const float* g2, const float* y, const float* mean,
                    const float* rstd, int32_t act, int32_t NC, int32_t H, int32_t W, float* sums_ws, float* dy,
                    ap_stream_t stream) {
    int rc = fold_args_ok(g1, g1_pad, H, W, "instnorm_bwd");
    if (rc) return rc;
    if (!y || !mean || !rstd || !sums_ws || !dy) return fail(AP_ERR_INVALID, "instnorm_bwd: null pointer");
    const bool ob16 = (act & 0x100) != 0;       // bit 8: dy is stored as bf16 values (256 x 256-class planes without a fold only)
    act &= 0xff;
    if (act < 0 || act > 2) return fail(AP_ERR_INVALID, "instnorm_bwd: act %d", act);
    if (NC < 1 || NC > 65535) return fail(AP_ERR_UNSUPPORTED, "instnorm_bwd: N*C=%d", NC);
    if (ob16 && !(g1_pad == 0 && H * W > 16384 && H * W <= 65536 && (W % 4) == 0))
        return fail(AP_ERR_UNSUPPORTED, "instnorm_bwd: a bf16 dy is written by the big-plane kernel only (%dx%d, fold %d)", H, W, g1_pad);
    constexpr bool fused_ok = true;
    const bool plain_vec = g1_pad == 0 && ((H * W) & 3) == 0;     // unfolded gradient, whole 16-byte groups: the load-phase form
    const bool fold1_vec = g1_pad == 1 && (W & 3) == 0 && H >= 3 && W >= 8;   // pad-1 fold, the same
    if (fused_ok && g1_pad == 0 && H * W <= 1024 && ((H * W) & 3) != 0) {      // small planes that are not whole 16-byte groups
        if (g2) hipLaunchKernelGGL(instnorm_bwd_fused_small_kernel<true>, dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H * W, dy);
        else hipLaunchKernelGGL(instnorm_bwd_fused_small_kernel<false>, dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H * W, dy);
        return check_launch("instnorm_bwd_fused_small_kernel");
    }
    if (fused_ok && H * W <= 4096) {
        if (plain_vec && g2) hipLaunchKernelGGL((instnorm_bwd_fused_vec_kernel<256, 16, true>), dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H * W, dy);
        else if (plain_vec) hipLaunchKernelGGL((instnorm_bwd_fused_vec_kernel<256, 16, false>), dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H * W, dy);
        else if (fold1_vec && g2) hipLaunchKernelGGL((instnorm_bwd_fused_fold1_kernel<256, 16, true>), dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H, W, dy);
        else if (fold1_vec) hipLaunchKernelGGL((instnorm_bwd_fused_fold1_kernel<256, 16, false>), dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H, W, dy);
        else hipLaunchKernelGGL((instnorm_bwd_fused_kernel<256, 16>), dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g1_pad, g2,
                                y, mean, rstd, act, H, W, dy);
        return check_launch("instnorm_bwd_fused_kernel");
    }
    if (fused_ok && H * W <= 16384) {
        if (plain_vec && g2) hipLaunchKernelGGL((instnorm_bwd_fused_vec_kernel<1024, 16, true>), dim3(NC), dim3(1024), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H * W, dy);
        else if (plain_vec) hipLaunchKernelGGL((instnorm_bwd_fused_vec_kernel<1024, 16, false>), dim3(NC), dim3(1024), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H * W, dy);
        else if (fold1_vec && g2) hipLaunchKernelGGL((instnorm_bwd_fused_fold1_kernel<1024, 16, true>), dim3(NC), dim3(1024), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H, W, dy);
        else if (fold1_vec) hipLaunchKernelGGL((instnorm_bwd_fused_fold1_kernel<1024, 16, false>), dim3(NC), dim3(1024), 0, (hipStream_t)stream, g1, g2, y, mean, rstd, act, H, W, dy);
        else hipLaunchKernelGGL((instnorm_bwd_fused_kernel<1024, 16>), dim3(NC), dim3(1024), 0, (hipStream_t)stream, g1, g1_pad,
                                g2, y, mean, rstd, act, H, W, dy);
        return check_launch("instnorm_bwd_fused_kernel");
    }
    if (fused_ok && H * W <= 65536 && (W % 4) == 0 && W >= 4) {
        static bool attr = false;
        const size_t lds = 8 * 1024 * sizeof(float4);
        if (!attr) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&instnorm_bwd_fused_big_kernel<false>),
                                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&instnorm_bwd_fused_big_kernel<true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e == hipSuccess)
                e = hipFuncSetAttribute(reinterpret_cast<const void*>(&instnorm_bwd_fused_big_kernel<false, true>),
                                        hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "hipFuncSetAttribute: %s", hipGetErrorString(e));
            attr = true;
        }
        if (ob16)
            hipLaunchKernelGGL((instnorm_bwd_fused_big_kernel<false, true>), dim3(NC), dim3(1024), lds, (hipStream_t)stream, g1, g1_pad,
                               g2, y, mean, rstd, act, H, W, dy);
        else if (g1_pad == 0)
            hipLaunchKernelGGL(instnorm_bwd_fused_big_kernel<false>, dim3(NC), dim3(1024), lds, (hipStream_t)stream, g1, g1_pad,
                               g2, y, mean, rstd, act, H, W, dy);
        else
            hipLaunchKernelGGL(instnorm_bwd_fused_big_kernel<true>, dim3(NC), dim3(1024), lds, (hipStream_t)stream, g1, g1_pad,
                               g2, y, mean, rstd, act, H, W, dy);
        return check_launch("instnorm_bwd_fused_big_kernel");
    }
    hipLaunchKernelGGL(instnorm_bwd_reduce_kernel, dim3(NC), dim3(256), 0, (hipStream_t)stream, g1, g1_pad, g2, y,
                       mean, rstd, act, H, W, sums_ws);
    rc = check_launch("instnorm_bwd_reduce_kernel");
    if (rc) return rc;
    int bx = (H * W + 1023) / 1024;
    if (bx > 32) bx = 32;
    hipLaunchKernelGGL(instnorm_bwd_apply_kernel, dim3(bx, NC), dim3(256), 0, (hipStream_t)stream, g1, g1_pad, g2, y,
                       mean, rstd, act, H, W, sums_ws, dy);
    return check_launch("instnorm_bwd_apply_kernel");
}

int ap_instnorm_bwd_split_ok(int32_t C, int32_t H, int32_t W, int32_t g1_pad) {
    const char* off = getenv("APAMD_NO_INBWD_SPLIT");      // (read per call: tests and A/B runs flip it inside one process)
    return !(off && atoi(off)) && C >= 8 && C % 8 == 0 && H >= 3 && W >= 8 && W % 8 == 0 && H * W <= 4096 && (g1_pad == 0 || g1_pad == 1) ? 1 : 0;
}

int ap_instnorm_bwd_split(const float* g1, int32_t g1_pad, const float* g2, const float* y, const float* mean, const float* rstd,
                          int32_t act, int32_t N, int32_t C, int32_t H, int32_t W, void* xs, void* gt, const int32_t* gt_dims,
                          float* strip, float* dy, int32_t heads_only, ap_stream_t stream) {
    int rc = fold_args_ok(g1, g1_pad, H, W, "instnorm_bwd_split");
    if (rc) return rc;
    if (!y || !mean || !rstd) return fail(AP_ERR_INVALID, "instnorm_bwd_split: null pointer");
    if (!xs && !gt && !dy) return fail(AP_ERR_INVALID, "instnorm_bwd_split: no output requested");
    if (act < 0 || act > 2) return fail(AP_ERR_INVALID, "instnorm_bwd_split: act %d", act);
    if (!ap_instnorm_bwd_split_ok(C, H, W, g1_pad))
        return fail(AP_ERR_UNSUPPORTED, "instnorm_bwd_split: C=%d %dx%d fold %d (needs C %% 8 == 0, W %% 8 == 0, H >= 3, H*W <= 4096)", C, H, W, g1_pad);
    if (N < 1 || N > 65535) return fail(AP_ERR_UNSUPPORTED, "instnorm_bwd_split: N=%d", N);
    InBwdSplitParams p;
    memset(&p, 0, sizeof(p));
    p.g1 = g1; p.g2 = g2; p.y = y; p.mean = mean; p.rstd = rstd;
    p.p1 = g1_pad; p.act = act; p.C = C; p.H = H; p.W = W;
    p.xs = reinterpret_cast<uint4*>(xs);
    p.gt = reinterpret_cast<uint4*>(gt);
    if (gt) {
        if (!gt_dims) return fail(AP_ERR_INVALID, "instnorm_bwd_split: gt without its dimensions");
        p.GHp = gt_dims[0]; p.GX8 = gt_dims[1]; p.Mp = gt_dims[2];
        if (p.GHp < H || p.GX8 * 8 < W || p.Mp < C)
            return fail(AP_ERR_INVALID, "instnorm_bwd_split: operand %d x %d x %d smaller than the gradient", p.GHp, p.GX8, p.Mp);
    }
    // heads_only: bit 0 = only head planes are written; bit 1 = y holds bf16 values; bit 2 = g1 holds bf16 values
    p.strip = strip; p.dy = dy; p.heads_only = (heads_only & 1) ? 1 : 0;
    const bool yb16 = (heads_only & 2) != 0, g16 = (heads_only & 4) != 0;
    p.N = N;
    static int cus = 0;
    if (cus == 0) {
        int dev = 0;
        hipDeviceProp_t prop;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return fail(AP_ERR_LAUNCH, "instnorm_bwd_split: no device");
        cus = prop.multiProcessorCount;
    }
    const int items = N * (C / 8);
    const bool small = H * W / 4 <= 256;
    const int slots = cus * (small ? 4 : 1);          // resident workgroups: registers hold one 1024-thread item per CU, four of 256
    const dim3 grid(items < slots ? items : slots);
    auto launch = [&](auto nt) {
        constexpr int NTH = decltype(nt)::value;
        auto go = [&](auto yt, auto gt_, auto ft) {
            hipLaunchKernelGGL((instnorm_bwd_split_kernel<NTH, decltype(yt)::value, decltype(gt_)::value, decltype(ft)::value>), grid, dim3(NTH), 0,
                               (hipStream_t)stream, p);
        };
        using T = std::true_type;
        using F = std::false_type;
        const int sel = (yb16 ? 4 : 0) | (g16 ? 2 : 0) | (g1_pad == 1 ? 1 : 0);
        switch (sel) {
            case 0: go(F{}, F{}, F{}); break;
            case 1: go(F{}, F{}, T{}); break;
            case 2: go(F{}, T{}, F{}); break;
            case 3: go(F{}, T{}, T{}); break;
            case 4: go(T{}, F{}, F{}); break;
            case 5: go(T{}, F{}, T{}); break;
            case 6: go(T{}, T{}, F{}); break;
            default: go(T{}, T{}, T{}); break;
        }
    };
    if (small) launch(std::integral_constant<int, 256>{});
    else launch(std::integral_constant<int, 1024>{});
    return check_launch("instnorm_bwd_split_kernel");
}

int ap_act_bwd(const float* g1, int32_t g1_pad, const float* g2, const float* out, int32_t act, int32_t NC,
               int32_t H, int32_t W, float* dy, ap_stream_t stream) {
    int rc = fold_args_ok(g1, g1_pad, H, W, "act_bwd");
    if (rc) return rc;
    if (!dy || (act != AP_ACT_NONE && !out)) return fail(AP_ERR_INVALID, "act_bwd: null pointer");
    if (act < 0 || act > 3) return fail(AP_ERR_INVALID, "act_bwd: act %d", act);
    if (NC < 1 || NC > 65535) return fail(AP_ERR_UNSUPPORTED, "act_bwd: N*C=%d", NC);
    int bx = (H * W + 1023) / 1024;
    if (bx > 32) bx = 32;
    if (g1_pad == 1 && (W & 3) == 0 && H >= 3 && W >= 4) {
        hipLaunchKernelGGL(act_bwd_fold1_kernel, dim3((H * W / 4 + 511) / 512, NC), dim3(256), 0, (hipStream_t)stream, g1,
                           g2, out, act, H, W, dy);
        return check_launch("act_bwd_fold1_kernel");
    }
    hipLaunchKernelGGL(act_bwd_kernel<false>, dim3(bx, NC), dim3(256), 0, (hipStream_t)stream, g1, g1_pad, g2, out, act, H, W,
                       dy, (float*)nullptr, 1);
    return check_launch("act_bwd_kernel");
}

// act_bwd + the bias gradient db[c] = sum_{n, y, x} dy of the same layer in one pass (workspace: ap_act_bwd_bias_workspace_floats)
int64_t ap_act_bwd_bias_workspace_floats(int32_t N, int32_t C, int32_t H, int32_t W) {
    if (N < 1 || C < 1 || H < 1 || W < 1) return fail(AP_ERR_INVALID, "act_bwd_bias: bad sizes");
    int bx = (H * W + 1023) / 1024;
    if (bx > 32) bx = 32;
    return (int64_t)N * C * bx;
}

int ap_act_bwd_bias(const float* g1, int32_t g1_pad, const float* g2, const float* out, int32_t act, int32_t N, int32_t C,
                    int32_t H, int32_t W, float* dy, float* workspace, float* db, ap_stream_t stream) {
    int rc = fold_args_ok(g1, g1_pad, H, W, "act_bwd_bias");
    if (rc) return rc;
    if (!dy || !workspace || !db || (act != AP_ACT_NONE && !out)) return fail(AP_ERR_INVALID, "act_bwd_bias: null pointer");
    if (act < 0 || act > 3) return fail(AP_ERR_INVALID, "act_bwd_bias: act %d", act);
    if (N < 1 || C < 1 || (long long)N * C > 65535) return fail(AP_ERR_UNSUPPORTED, "act_bwd_bias: N*C=%lld", (long long)N * C);
    int bx = (H * W + 1023) / 1024;
    if (bx > 32) bx = 32;
    hipLaunchKernelGGL(act_bwd_kernel<true>, dim3(bx, N * C), dim3(256), 0, (hipStream_t)stream, g1, g1_pad, g2, out, act, H, W, dy,
                       workspace, C);
    rc = check_launch("act_bwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(bias_sums_final_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, workspace, N * bx, db);
    return check_launch("bias_sums_final_kernel");
}

int ap_bias_grad(const float* dy, int32_t N, int32_t C, int32_t HW, float* db, ap_stream_t stream) {
    if (!dy || !db || N < 1 || C < 1 || HW < 1) return fail(AP_ERR_INVALID, "bias_grad: bad arguments");
    hipLaunchKernelGGL(bias_grad_kernel, dim3(C), dim3(256), 0, (hipStream_t)stream, dy, N, C, HW, db);
    return check_launch("bias_grad_kernel");
}

static int bias_grad_split(int N, int C, int HW) {
    // enough slices to put ~2 workgroups on every CU, each still streaming >= 4096 elements
    int split = 512 / (N * C > 0 ? N * C : 1);
    const int cap = HW / 4096;
    if (split > cap) split = cap;
    return split < 1 ? 1 : split;
}

int64_t ap_bias_grad_workspace_floats(int32_t N, int32_t C, int32_t HW) {
    if (N < 1 || C < 1 || HW < 1) return fail(AP_ERR_INVALID, "bias_grad: bad arguments");
    return (int64_t)N * C * bias_grad_split(N, C, HW);
}

int ap_bias_grad_ws(const float* dy, int32_t N, int32_t C, int32_t HW, float* workspace, float* db, ap_stream_t stream) {
    if (!dy || !db || !workspace || N < 1 || C < 1 || HW < 1) return fail(AP_ERR_INVALID, "bias_grad: bad arguments");
    if (N > 65535) return fail(AP_ERR_UNSUPPORTED, "bias_grad: N too large");
    const int split = bias_grad_split(N, C, HW);
    hipLaunchKernelGGL(bias_grad_partial_kernel, dim3(C, N, split), dim3(256), 0, (hipStream_t)stream, dy, C, HW, split,
                       workspace);
    int rc = check_launch("bias_grad_partial_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(bias_grad_final_kernel, dim3(C), dim3(64), 0, (hipStream_t)stream, workspace, N * split, db);
    return check_launch("bias_grad_final_kernel");
}

}  // extern "C"
