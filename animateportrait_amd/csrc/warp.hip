// warp.hip -- fused double_feature_warping (reference: Module2/models/networks.py:1298-1313 and
// warp_acc_flow, Module2/intrinsic_flow_models/modules.py:596-625).
//
// One pass produces the 2C-channel concat the next convolution reads:
//   out[:, 0:C ] = grid_sample(x, motion_L)                       bilinear / zeros / align_corners=False
//   out[:, C:2C] = where(mask_L > 0.5, grid_sample(x, grid(flow_L)), -1)
// motion_L, flow_L, mask_L are the align_corners=True bilinear resizes of the full-resolution inputs
// to the feature resolution, evaluated per output pixel instead of being materialised, and x may be a
// raw conv output whose InstanceNorm+ReLU is applied per gathered tap.
//
// HBM-bound gather: one lane per output pixel (consecutive lanes = consecutive x, so the two stores
// per channel are coalesced 256-B rows); sample coordinates and the 2x4 tap offsets/weights are
// computed once per pixel and reused for CG channels.
#include "common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace apamd {

struct Taps {
    int off[4];     // offset inside a channel plane, or -1 when the tap is out of range (contributes 0)
    float w[4];
};

__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W) {
    // grid_sampler_unnormalize, align_corners=False: ((g + 1) * size - 1) / 2
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);   // everything beyond is all-zero taps anyway
    iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ex = fx + 1.f, ey = fy + 1.f;
    Taps t;
    t.w[0] = (ex - ix) * (ey - iy);   // nw
    t.w[1] = (ix - fx) * (ey - iy);   // ne
    t.w[2] = (ex - ix) * (iy - fy);   // sw
    t.w[3] = (ix - fx) * (iy - fy);   // se
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
    t.off[0] = (xin0 && yin0) ? y0 * W + x0 : -1;
    t.off[1] = (xin1 && yin0) ? y0 * W + x1 : -1;
    t.off[2] = (xin0 && yin1) ? y1 * W + x0 : -1;
    t.off[3] = (xin1 && yin1) ? y1 * W + x1 : -1;
    return t;
}

// north-west corner of a sample (the clamped floor make_taps uses): what the quad-cooperative gather addresses rows by
__device__ __forceinline__ void tap_corner(float gx, float gy, int H, int W, int& x0, int& y0) {
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);
    iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);
    x0 = (int)floorf(ix);
    y0 = (int)floorf(iy);
}

struct Lerp { int i0, i1; float l0, l1; };

// F.interpolate(mode='bilinear', align_corners=True): src = dst * (S-1)/(H-1)
__device__ __forceinline__ Lerp make_lerp(int dst, int S, int H) {
    const float scale = H > 1 ? (float)(S - 1) / (float)(H - 1) : 0.f;
    const float src = scale * (float)dst;
    Lerp l;
    l.i0 = (int)src;
    if (l.i0 > S - 1) l.i0 = S - 1;
    l.i1 = l.i0 + (l.i0 < S - 1 ? 1 : 0);
    l.l1 = src - (float)l.i0;
    l.l0 = 1.f - l.l1;
    return l;
}

__device__ __forceinline__ float bilerp(float v00, float v01, float v10, float v11, const Lerp& ly, const Lerp& lx) {
    return ly.l0 * (lx.l0 * v00 + lx.l1 * v01) + ly.l1 * (lx.l0 * v10 + lx.l1 * v11);
}

__device__ __forceinline__ float tap_val(const float* plane, int off, float m, float r, int act) {
    if (off < 0) return 0.f;
    float v = (plane[off] - m) * r;
    if (act == 1) v = v > 0.f ? v : 0.f;
    else if (act == 2) v = v > 0.f ? v : 0.2f * v;
    return v;
}

constexpr int kWarpCG = 8;   // channels per thread = one 16-byte slot of the split-bf16 layout

typedef __bf16 wbf16x8 __attribute__((ext_vector_type(8)));

// store 8 channels of one pixel as a head / tail slot pair of XS[n][part][cg][HW + 1] (conv_bf16x3.h)
__device__ __forceinline__ void store_split_slot(uint4* xs, int n, int CG2, int cg, int HW, int pix, const float (&v)[8]) {
    wbf16x8 hv, lv;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const __bf16 h = (__bf16)v[c];
        hv[c] = h;
        lv[c] = (__bf16)(v[c] - (float)h);
    }
    *reinterpret_cast<wbf16x8*>(xs + ((long long)(n * 2 + 0) * CG2 + cg) * (HW + 1) + pix) = hv;
    *reinterpret_cast<wbf16x8*>(xs + ((long long)(n * 2 + 1) * CG2 + cg) * (HW + 1) + pix) = lv;
}

// The same slot pair in the SPACE-TO-DEPTH split layout a stride-2 3x3 consumer stages (conv_bf16x3.h, split_s2d_kernel):
//   X'[(ry * 2 + rx) * C2 + c][qy][qx] = pad1(v)[c][2 qy + ry][2 qx + rx]   on an (H/2 + 1) x (W/2 + 1) map,
// so pixel (y, x) lands in phase ((y + 1) & 1, (x + 1) & 1) at ((y + 1) >> 1, (x + 1) >> 1).  The pixels of the image
// border also write the all-zero slots of the padding ring next to them (every phase plane has one zero row and column).
__device__ __forceinline__ void store_split_slot_s2d(uint4* xs, int n, int CG2, int cg, int H, int W, int y, int x,
                                                     const float (&v)[8]) {
    const int H2 = H / 2 + 1, W2 = W / 2 + 1, HW2 = H2 * W2;
    wbf16x8 hv, lv;
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const __bf16 h = (__bf16)v[c];
        hv[c] = h;
        lv[c] = (__bf16)(v[c] - (float)h);
    }
    auto plane = [&](int part, int ry, int rx) { return xs + ((long long)(n * 2 + part) * (4 * CG2) + (ry * 2 + rx) * CG2 + cg) * (HW2 + 1); };
    const int py = y + 1, px = x + 1;
    const int ry = py & 1, rx = px & 1, qy = py >> 1, qx = px >> 1;
    *reinterpret_cast<wbf16x8*>(plane(0, ry, rx) + qy * W2 + qx) = hv;
    *reinterpret_cast<wbf16x8*>(plane(1, ry, rx) + qy * W2 + qx) = lv;
    const uint4 z = make_uint4(0u, 0u, 0u, 0u);
    const bool top = y == 0, bot = y == H - 1, lef = x == 0, rig = x == W - 1;
    if (top || bot) {           // padded rows 0 / H + 1 at this column
        const int zy = top ? 0 : H + 1;
        plane(0, zy & 1, rx)[(zy >> 1) * W2 + qx] = z;
        plane(1, zy & 1, rx)[(zy >> 1) * W2 + qx] = z;
    }
    if (lef || rig) {           // padded columns 0 / W + 1 at this row
        const int zx = lef ? 0 : W + 1;
        plane(0, ry, zx & 1)[qy * W2 + (zx >> 1)] = z;
        plane(1, ry, zx & 1)[qy * W2 + (zx >> 1)] = z;
    }
    if ((top || bot) && (lef || rig)) {     // the four corners of the ring
        const int zy = top ? 0 : H + 1, zx = lef ? 0 : W + 1;
        plane(0, zy & 1, zx & 1)[(zy >> 1) * W2 + (zx >> 1)] = z;
        plane(1, zy & 1, zx & 1)[(zy >> 1) * W2 + (zx >> 1)] = z;
    }
    if (y == 0 && x == 0) {     // the closing all-zero slot of the four phase planes of this channel group
#pragma unroll
        for (int r = 0; r < 4; ++r) { plane(0, r >> 1, r & 1)[HW2] = z; plane(1, r >> 1, r & 1)[HW2] = z; }
    }
}

// grid: (ceil(H*W/256), ceil(C/CG), N).  out (fp32 [N, 2C, H, W]) and xs (its split-bf16 copy) are both optional.
// ACT (= x_act) is a template parameter: with a run-time activation the compiler evaluated ReLU AND LeakyReLU for every
// one of the 64 gathered values and selected (12 vector instructions per value; the kernel is as much VALU- as memory-bound).
// GATHER = 1 (tools / A-B runs only, APAMD_WARP_GATHER=quad): the "wavefront-shuffle" gather BASELINE.json names, in its cheapest
// form for the channel-octet layout -- the four lanes of a quad fetch each other's 64-byte tap-row segments (two neighbouring
// pixels x 8 channels) as ONE contiguous request per member and row, then transpose the 4 x 4 x 16-byte block inside the quad
// with DPP quad_perm exchanges: a quarter of the L1 requests of the lane-per-pixel gather for ~200 more vector instructions per
// pixel.  Measured slower (profiles/r04_warp_variants.md); kept so that the rejection has its measurement.
#ifndef APAMD_WARP_WPE
#define APAMD_WARP_WPE 4
#endif
template <int ACT, int WPE, int GATHER = 0>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void warp_concat_kernel(const float* __restrict__ x, const float* __restrict__ x_mean,
                                                          const float* __restrict__ x_rstd, int x_act,
                                                          const float* __restrict__ motion,
                                                          const float* __restrict__ flow,
                                                          const float* __restrict__ ifmask, float* __restrict__ out,
                                                          uint4* __restrict__ xs,
                                                          int C, int H, int W, int S, float flow_scale, int flags,
                                                          int tw_shift) {
    const int s2d = flags & 1;
    const int xoct = flags & 2;      // x is the channel-octet layout [N][C/8][H*W][8] (ap_conv2d_fwd_octet)
    // logical block (pixel block fastest, then channel group, then image): contiguous per XCD (common.h) -- the gathers of
    // neighbouring rows and the motion / flow / mask lines of an image then hit ONE L2 (r03z: 481 MB fetched per launch
    // for 78 MB of input when every XCD saw every eighth row of every plane)
    int bx, by, bz;
    {
        const unsigned L = xcd_logical_block(gridDim.x * gridDim.y * gridDim.z,
                                             blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
        bx = L % gridDim.x;
        const unsigned t = L / gridDim.x;
        by = t % gridDim.y;
        bz = t / gridDim.y;
    }
    if (xs != nullptr && !s2d && bx == 0 && threadIdx.x < 4) {
        // the all-zero slot that closes every plane of the split layout (this block's two channel groups x 2 parts)
        const int CG2 = (2 * C) >> 3, HWz = H * W;
        const int part = threadIdx.x & 1, cg = (threadIdx.x >> 1) ? (C >> 3) + by : by;
        xs[((long long)(bz * 2 + part) * CG2 + cg) * (HWz + 1) + HWz] = make_uint4(0u, 0u, 0u, 0u);
    }
    // a workgroup covers a tw x (256 / tw) pixel tile when the map divides into such tiles (tw_shift > 0), else 256
    // consecutive pixels: the taps of a compact tile fall into a window the CU's L1 holds, those of a 256-pixel row
    // segment (16 noisy rows high) do not -- the gathers are then served line by line from the L2
    int pix, oy, ox;
    if (tw_shift > 0) {
        const int tiles_x = W >> tw_shift;
        const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
        oy = ty * (256 >> tw_shift) + ((int)threadIdx.x >> tw_shift);
        ox = (tx << tw_shift) + ((int)threadIdx.x & ((1 << tw_shift) - 1));
        pix = oy * W + ox;
    } else {
        pix = bx * 256 + threadIdx.x;
        if (pix >= H * W) return;
        oy = pix / W;
        ox = pix - oy * W;
    }
    const int n = bz;
    const long long SS = (long long)S * S;

    float gx, gy, fx, fy, mk;
    if (H == S && W == S) {
        const float2 g = reinterpret_cast<const float2*>(motion)[n * SS + pix];
        gx = g.x; gy = g.y;
        fx = flow[(n * 2 + 0) * SS + pix] * flow_scale;
        fy = flow[(n * 2 + 1) * SS + pix] * flow_scale;
        mk = ifmask[n * SS + pix];
    } else {
        const Lerp ly = make_lerp(oy, S, H), lx = make_lerp(ox, S, W);
        const int o00 = ly.i0 * S + lx.i0, o01 = ly.i0 * S + lx.i1, o10 = ly.i1 * S + lx.i0, o11 = ly.i1 * S + lx.i1;
        const float2* mo = reinterpret_cast<const float2*>(motion) + n * SS;
        const float2 a = mo[o00], b = mo[o01], c = mo[o10], d = mo[o11];
#ifdef APAMD_HZ_NOP
        asm volatile("s_nop 4" ::: "memory");       // co-residency lab, victim-side variant: idle states behind the four tap loads
#endif
        gx = bilerp(a.x, b.x, c.x, d.x, ly, lx);
        gy = bilerp(a.y, b.y, c.y, d.y, ly, lx);
        const float* f0 = flow + (n * 2 + 0) * SS;
        const float* f1 = flow + (n * 2 + 1) * SS;
        // the reference resizes flow / 2^level; the scale is a power of two, so scaling taps is exact
        fx = bilerp(f0[o00] * flow_scale, f0[o01] * flow_scale, f0[o10] * flow_scale, f0[o11] * flow_scale, ly, lx);
        fy = bilerp(f1[o00] * flow_scale, f1[o01] * flow_scale, f1[o10] * flow_scale, f1[o11] * flow_scale, ly, lx);
        const float* mp = ifmask + n * SS;
        mk = bilerp(mp[o00], mp[o01], mp[o10], mp[o11], ly, lx);
    }
    const Taps tm = make_taps(gx, gy, H, W);
    // warp_acc_flow: grid = 2 * (pixel + flow) / max(size - 1, 1) - 1
    const float wgx = 2.0f * ((float)ox + fx) / (float)(W - 1 > 1 ? W - 1 : 1) - 1.0f;
    const float wgy = 2.0f * ((float)oy + fy) / (float)(H - 1 > 1 ? H - 1 : 1) - 1.0f;
    const Taps tf = make_taps(wgx, wgy, H, W);
    const bool keep = mk > 0.5f;

    const int HW = H * W;
    const int c0 = by * kWarpCG;
    const int c1 = c0 + kWarpCG < C ? c0 + kWarpCG : C;
    if (c0 + kWarpCG <= C) {
        // full channel group: all 64 gathers of the thread are issued before the first use (out-of-range taps read
        // offset 0 with weight 0: same sums as skipping them), then the 8 + 8 results leave as coalesced rows
        int om[4], of[4];
        float wm[4], wf[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            om[k] = tm.off[k] < 0 ? 0 : tm.off[k];
            wm[k] = tm.off[k] < 0 ? 0.f : tm.w[k];
            of[k] = (tf.off[k] < 0 || !keep) ? 0 : tf.off[k];
            wf[k] = tf.off[k] < 0 ? 0.f : tf.w[k];
        }
        float v1[8], v2[8];
        float a[8][4], b[8][4], m[8], r[8];
        if (xoct) {
            // a tap is the 32 contiguous bytes of the group's 8 channels: two 16-byte loads instead of 8 dword gathers
            // from 8 planes (the NCHW form is bound by the L1's line rate: every lane's tap is its own line per channel)
            const float4* og = reinterpret_cast<const float4*>(x + ((long long)n * (C >> 3) + by) * HW * 8);
            if constexpr (GATHER == 1) {
                const int qi = (int)threadIdx.x & 3;
                auto dppi = [](int v, auto ctl) { return __builtin_amdgcn_update_dpp(0, v, decltype(ctl)::value, 0xf, 0xf, false); };
                auto dppf = [&](float v, auto ctl) { return __int_as_float(dppi(__float_as_int(v), ctl)); };
                using BC0 = std::integral_constant<int, 0x00>; using BC1 = std::integral_constant<int, 0x55>;
                using BC2 = std::integral_constant<int, 0xAA>; using BC3 = std::integral_constant<int, 0xFF>;
                using X1 = std::integral_constant<int, 0xB1>;  using X2 = std::integral_constant<int, 0x4E>;
                // one tap row of one sampler: own segment = pixels (xb, xb + 1) of row y, 4 x 16 bytes; returns the segment's chunks
                auto row_segment = [&](int y, int xb, float4 (&B)[4]) __attribute__((always_inline)) {
                    const int base = (y * W + xb) * 2;
                    const int b0 = dppi(base, BC0{}), b1 = dppi(base, BC1{}), b2 = dppi(base, BC2{}), b3 = dppi(base, BC3{});
                    B[0] = og[b0 + qi]; B[1] = og[b1 + qi]; B[2] = og[b2 + qi]; B[3] = og[b3 + qi];     // round j: member j's segment
                    // 4 x 4 transpose of 16-byte elements inside the quad: B[j] of lane i  ->  B[i] of lane j
                    const bool o1 = qi & 1, o2 = qi & 2;
                    auto xchg = [&](float4& lo, float4& hi, bool odd, auto ctl) __attribute__((always_inline)) {
                        float4 snd = odd ? lo : hi, rcv;
                        rcv.x = dppf(snd.x, ctl); rcv.y = dppf(snd.y, ctl); rcv.z = dppf(snd.z, ctl); rcv.w = dppf(snd.w, ctl);
                        if (odd) lo = rcv; else hi = rcv;
                    };
                    xchg(B[0], B[1], o1, X1{}); xchg(B[2], B[3], o1, X1{});
                    xchg(B[0], B[2], o2, X2{}); xchg(B[1], B[3], o2, X2{});
                };
                auto sampler = [&](float sgx, float sgy, float (&dst)[8][4]) __attribute__((always_inline)) {
                    int x0, y0;
                    tap_corner(sgx, sgy, H, W, x0, y0);
                    const int xb = min(max(x0, 0), W - 2);
                    const bool i0 = x0 - xb >= 1, i1 = x0 + 1 - xb >= 1;          // pixel of the segment that is the west / east tap
#pragma unroll
                    for (int rowk = 0; rowk < 2; ++rowk) {
                        float4 B[4];
                        row_segment(min(max(y0 + rowk, 0), H - 1), xb, B);
                        const float4 w0 = i0 ? B[2] : B[0], w1 = i0 ? B[3] : B[1], e0 = i1 ? B[2] : B[0], e1 = i1 ? B[3] : B[1];
                        const int kw = rowk * 2, ke = rowk * 2 + 1;
                        dst[0][kw] = w0.x; dst[1][kw] = w0.y; dst[2][kw] = w0.z; dst[3][kw] = w0.w;
                        dst[4][kw] = w1.x; dst[5][kw] = w1.y; dst[6][kw] = w1.z; dst[7][kw] = w1.w;
                        dst[0][ke] = e0.x; dst[1][ke] = e0.y; dst[2][ke] = e0.z; dst[3][ke] = e0.w;
                        dst[4][ke] = e1.x; dst[5][ke] = e1.y; dst[6][ke] = e1.z; dst[7][ke] = e1.w;
                    }
                };
                sampler(gx, gy, a);          // (taps outside the frame carry weight 0, whatever in-range pixel stands in for them)
                sampler(wgx, wgy, b);
            } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 a0 = og[om[k] * 2], a1 = og[om[k] * 2 + 1];
                const float4 b0 = og[of[k] * 2], b1 = og[of[k] * 2 + 1];
                a[0][k] = a0.x; a[1][k] = a0.y; a[2][k] = a0.z; a[3][k] = a0.w;
                a[4][k] = a1.x; a[5][k] = a1.y; a[6][k] = a1.z; a[7][k] = a1.w;
                b[0][k] = b0.x; b[1][k] = b0.y; b[2][k] = b0.z; b[3][k] = b0.w;
                b[4][k] = b1.x; b[5][k] = b1.y; b[6][k] = b1.z; b[7][k] = b1.w;
            }
            }
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                m[c] = 0.f; r[c] = 1.f;
                if (x_mean != nullptr) { m[c] = x_mean[n * C + c0 + c]; r[c] = x_rstd[n * C + c0 + c]; }
            }
        } else {
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            const float* plane = x + ((long long)n * C + c0 + c) * HW;
            m[c] = 0.f; r[c] = 1.f;
            if (x_mean != nullptr) { m[c] = x_mean[n * C + c0 + c]; r[c] = x_rstd[n * C + c0 + c]; }
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[c][k] = plane[om[k]]; b[c][k] = plane[of[k]]; }
        }
        }
        // (an out-of-range tap reads element 0 with weight 0: the same sum as skipping it, no select per value)
        // Channel pairs as 2-vectors: multiplications and additions become v_pk_* instructions (two values each, the same
        // unfused arithmetic per value; 1135 -> 1050 vector instructions in the kernel).  The waves issue instructions 77 %
        // of the time (profiles/r03zz_pmc_wait.md: SQ_ACTIVE_INST_ANY), yet this did not move the kernel (156 us at 256^2).
        typedef float f2 __attribute__((ext_vector_type(2)));
        auto act2 = [](f2 t) -> f2 {
            if (ACT == 1) return f2{fmaxf(t.x, 0.f), fmaxf(t.y, 0.f)};
            if (ACT == 2) { const f2 u = t * 0.2f; return f2{t.x > 0.f ? t.x : u.x, t.y > 0.f ? t.y : u.y}; }
            return t;
        };
#pragma unroll
        for (int c = 0; c < 8; c += 2) {
            const f2 mm = {m[c], m[c + 1]}, rr = {r[c], r[c + 1]};
            f2 s1 = act2((f2{a[c][0], a[c + 1][0]} - mm) * rr) * wm[0];
            f2 s2 = act2((f2{b[c][0], b[c + 1][0]} - mm) * rr) * wf[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) {
                s1 += act2((f2{a[c][k], a[c + 1][k]} - mm) * rr) * wm[k];
                s2 += act2((f2{b[c][k], b[c + 1][k]} - mm) * rr) * wf[k];
            }
            v1[c] = s1.x; v1[c + 1] = s1.y;
            v2[c] = keep ? s2.x : -1.f; v2[c + 1] = keep ? s2.y : -1.f;
        }
        if (out != nullptr) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                out[((long long)n * 2 * C + c0 + c) * HW + pix] = v1[c];
                out[((long long)n * 2 * C + C + c0 + c) * HW + pix] = v2[c];
            }
        }
        if (xs != nullptr && s2d) {
            store_split_slot_s2d(xs, n, (2 * C) >> 3, by, H, W, oy, ox, v1);
            store_split_slot_s2d(xs, n, (2 * C) >> 3, (C >> 3) + by, H, W, oy, ox, v2);
        } else if (xs != nullptr) {
            store_split_slot(xs, n, (2 * C) >> 3, by, HW, pix, v1);
            store_split_slot(xs, n, (2 * C) >> 3, (C >> 3) + by, HW, pix, v2);
        }
        return;
    }
    for (int c = c0; c < c1; ++c) {
        const float* plane = x + ((long long)n * C + c) * HW;
        float m = 0.f, r = 1.f;
        if (x_mean != nullptr) { m = x_mean[n * C + c]; r = x_rstd[n * C + c]; }
        float v1 = 0.f, v2 = 0.f;
#pragma unroll
        for (int k = 0; k < 4; ++k) v1 += tap_val(plane, tm.off[k], m, r, x_act) * tm.w[k];
        if (keep) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v2 += tap_val(plane, tf.off[k], m, r, x_act) * tf.w[k];
        } else {
            v2 = -1.f;
        }
        out[((long long)n * 2 * C + c) * HW + pix] = v1;
        out[((long long)n * 2 * C + C + c) * HW + pix] = v2;
    }
}

// backward w.r.t. x only (motion / flow / mask carry no gradient: geomgm_ifw_fore_model.py:443-505 feeds
// them as data).  dx must be zero on entry; taps are scattered with fp32 atomics (L2).
__global__ __launch_bounds__(256) void warp_concat_bwd_kernel(const float* __restrict__ gout,
                                                              const float* __restrict__ motion,
                                                              const float* __restrict__ flow,
                                                              const float* __restrict__ ifmask, float* __restrict__ dx,
                                                              int C, int H, int W, int S, float flow_scale) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int n = blockIdx.z;
    const int oy = pix / W, ox = pix - oy * W;
    const long long SS = (long long)S * S;
    float gx, gy, fx, fy, mk;
    if (H == S && W == S) {
        const float2 g = reinterpret_cast<const float2*>(motion)[n * SS + pix];
        gx = g.x; gy = g.y;
        fx = flow[(n * 2 + 0) * SS + pix] * flow_scale;
        fy = flow[(n * 2 + 1) * SS + pix] * flow_scale;
        mk = ifmask[n * SS + pix];
    } else {
        const Lerp ly = make_lerp(oy, S, H), lx = make_lerp(ox, S, W);
        const int o00 = ly.i0 * S + lx.i0, o01 = ly.i0 * S + lx.i1, o10 = ly.i1 * S + lx.i0, o11 = ly.i1 * S + lx.i1;
        const float2* mo = reinterpret_cast<const float2*>(motion) + n * SS;
        const float2 a = mo[o00], b = mo[o01], c = mo[o10], d = mo[o11];
#ifdef APAMD_HZ_NOP
        asm volatile("s_nop 4" ::: "memory");       // co-residency lab, victim-side variant: idle states behind the four tap loads
#endif
        gx = bilerp(a.x, b.x, c.x, d.x, ly, lx);
        gy = bilerp(a.y, b.y, c.y, d.y, ly, lx);
        const float* f0 = flow + (n * 2 + 0) * SS;
        const float* f1 = flow + (n * 2 + 1) * SS;
        fx = bilerp(f0[o00] * flow_scale, f0[o01] * flow_scale, f0[o10] * flow_scale, f0[o11] * flow_scale, ly, lx);
        fy = bilerp(f1[o00] * flow_scale, f1[o01] * flow_scale, f1[o10] * flow_scale, f1[o11] * flow_scale, ly, lx);
        const float* mp = ifmask + n * SS;
        mk = bilerp(mp[o00], mp[o01], mp[o10], mp[o11], ly, lx);
    }
    const Taps tm = make_taps(gx, gy, H, W);
    const float wgx = 2.0f * ((float)ox + fx) / (float)(W - 1 > 1 ? W - 1 : 1) - 1.0f;
    const float wgy = 2.0f * ((float)oy + fy) / (float)(H - 1 > 1 ? H - 1 : 1) - 1.0f;
    const Taps tf = make_taps(wgx, wgy, H, W);
    const bool keep = mk > 0.5f;
    const int HW = H * W;
    const int c0 = blockIdx.y * kWarpCG;
    const int c1 = c0 + kWarpCG < C ? c0 + kWarpCG : C;
    for (int c = c0; c < c1; ++c) {
        float* plane = dx + ((long long)n * C + c) * HW;
        const float g1 = gout[((long long)n * 2 * C + c) * HW + pix];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (tm.off[k] >= 0) atomicAdd(plane + tm.off[k], g1 * tm.w[k]);
        if (keep) {
            const float g2 = gout[((long long)n * 2 * C + C + c) * HW + pix];
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (tf.off[k] >= 0) atomicAdd(plane + tf.off[k], g2 * tf.w[k]);
        }
    }
}

// ---- tiled form of the same scatter.  Both sampling maps are smooth, so the taps of a 32 x 16 output tile land in a
// compact window of the input; the tile accumulates them in LDS for 8 channels at once and flushes every window element
// with ONE global atomic -- ~5x fewer L2 atomics than a tap-by-tap scatter, on contiguous rows.  Taps beyond the window
// (strong magnification, noise) scatter directly, as warp_concat_bwd_kernel does.
struct PixelTaps {
    int x0[2], y0[2];       // north-west tap of the motion sample / the flow sample
    float w[2][4];          // nw, ne, sw, se weights; 0 for out-of-range taps
    bool live;              // pixel inside the image
    bool keep;              // mask > 0.5: the flow branch carries gradient
};

__device__ __forceinline__ void taps_xy(float gx, float gy, int H, int W, int& x0, int& y0, float (&w)[4]) {
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f;
    float iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);
    iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);
    const float fx = floorf(ix), fy = floorf(iy);
    x0 = (int)fx; y0 = (int)fy;
    const float ex = fx + 1.f, ey = fy + 1.f;
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x0 + 1 >= 0 && x0 + 1 < W;
    const bool yin0 = y0 >= 0 && y0 < H, yin1 = y0 + 1 >= 0 && y0 + 1 < H;
    w[0] = (xin0 && yin0) ? (ex - ix) * (ey - iy) : 0.f;
    w[1] = (xin1 && yin0) ? (ix - fx) * (ey - iy) : 0.f;
    w[2] = (xin0 && yin1) ? (ex - ix) * (iy - fy) : 0.f;
    w[3] = (xin1 && yin1) ? (ix - fx) * (iy - fy) : 0.f;
}

__device__ __forceinline__ PixelTaps pixel_taps(int n, int oy, int ox, const float* __restrict__ motion,
                                                const float* __restrict__ flow, const float* __restrict__ ifmask,
                                                int H, int W, int S, float flow_scale) {
    PixelTaps t;
    t.live = oy < H && ox < W;
    if (!t.live) {
        t.keep = false;
        t.x0[0] = t.x0[1] = t.y0[0] = t.y0[1] = 0;
#pragma unroll
        for (int k = 0; k < 4; ++k) t.w[0][k] = t.w[1][k] = 0.f;
        return t;
    }
    const long long SS = (long long)S * S;
    float gx, gy, fx, fy, mk;
    if (H == S && W == S) {
        const int pix = oy * W + ox;
        const float2 g = reinterpret_cast<const float2*>(motion)[n * SS + pix];
        gx = g.x; gy = g.y;
        fx = flow[(n * 2 + 0) * SS + pix] * flow_scale;
        fy = flow[(n * 2 + 1) * SS + pix] * flow_scale;
        mk = ifmask[n * SS + pix];
    } else {
        const Lerp ly = make_lerp(oy, S, H), lx = make_lerp(ox, S, W);
        const int o00 = ly.i0 * S + lx.i0, o01 = ly.i0 * S + lx.i1, o10 = ly.i1 * S + lx.i0, o11 = ly.i1 * S + lx.i1;
        const float2* mo = reinterpret_cast<const float2*>(motion) + n * SS;
        const float2 a = mo[o00], b = mo[o01], c = mo[o10], d = mo[o11];
#ifdef APAMD_HZ_NOP
        asm volatile("s_nop 4" ::: "memory");       // co-residency lab, victim-side variant: idle states behind the four tap loads
#endif
        gx = bilerp(a.x, b.x, c.x, d.x, ly, lx);
        gy = bilerp(a.y, b.y, c.y, d.y, ly, lx);
        const float* f0 = flow + (n * 2 + 0) * SS;
        const float* f1 = flow + (n * 2 + 1) * SS;
        fx = bilerp(f0[o00] * flow_scale, f0[o01] * flow_scale, f0[o10] * flow_scale, f0[o11] * flow_scale, ly, lx);
        fy = bilerp(f1[o00] * flow_scale, f1[o01] * flow_scale, f1[o10] * flow_scale, f1[o11] * flow_scale, ly, lx);
        const float* mp = ifmask + n * SS;
        mk = bilerp(mp[o00], mp[o01], mp[o10], mp[o11], ly, lx);
    }
    taps_xy(gx, gy, H, W, t.x0[0], t.y0[0], t.w[0]);
    const float wgx = 2.0f * ((float)ox + fx) / (float)(W - 1 > 1 ? W - 1 : 1) - 1.0f;
    const float wgy = 2.0f * ((float)oy + fy) / (float)(H - 1 > 1 ? H - 1 : 1) - 1.0f;
    taps_xy(wgx, wgy, H, W, t.x0[1], t.y0[1], t.w[1]);
    t.keep = mk > 0.5f;
    return t;
}

constexpr int kBwdTileW = 32, kBwdTileH = 16, kBwdPix = kBwdTileW * kBwdTileH;
constexpr int kBwdWin = 1344;       // window floats per channel (8 channels: 42 KiB; two workgroups per CU with the rest)
constexpr int kBwdWinW = 48;        // width of a clipped window

// ds_add_f32 retires about ONE LANE per clock per CU on gfx950 (measured: 537 M lane-atomics of the 256 x 256 level took
// 1.3 ms of the launch's 2.2 ms), against 32 lanes per clock for plain LDS reads / writes.  So the accumulation is done
// with plain read-modify-writes: each of the four waves OWNS two channel windows (no other wave touches them) and walks
// all 512 pixels of the tile, one tap (branch, corner) of all 512 pixels at a time.  Taps of one batch that hit the same
// window element are found with a claim word (write a unique id per (lane, pixel), read it back: exactly one writer
// wins): the winners' addresses are pairwise distinct, so their reads, adds and writes are issued as batches (two LDS
// round trips per 8 taps instead of one per tap); the rare losers use ds_add_f32 right after, in program order of the
// same wave.  (Measured on the 256 x 256 level, B = 32, smooth maps: 2.14 ms with ds_add_f32, 0.55 ms this way.)  The taps of the tile are computed once by the whole
// workgroup and handed to the waves through LDS.  When the taps' bounding box exceeds the window (noisy or magnifying
// maps) the window is the box's centre part and the taps outside it go to global memory one by one.
// Round 5: (a) the claims are resolved once per workgroup (wave w: batches 2w, 2w + 1; the result per tap in s_q) instead of by
// every wave, and the two channels of a wave are interleaved so that a tap is ONE 8-byte read-modify-write: 56 -> 32 LDS
// instructions per batch and wave; 1743 -> ~1630 us for the three levels of the train step.  (b) Tried and dropped: one window
// per sampling map when the box around both does not fit (the motion map may sample far from the tile): on the bench's maps --
// identity + WHITE noise of 2.5 px per pixel, boxes of 42 x 28 per map -- two half windows clip more taps than one whole one
// (256 x 256 level: 0.96 -> 1.31 ms); on smooth maps both forms take the single-window path (0.54 ms).  What the bench measures
// is the noise: a tile's 4096 taps collide and overflow the window whatever its shape.
// grid: (tiles_x * tiles_y, ceil(C/8), N)
__global__ __launch_bounds__(256) void warp_concat_bwd_tiled_kernel(const float* __restrict__ gout,
                                                                    const float* __restrict__ motion,
                                                                    const float* __restrict__ flow,
                                                                    const float* __restrict__ ifmask,
                                                                    float* __restrict__ dx, int C, int H, int W, int S,
                                                                    float flow_scale, int tiles_x) {
    __shared__ int s_box[4];
    __shared__ float2 win2[kWarpCG / 2][kBwdWin + 64];   // channel PAIRS interleaved (+ 64: a trash element per lane)
    __shared__ int s_xy[2][kBwdPix];                 // (y0 + 8) << 16 | (x0 + 8) of the north-west tap; bit 31: branch live
    __shared__ float s_w[2][4][kBwdPix];             // tap weights: > 0 inside the window, < 0 (negated) outside, 0 dead
    __shared__ unsigned short s_q[8][kBwdPix];       // per batch: the window element a tap adds to (trash: dead, outside, or lost its claim)
    // logical block (tile fastest, then channel group, then image) contiguous per XCD, as in the forward kernel: the tiles whose
    // windows overlap -- and whose flushes add into the same lines of dx -- run on one XCD
    int bx, by, bz;
    {
        const unsigned L = xcd_logical_block(gridDim.x * gridDim.y * gridDim.z,
                                             blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
        bx = L % gridDim.x;
        const unsigned t = L / gridDim.x;
        by = t % gridDim.y;
        bz = t / gridDim.y;
    }
    const int tid = threadIdx.x, n = bz, wave = tid >> 6, lane = tid & 63;
    const int ty0 = (bx / tiles_x) * kBwdTileH, tx0 = (bx % tiles_x) * kBwdTileW;
    if (tid == 0) { s_box[0] = W; s_box[1] = H; s_box[2] = -1; s_box[3] = -1; }
    __syncthreads();
    PixelTaps pt[2];
    {
        int xmin = W, ymin = H, xmax = -1, ymax = -1;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int p = tid + 256 * j;
            pt[j] = pixel_taps(n, ty0 + (p >> 5), tx0 + (p & 31), motion, flow, ifmask, H, W, S, flow_scale);
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const bool on = pt[j].live && (b == 0 || pt[j].keep);
                float ws = 0.f;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    if (!on) pt[j].w[b][k] = 0.f;
                    ws += pt[j].w[b][k];
                }
                if (!(ws != 0.f)) continue;                              // every tap out of range
                const int xa = pt[j].x0[b] < 0 ? 0 : pt[j].x0[b], xb = pt[j].x0[b] + 1 >= W ? W - 1 : pt[j].x0[b] + 1;
                const int ya = pt[j].y0[b] < 0 ? 0 : pt[j].y0[b], yb = pt[j].y0[b] + 1 >= H ? H - 1 : pt[j].y0[b] + 1;
                xmin = xa < xmin ? xa : xmin; xmax = xb > xmax ? xb : xmax;
                ymin = ya < ymin ? ya : ymin; ymax = yb > ymax ? yb : ymax;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            xmin = min(xmin, __shfl_xor(xmin, m)); ymin = min(ymin, __shfl_xor(ymin, m));
            xmax = max(xmax, __shfl_xor(xmax, m)); ymax = max(ymax, __shfl_xor(ymax, m));
        }
        if (lane == 0 && xmax >= 0) {
            atomicMin(&s_box[0], xmin); atomicMin(&s_box[1], ymin);
            atomicMax(&s_box[2], xmax); atomicMax(&s_box[3], ymax);
        }
    }
    __syncthreads();
    if (s_box[2] < 0) return;                                            // the tile scatters nothing
    int bx0 = s_box[0], by0 = s_box[1];
    int bw = s_box[2] - bx0 + 1, bh = s_box[3] - by0 + 1;
    if (bw * bh > kBwdWin) {                                             // keep the centre of the box
        const int nw = bw < kBwdWinW ? bw : kBwdWinW;
        const int nh = bh < kBwdWin / nw ? bh : kBwdWin / nw;
        bx0 += (bw - nw) / 2; by0 += (bh - nh) / 2;
        bw = nw; bh = nh;
    }
    const int c0 = by * kWarpCG;
    const int nc = c0 + kWarpCG <= C ? kWarpCG : C - c0;
    const int HW = H * W, area = bw * bh;
    // the taps in window terms, so that the walk below spends a handful of VALU instructions per tap (it is VALU-bound)
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int p = tid + 256 * j;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int x0 = pt[j].x0[b] - bx0, y0 = pt[j].y0[b] - by0;
            bool any = false;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float wk = pt[j].w[b][k];
                const bool inside = (unsigned)(x0 + (k & 1)) < (unsigned)bw && (unsigned)(y0 + (k >> 1)) < (unsigned)bh;
                s_w[b][k][p] = inside ? wk : -wk;
                any |= wk != 0.f;
            }
            s_xy[b][p] = ((pt[j].y0[b] + 8) << 16) | (pt[j].x0[b] + 8) | (any ? (int)0x80000000 : 0);
        }
    }
    __syncthreads();
    // ---- the claims, ONCE per workgroup: every wave walks the 512 pixels in the same (lane, i) order, so which taps of a batch
    // collide is the same for all of them.  Wave w resolves batches 2w and 2w + 1 and leaves, per tap, the element it may add to
    // with a plain read-modify-write (s_q).  (The claim words live in the window's memory, which is cleared afterwards.)
    constexpr int NI = kBwdPix / 64;
    const int trash = kBwdWin + lane;        // window element nobody reads: where dead taps and claim losers write
    {
        typedef __attribute__((address_space(3))) volatile unsigned short lds_vu16;
        lds_vu16* claim = (lds_vu16*)(reinterpret_cast<unsigned short*>(&win2[0][0]) + wave * (kBwdWin + 64));
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int bk = 2 * wave + h, b = bk >> 2, k = bk & 3;
            const int off = (k & 1) + (k >> 1) * bw;
            int a[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const int p = i * 64 + lane;
                const int v = s_xy[b][p];
                const int a0 = (((v >> 16) & 0x7fff) - 8 - by0) * bw + (v & 0xffff) - 8 - bx0;
                a[i] = s_w[b][k][p] > 0.f ? a0 + off : trash;
            }
#pragma unroll
            for (int i = 0; i < NI; ++i) claim[a[i]] = (unsigned short)(lane * NI + i);
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const bool own = claim[a[i]] == (unsigned short)(lane * NI + i);   // (a lane always owns its trash element)
                s_q[bk][i * 64 + lane] = (unsigned short)(own ? a[i] : trash);
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < kWarpCG / 2; ++c)
        for (int i = tid; i < area; i += 256) win2[c][i] = make_float2(0.f, 0.f);
    __syncthreads();
    // ---- this wave's two channels over the whole tile: one 8-byte read-modify-write per tap (ds_read_b64 costs what
    // ds_read_b32 does, ds_write_b64 6 LDS cycles against 2 x 4: MI355X_MICROARCH, LDS table)
    const int ca = 2 * wave, cb = 2 * wave + 1;
    if (ca < nc) {
        const bool two = cb < nc;
        // (address-space-3 pointers: through a generic volatile pointer these accesses become FLAT loads / stores)
        typedef __attribute__((address_space(3))) volatile unsigned long long lds_vu64;     // (one 8-byte access per element)
        lds_vu64* wab = (lds_vu64*)&win2[wave][0];
        const float* ga = gout + ((long long)n * 2 * C + c0 + ca) * HW;
        float* pa = dx + ((long long)n * C + c0 + ca) * HW;
        // every gradient value of the wave's walk is requested up front (8 pixels x 2 branches x 2 channels per lane)
        int a0[NI][2];
        float g0[NI][2], g1[NI][2];
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int p = i * 64 + lane;
            const int pix = (ty0 + (p >> 5)) * W + tx0 + (p & 31);       // (only dereferenced when the branch is live)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const int v = s_xy[b][p];
                a0[i][b] = (((v >> 16) & 0x7fff) - 8 - by0) * bw + (v & 0xffff) - 8 - bx0;
                g0[i][b] = g1[i][b] = 0.f;
                if (v < 0) {                                             // bit 31: some tap of the branch is in range
                    g0[i][b] = ga[(long long)b * C * HW + pix];
                    if (two) g1[i][b] = ga[((long long)b * C + 1) * HW + pix];
                }
            }
        }
        // one batch = tap (b, k) of the lane's 8 pixels (8 different row pairs of the tile: for a locally injective map
        // the 512 addresses of a batch are distinct; taps k and k' of NEIGHBOURING pixels coincide all the time, which
        // is why a batch never mixes them)
#pragma unroll
        for (int bk = 0; bk < 8; ++bk) {
            const int b = bk >> 2, k = bk & 3;
            const int off = (k & 1) + (k >> 1) * bw;
            int q[NI];
            float va[NI], vb[NI];
            float wmin = 0.f;
            bool lost = false;
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float wk = s_w[b][k][i * 64 + lane];
                q[i] = s_q[bk][i * 64 + lane];
                va[i] = g0[i][b] * wk;
                vb[i] = g1[i][b] * wk;
                wmin = fminf(wmin, wk);
                lost |= wk > 0.f && q[i] == trash;                       // inside the window, but another tap of the batch owns the element
            }
            if (wmin < 0.f) {                                            // taps beyond a clipped window: one by one
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (s_w[b][k][i * 64 + lane] < 0.f) {
                        const int v = s_xy[b][i * 64 + lane];
                        const int x = (v & 0xffff) - 8 + (k & 1), y = ((v >> 16) & 0x7fff) - 8 + (k >> 1);
                        atomicAdd(pa + y * W + x, -va[i]);
                        if (two) atomicAdd(pa + HW + y * W + x, -vb[i]);
                    }
            }
            unsigned long long r[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) r[i] = wab[q[i]];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                const float sa = __uint_as_float((unsigned)r[i]) + va[i], sb = __uint_as_float((unsigned)(r[i] >> 32)) + vb[i];
                wab[q[i]] = (unsigned long long)__float_as_uint(sa) | ((unsigned long long)__float_as_uint(sb) << 32);
            }
            if (lost) {
#pragma unroll
                for (int i = 0; i < NI; ++i)
                    if (s_w[b][k][i * 64 + lane] > 0.f && q[i] == trash) {
                        atomicAdd(&win2[wave][a0[i][b] + off].x, va[i]);
                        if (two) atomicAdd(&win2[wave][a0[i][b] + off].y, vb[i]);
                    }
            }
        }
    }
    __syncthreads();
    // (no integer divisions in this loop: at ~40 VALU instructions each they cost more than the stores)
    const float inv_bw = 1.0f / (float)bw;
    for (int r = tid; r < area; r += 256) {
        const int y = (int)(((float)r + 0.5f) * inv_bw);                 // == r / bw for r < 2^20
        float* q = dx + ((long long)n * C + c0) * HW + (by0 + y) * W + bx0 + (r - y * bw);
        for (int c = 0; c < nc; c += 2) {
            const float2 v = win2[c >> 1][r];
            if (v.x != 0.f) atomicAdd(q + (long long)c * HW, v.x);
            if (c + 1 < nc && v.y != 0.f) atomicAdd(q + (long long)(c + 1) * HW, v.y);
        }
    }
}

// ---- image-level helpers of the streaming-inference model (geomcgt_ifw_test_model.py:282-285, 294):
// F.interpolate(mode='bilinear', align_corners=False) and F.grid_sample(bilinear, zeros padding).
// grid: (ceil(OH*OW/256), N*C)
__global__ __launch_bounds__(256) void resize_bilinear_kernel(const float* __restrict__ x, float* __restrict__ y, int H,
                                                              int W, int OH, int OW) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= OH * OW) return;
    const int oy = pix / OW, ox = pix - oy * OW;
    const float sh = (float)H / (float)OH, sw = (float)W / (float)OW;      // area_pixel_compute_scale
    float fy = sh * ((float)oy + 0.5f) - 0.5f, fx = sw * ((float)ox + 0.5f) - 0.5f;
    fy = fy < 0.f ? 0.f : fy;
    fx = fx < 0.f ? 0.f : fx;
    int y0 = (int)fy, x0 = (int)fx;
    y0 = y0 > H - 1 ? H - 1 : y0;
    x0 = x0 > W - 1 ? W - 1 : x0;
    const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
    const float ly1 = fy - (float)y0, lx1 = fx - (float)x0, ly0 = 1.f - ly1, lx0 = 1.f - lx1;
    const float* p = x + (long long)blockIdx.y * H * W;
    y[(long long)blockIdx.y * OH * OW + pix] =
        ly0 * (lx0 * p[y0 * W + x0] + lx1 * p[y0 * W + x1]) + ly1 * (lx0 * p[y1 * W + x0] + lx1 * p[y1 * W + x1]);
}

// grid: (ceil(OH*OW/256), C, N); sampling grid (N, OH, OW, 2) in [-1, 1]
__global__ __launch_bounds__(256) void grid_sample_kernel(const float* __restrict__ x, const float* __restrict__ grid,
                                                          float* __restrict__ y, int C, int H, int W, int OH, int OW,
                                                          int align_corners) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= OH * OW) return;
    const int c = blockIdx.y, n = blockIdx.z;
    const float2 g = reinterpret_cast<const float2*>(grid)[(long long)n * OH * OW + pix];
    float ix, iy;
    if (align_corners) {
        ix = (g.x + 1.f) / 2.f * (float)(W - 1);
        iy = (g.y + 1.f) / 2.f * (float)(H - 1);
    } else {
        ix = ((g.x + 1.f) * (float)W - 1.f) / 2.f;
        iy = ((g.y + 1.f) * (float)H - 1.f) / 2.f;
    }
    ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);
    iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ex = fx + 1.f, ey = fy + 1.f;
    const float* p = x + ((long long)n * C + c) * H * W;
    auto at = [&](int yy, int xx) { return (yy >= 0 && yy < H && xx >= 0 && xx < W) ? p[yy * W + xx] : 0.f; };
    float v = at(y0, x0) * ((ex - ix) * (ey - iy));
    v += at(y0, x1) * ((ix - fx) * (ey - iy));
    v += at(y1, x0) * ((ex - ix) * (iy - fy));
    v += at(y1, x1) * ((ix - fx) * (iy - fy));
    y[((long long)n * C + c) * OH * OW + pix] = v;
}

// ---- motion grid of the data layer (Module2/data/umlvd_ifw_dataset.py:60-74, umlvdfw_test_dataset.py:67-81:
// scipy.interpolate.griddata(destination, source, grid, method='linear')): piecewise-linear interpolation of the
// source landmark positions over the Delaunay triangulation of the destination landmarks (+ 8 border points), evaluated
// at every pixel and normalised to grid_sample coordinates.  The triangulation (76 points) comes from the host; the
// kernel rasterises it: one lane per pixel, triangles in LDS, the triangle that contains the pixel best (largest
// minimum barycentric coordinate -- on shared edges both candidates give the same value) is interpolated.
// pts: [N][P][2] destination (row, col); val: [N][P][2] source (row, col); tri: [N][T][3] point indices (-1: unused)
// out: [N][S][S][2] = (col, row) / ((S-1)/2) - 1.   grid: (ceil(S*S/256), N)
__global__ __launch_bounds__(256) void motion_grid_kernel(const float* __restrict__ pts, const float* __restrict__ val,
                                                          const int* __restrict__ tri, int P, int T, int S,
                                                          float* __restrict__ out) {
    extern __shared__ float tsm[];                       // per triangle: 3 x (row, col) destination, 3 x (row, col) source
    const int n = blockIdx.y;
    for (int i = threadIdx.x; i < T; i += 256) {
        const int a = tri[((long long)n * T + i) * 3], b = tri[((long long)n * T + i) * 3 + 1],
                  c = tri[((long long)n * T + i) * 3 + 2];
        float* t = tsm + i * 12;
        if (a < 0 || b < 0 || c < 0 || a >= P || b >= P || c >= P) {
            for (int k = 0; k < 12; ++k) t[k] = 0.f;     // degenerate: never selected (zero area)
        } else {
            const int idx[3] = {a, b, c};
            for (int k = 0; k < 3; ++k) {
                t[k * 2] = pts[((long long)n * P + idx[k]) * 2];
                t[k * 2 + 1] = pts[((long long)n * P + idx[k]) * 2 + 1];
                t[6 + k * 2] = val[((long long)n * P + idx[k]) * 2];
                t[6 + k * 2 + 1] = val[((long long)n * P + idx[k]) * 2 + 1];
            }
        }
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= S * S) return;
    const float pr = (float)(pix / S), pc = (float)(pix % S);
    float best = -1e30f, vr = 0.f, vc = 0.f;
    for (int i = 0; i < T; ++i) {
        const float* t = tsm + i * 12;
        const float r0 = t[0], c0 = t[1], r1 = t[2], c1 = t[3], r2 = t[4], c2 = t[5];
        const float det = (r1 - r0) * (c2 - c0) - (r2 - r0) * (c1 - c0);
        if (det == 0.f) continue;
        const float l1 = ((pr - r0) * (c2 - c0) - (r2 - r0) * (pc - c0)) / det;
        const float l2 = ((r1 - r0) * (pc - c0) - (pr - r0) * (c1 - c0)) / det;
        const float l0 = 1.f - l1 - l2;
        const float m = fminf(l0, fminf(l1, l2));
        if (m > best) {
            best = m;
            vr = l0 * t[6] + l1 * t[8] + l2 * t[10];
            vc = l0 * t[7] + l1 * t[9] + l2 * t[11];
        }
    }
    const float half = (float)(S - 1) / 2.f;              // 127.5 for 256
    float2 o;
    o.x = vc / half - 1.f;
    o.y = vr / half - 1.f;
    reinterpret_cast<float2*>(out)[(long long)n * S * S + pix] = o;
}

}  // namespace apamd

using namespace apamd;

extern "C" int ap_motion_grid(const float* pts, const float* val, const int32_t* tri, int32_t N, int32_t P, int32_t T,
                              int32_t S, float* out, ap_stream_t stream) {
    if (!pts || !val || !tri || !out) return fail(AP_ERR_INVALID, "motion_grid: null pointer");
    if (N < 1 || N > 65535 || P < 3 || T < 1 || S < 2) return fail(AP_ERR_INVALID, "motion_grid: bad sizes");
    const size_t lds = (size_t)T * 12 * sizeof(float);
    if (lds > 60 * 1024) return fail(AP_ERR_UNSUPPORTED, "motion_grid: %d triangles do not fit the LDS table", T);
    hipLaunchKernelGGL(motion_grid_kernel, dim3((S * S + 255) / 256, N), dim3(256), lds, (hipStream_t)stream, pts, val, tri,
                       P, T, S, out);
    return check_launch("motion_grid_kernel");
}

extern "C" int ap_resize_bilinear(const float* x, int32_t NC, int32_t H, int32_t W, int32_t OH, int32_t OW, float* y,
                                  ap_stream_t stream) {
    if (!x || !y) return fail(AP_ERR_INVALID, "resize_bilinear: null pointer");
    if (NC < 1 || NC > 65535 || H < 1 || W < 1 || OH < 1 || OW < 1) return fail(AP_ERR_INVALID, "resize_bilinear: bad sizes");
    hipLaunchKernelGGL(resize_bilinear_kernel, dim3((OH * OW + 255) / 256, NC), dim3(256), 0, (hipStream_t)stream, x, y, H,
                       W, OH, OW);
    return check_launch("resize_bilinear_kernel");
}

extern "C" int ap_grid_sample(const float* x, const float* grid, int32_t N, int32_t C, int32_t H, int32_t W, int32_t OH,
                              int32_t OW, int32_t align_corners, float* y, ap_stream_t stream) {
    if (!x || !grid || !y) return fail(AP_ERR_INVALID, "grid_sample: null pointer");
    if (N < 1 || N > 65535 || C < 1 || C > 65535 || H < 1 || W < 1 || OH < 1 || OW < 1)
        return fail(AP_ERR_INVALID, "grid_sample: bad sizes");
    hipLaunchKernelGGL(grid_sample_kernel, dim3((OH * OW + 255) / 256, C, N), dim3(256), 0, (hipStream_t)stream, x, grid,
                       y, C, H, W, OH, OW, align_corners);
    return check_launch("grid_sample_kernel");
}

extern "C" int ap_warp_concat_bwd(const float* gout, const float* motion, const float* flow, const float* ifmask,
                                  float* dx, int32_t N, int32_t C, int32_t H, int32_t W, int32_t S, float flow_scale,
                                  ap_stream_t stream) {
    if (!gout || !motion || !flow || !ifmask || !dx) return fail(AP_ERR_INVALID, "warp_concat_bwd: null pointer");
    if (N < 1 || C < 1 || H < 1 || W < 1 || S < 1 || N > 65535) return fail(AP_ERR_INVALID, "warp_concat_bwd: bad sizes");
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)N * C * H * W * sizeof(float), (hipStream_t)stream);
    if (e != hipSuccess) return fail(AP_ERR_LAUNCH, "warp_concat_bwd memset: %s", hipGetErrorString(e));
    const char* plain = getenv("APAMD_WARP_BWD_PLAIN");
    if (plain && atoi(plain)) {   // tap-by-tap scatter (kept for comparison)
        dim3 grid((H * W + 255) / 256, (C + kWarpCG - 1) / kWarpCG, N);
        hipLaunchKernelGGL(warp_concat_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, gout, motion, flow, ifmask,
                           dx, C, H, W, S, flow_scale);
        return check_launch("warp_concat_bwd_kernel");
    }
    const int tiles_x = (W + kBwdTileW - 1) / kBwdTileW, tiles_y = (H + kBwdTileH - 1) / kBwdTileH;
    dim3 grid(tiles_x * tiles_y, (C + kWarpCG - 1) / kWarpCG, N);
    hipLaunchKernelGGL(warp_concat_bwd_tiled_kernel, grid, dim3(256), 0, (hipStream_t)stream, gout, motion, flow, ifmask,
                       dx, C, H, W, S, flow_scale, tiles_x);
    return check_launch("warp_concat_bwd_tiled_kernel");
}

extern "C" int ap_warp_concat_fwd_split(const float* x, const float* x_mean, const float* x_rstd, int32_t x_act,
                                        const float* motion, const float* flow, const float* ifmask, float* out,
                                        void* xs, int32_t N, int32_t C, int32_t H, int32_t W, int32_t S,
                                        float flow_scale, ap_stream_t stream) {
    return ap_warp_concat_fwd_ex(x, x_mean, x_rstd, x_act, motion, flow, ifmask, out, xs, N, C, H, W, S, flow_scale, 0, stream);
}

extern "C" int ap_warp_concat_fwd_ex(const float* x, const float* x_mean, const float* x_rstd, int32_t x_act,
                                     const float* motion, const float* flow, const float* ifmask, float* out,
                                     void* xs, int32_t N, int32_t C, int32_t H, int32_t W, int32_t S,
                                     float flow_scale, int32_t flags, ap_stream_t stream) {
    const int s2d = flags & 1;
    if ((flags & 2) && (C % kWarpCG) != 0)
        return fail(AP_ERR_INVALID, "warp_concat_fwd: the channel-octet input layout needs C %% 8 == 0");
    if (s2d && (!xs || (H & 1) || (W & 1) || (C % kWarpCG) != 0))
        return fail(AP_ERR_INVALID, "warp_concat_fwd: the space-to-depth split output needs xs, an even map and C %% 8 == 0");
    if (!x || !motion || !flow || !ifmask) return fail(AP_ERR_INVALID, "warp_concat_fwd: null pointer");
    if (!out && !xs) return fail(AP_ERR_INVALID, "warp_concat_fwd: neither an fp32 nor a split output");
    if ((x_mean == nullptr) != (x_rstd == nullptr)) return fail(AP_ERR_INVALID, "warp_concat_fwd: mean/rstd mismatch");
    if (N < 1 || C < 1 || H < 1 || W < 1 || S < 1) return fail(AP_ERR_INVALID, "warp_concat_fwd: bad sizes");
    if (x_act < 0 || x_act > 2) return fail(AP_ERR_INVALID, "warp_concat_fwd: act %d", x_act);
    if (N > 65535) return fail(AP_ERR_UNSUPPORTED, "warp_concat_fwd: N too large");
    if ((C % kWarpCG) != 0 && (xs || !out))
        return fail(AP_ERR_UNSUPPORTED, "warp_concat_fwd: the split output needs C %% 8 == 0 (C=%d)", C);
    dim3 grid((H * W + 255) / 256, (C + kWarpCG - 1) / kWarpCG, N);
    // tile width (log2): 32 x 8 pixels when the map divides into them (APAMD_WARP_TILE = 0 / 4 / 5 / 6 for A/B runs)
    static const int tile_env = getenv("APAMD_WARP_TILE") ? atoi(getenv("APAMD_WARP_TILE")) : 5;
    int tw_shift = tile_env;
    if (tw_shift < 4 || tw_shift > 6 || (W & ((1 << tw_shift) - 1)) || (H & ((256 >> tw_shift) - 1))) tw_shift = 0;
    // (WPE = 4: 112 registers, no spills.  A budget for 5 waves per SIMD -- 96 registers, a dozen spills -- measured 210 us
    // against 154 us at the 256^2 level.)
    // (APAMD_WARP_WPE: victim-side variants of the co-residency lab, tools/hazard/build_victims.sh; the product builds 4)
    auto* kern = x_act == 1 ? warp_concat_kernel<1, APAMD_WARP_WPE> : (x_act == 2 ? warp_concat_kernel<2, APAMD_WARP_WPE> : warp_concat_kernel<0, APAMD_WARP_WPE>);
    // A/B switch for tools/warp_fwd_bench.py: the quad-cooperative gather (channel-octet inputs, whole 256-pixel blocks of
    // 4-aligned rows, so that every quad is four live neighbours of one row)
    const char* gather = getenv("APAMD_WARP_GATHER");           // read per call: tests flip it inside one process
    const bool quad = gather && !strcmp(gather, "quad");
    if (quad && (flags & 2) && (W & 3) == 0 && W >= 4 && ((H * W) & 255) == 0)
        kern = x_act == 1 ? warp_concat_kernel<1, APAMD_WARP_WPE, 1> : (x_act == 2 ? warp_concat_kernel<2, APAMD_WARP_WPE, 1> : warp_concat_kernel<0, APAMD_WARP_WPE, 1>);
    hipLaunchKernelGGL(kern, grid, dim3(256), 0, (hipStream_t)stream, x, x_mean, x_rstd, x_act, motion,
                       flow, ifmask, out, reinterpret_cast<uint4*>(xs), C, H, W, S, flow_scale, flags & 3, tw_shift);
    return check_launch("warp_concat_kernel");
}

extern "C" int ap_warp_concat_fwd(const float* x, const float* x_mean, const float* x_rstd, int32_t x_act,
                                  const float* motion, const float* flow, const float* ifmask, float* out, int32_t N,
                                  int32_t C, int32_t H, int32_t W, int32_t S, float flow_scale, ap_stream_t stream) {
    if (!out) return fail(AP_ERR_INVALID, "warp_concat_fwd: null pointer");
    return ap_warp_concat_fwd_split(x, x_mean, x_rstd, x_act, motion, flow, ifmask, out, nullptr, N, C, H, W, S,
                                    flow_scale, stream);
}


// ---- nn.PixelShuffle(2) (the decoder upsampling of FlowUnet_v2, Module2/intrinsic_flow_models/networks.py:693-698):
//   y[n][c][2 h + i][2 w + j] = x[n][4 c + 2 i + j][h][w]
// One lane per OUTPUT pixel pair (two x-neighbours = channels 4c + 2i, 4c + 2i + 1 at the same input pixel): 8-byte stores,
// coalesced 4-byte loads from two planes.  grid: (ceil(H * W / 256), 2 * C, N)  (y index = c * 2 + i)
namespace apamd {
__global__ __launch_bounds__(256) void pixel_shuffle2_kernel(const float* __restrict__ x, float* __restrict__ y, int C, int H, int W) {
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= H * W) return;
    const int c = blockIdx.y >> 1, i = blockIdx.y & 1, n = blockIdx.z;
    const int h = pix / W, w = pix - h * W;
    const float* px = x + ((long long)n * 4 * C + 4 * c + 2 * i) * H * W + pix;
    const float a = px[0], b = px[(long long)H * W];
    *reinterpret_cast<float2*>(y + (((long long)n * C + c) * 2 * H + 2 * h + i) * 2 * W + 2 * w) = make_float2(a, b);
}
}  // namespace apamd

extern "C" int ap_pixel_shuffle2(const float* x, int32_t N, int32_t C, int32_t H, int32_t W, float* y, ap_stream_t stream) {
    using namespace apamd;
    if (!x || !y) return fail(AP_ERR_INVALID, "pixel_shuffle2: null pointer");
    if (N < 1 || N > 65535 || C < 1 || 2 * C > 65535 || H < 1 || W < 1) return fail(AP_ERR_INVALID, "pixel_shuffle2: bad sizes");
    hipLaunchKernelGGL(pixel_shuffle2_kernel, dim3((H * W + 255) / 256, 2 * C, N), dim3(256), 0, (hipStream_t)stream, x, y, C, H, W);
    return check_launch("pixel_shuffle2_kernel");
}
