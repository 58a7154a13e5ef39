// victim_product.hip -- library-free reproducer of the co-residency hazard, bisection of the VICTIM: the product's gather kernel
// (animateportrait_amd/csrc/warp.hip, warp_concat_kernel, copied here) with pieces removed by -DKNOB=<n>, one object per n:
//   0  verbatim                                   1  without the split-bf16 output paths (xs / s2d)
//   2  ... and without the channel-octet input path 3  ... and without the partial-channel-group tail loop
//   4  ... and without the `H == S` direct-read branch   5  ... and with the 32 x 8 tile as a constant (no run-time tw_shift)
//   6  ... and without the null tests of x_mean / out
// Built by the Makefile as victim_k<n>.o; launch_victim_k<n>() has the signature of launch_victim().
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>
#ifndef KNOB
#define KNOB 0
#endif
#define CAT_(a, b) a##b
#define CAT(a, b) CAT_(a, b)
namespace CAT(vk, KNOB) {
__device__ __forceinline__ unsigned xcd_logical_block(unsigned nblk, unsigned b) {
    const unsigned q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}
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
#if KNOB < 1
    if (xs != nullptr && !s2d && bx == 0 && threadIdx.x < 4) {
        // the all-zero slot that closes every plane of the split layout (this block's two channel groups x 2 parts)
        const int CG2 = (2 * C) >> 3, HWz = H * W;
        const int part = threadIdx.x & 1, cg = (threadIdx.x >> 1) ? (C >> 3) + by : by;
        xs[((long long)(bz * 2 + part) * CG2 + cg) * (HWz + 1) + HWz] = make_uint4(0u, 0u, 0u, 0u);
    }
#endif
    // a workgroup covers a tw x (256 / tw) pixel tile when the map divides into such tiles (tw_shift > 0), else 256
    // consecutive pixels: the taps of a compact tile fall into a window the CU's L1 holds, those of a 256-pixel row
    // segment (16 noisy rows high) do not -- the gathers are then served line by line from the L2
    int pix, oy, ox;
#if KNOB >= 5
    {
        const int tiles_x = W >> 5;
        const int ty = bx / tiles_x, tx = bx - ty * tiles_x;
        oy = ty * 8 + ((int)threadIdx.x >> 5);
        ox = (tx << 5) + ((int)threadIdx.x & 31);
        pix = oy * W + ox;
    }
#else
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
#endif
    const int n = bz;
    const long long SS = (long long)S * S;

    float gx, gy, fx, fy, mk;
    if (KNOB < 4 && H == S && W == S) {
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
        if (KNOB < 2 && xoct) {
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
            if (KNOB >= 6 || x_mean != nullptr) { m[c] = x_mean[n * C + c0 + c]; r[c] = x_rstd[n * C + c0 + c]; }
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
        if (KNOB >= 6 || out != nullptr) {
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                out[((long long)n * 2 * C + c0 + c) * HW + pix] = v1[c];
                out[((long long)n * 2 * C + C + c0 + c) * HW + pix] = v2[c];
            }
        }
#if KNOB < 1
        if (xs != nullptr && s2d) {
            store_split_slot_s2d(xs, n, (2 * C) >> 3, by, H, W, oy, ox, v1);
            store_split_slot_s2d(xs, n, (2 * C) >> 3, (C >> 3) + by, H, W, oy, ox, v2);
        } else if (xs != nullptr) {
            store_split_slot(xs, n, (2 * C) >> 3, by, HW, pix, v1);
            store_split_slot(xs, n, (2 * C) >> 3, (C >> 3) + by, HW, pix, v2);
        }
#endif
        return;
    }
    if (KNOB >= 3) return;
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
}  // namespace
extern "C" hipError_t CAT(launch_victim_k, KNOB)(const float* x, const float* mean, const float* rstd, const float* motion, const float* flow,
                                                  const float* ifmask, float* out, int N, int C, int H, int W, int S, float flow_scale, hipStream_t stream) {
    dim3 grid((H * W + 255) / 256, (C + 7) / 8, N);
    hipLaunchKernelGGL((CAT(vk, KNOB)::warp_concat_kernel<1, 4, 0>), grid, dim3(256), 0, stream, x, mean, rstd, 1, motion, flow, ifmask, out,
                       (uint4*)nullptr, C, H, W, S, flow_scale, 0, 5);
    return hipGetLastError();
}
