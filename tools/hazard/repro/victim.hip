// victim.hip -- library-free reproducer of the co-residency hazard (profiles/r04_cohazard.md), part 2 of 3.
// The gather kernel that computes wrong values beside the aggressor, reduced in steps.  LEVEL selects how much of it is left:
//   0  the whole kernel: per output pixel of a 64 x 64 map, bilinear interpolation (align_corners = True) of a 256 x 256 float2
//      sampling map + a 2-channel flow + a mask, two 4-tap samplers, 2 x 4 x 8 gathered feature values with InstanceNorm + ReLU
//      applied per tap, 16 coalesced stores (what ap_warp_concat_fwd of libapamd.so runs at the generator's third warp level)
//   1  the same without the second sampler (flow / mask): 32 gathers, 8 stores
//   2  only the front part: the interpolated sampling coordinate and the four tap offsets / weights of the first sampler are
//      stored, no feature gather at all
// Deterministic, no atomics: two launches on the same inputs must agree bit for bit.
//     hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -c victim.hip
#include <hip/hip_runtime.h>
#include <cstdint>

struct Taps { int off[4]; float w[4]; };

__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W) {
    float ix = ((gx + 1.f) * (float)W - 1.f) / 2.f, iy = ((gy + 1.f) * (float)H - 1.f) / 2.f;
    ix = fminf(fmaxf(ix, -2.f), (float)W + 1.f);
    iy = fminf(fmaxf(iy, -2.f), (float)H + 1.f);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float ex = fx + 1.f, ey = fy + 1.f;
    Taps t;
    t.w[0] = (ex - ix) * (ey - iy); t.w[1] = (ix - fx) * (ey - iy); t.w[2] = (ex - ix) * (iy - fy); t.w[3] = (ix - fx) * (iy - fy);
    const bool xin0 = x0 >= 0 && x0 < W, xin1 = x1 >= 0 && x1 < W, yin0 = y0 >= 0 && y0 < H, yin1 = y1 >= 0 && y1 < H;
    t.off[0] = (xin0 && yin0) ? y0 * W + x0 : -1; t.off[1] = (xin1 && yin0) ? y0 * W + x1 : -1;
    t.off[2] = (xin0 && yin1) ? y1 * W + x0 : -1; t.off[3] = (xin1 && yin1) ? y1 * W + x1 : -1;
    return t;
}
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp make_lerp(int dst, int S, int H) {
    const float scale = H > 1 ? (float)(S - 1) / (float)(H - 1) : 0.f, src = scale * (float)dst;
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
// workgroup b runs on XCD b % 8: give every XCD a contiguous range of the logical block list (as the product does)
__device__ __forceinline__ unsigned xcd_logical_block(unsigned nblk, unsigned b) {
    const unsigned q = nblk >> 3, r = nblk & 7, xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// grid (H * W / 256, C / 8, N), 256 threads = a 32 x 8 pixel tile; out: [N][2C][H][W] (LEVEL 0), [N][C][H][W] (1), [N][C/8][10][H][W] (2)
template <int LEVEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void victim_kernel(
    const float* __restrict__ x, const float* __restrict__ x_mean, const float* __restrict__ x_rstd, const float* __restrict__ motion,
    const float* __restrict__ flow, const float* __restrict__ ifmask, float* __restrict__ out, int C, int H, int W, int S, float flow_scale) {
    int bx, by, bz;
    {
        const unsigned L = xcd_logical_block(gridDim.x * gridDim.y * gridDim.z, blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z));
        bx = L % gridDim.x;
        const unsigned t = L / gridDim.x;
        by = t % gridDim.y;
        bz = t / gridDim.y;
    }
    const int tiles_x = W >> 5, ty = bx / tiles_x, tx = bx - ty * tiles_x;
    const int oy = ty * 8 + ((int)threadIdx.x >> 5), ox = (tx << 5) + ((int)threadIdx.x & 31), pix = oy * W + ox;
    const int n = bz;
    const long long SS = (long long)S * S;
    const Lerp ly = make_lerp(oy, S, H), lx = make_lerp(ox, S, W);
    const int o00 = ly.i0 * S + lx.i0, o01 = ly.i0 * S + lx.i1, o10 = ly.i1 * S + lx.i0, o11 = ly.i1 * S + lx.i1;
    const float2* mo = reinterpret_cast<const float2*>(motion) + n * SS;
    const float2 a4 = mo[o00], b4 = mo[o01], c4 = mo[o10], d4 = mo[o11];
    const float gx = bilerp(a4.x, b4.x, c4.x, d4.x, ly, lx), gy = bilerp(a4.y, b4.y, c4.y, d4.y, ly, lx);
    const Taps tm = make_taps(gx, gy, H, W);
    const int HW = H * W, c0 = by * 8;
    if (LEVEL == 2) {
        float* o = out + ((long long)n * gridDim.y + by) * 10 * HW + pix;
        o[0] = gx; o[HW] = gy;
        for (int k = 0; k < 4; ++k) { o[(2 + k) * HW] = (float)tm.off[k]; o[(6 + k) * HW] = tm.w[k]; }
        return;
    }
    float fx = 0.f, fy = 0.f, mk = 1.f;
    if (LEVEL == 0) {
        const float* f0 = flow + (n * 2 + 0) * SS;
        const float* f1 = flow + (n * 2 + 1) * SS;
        fx = bilerp(f0[o00] * flow_scale, f0[o01] * flow_scale, f0[o10] * flow_scale, f0[o11] * flow_scale, ly, lx);
        fy = bilerp(f1[o00] * flow_scale, f1[o01] * flow_scale, f1[o10] * flow_scale, f1[o11] * flow_scale, ly, lx);
        const float* mp = ifmask + n * SS;
        mk = bilerp(mp[o00], mp[o01], mp[o10], mp[o11], ly, lx);
    }
    const float wgx = 2.0f * ((float)ox + fx) / (float)(W - 1 > 1 ? W - 1 : 1) - 1.0f;
    const float wgy = 2.0f * ((float)oy + fy) / (float)(H - 1 > 1 ? H - 1 : 1) - 1.0f;
    const Taps tf = make_taps(wgx, wgy, H, W);
    const bool keep = mk > 0.5f;
    int om[4], of[4];
    float wm[4], wf[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        om[k] = tm.off[k] < 0 ? 0 : tm.off[k];
        wm[k] = tm.off[k] < 0 ? 0.f : tm.w[k];
        of[k] = (tf.off[k] < 0 || !keep) ? 0 : tf.off[k];
        wf[k] = tf.off[k] < 0 ? 0.f : tf.w[k];
    }
    float a[8][4], b[8][4], m[8], r[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) {
        const float* plane = x + ((long long)n * C + c0 + c) * HW;
        m[c] = x_mean[n * C + c0 + c];
        r[c] = x_rstd[n * C + c0 + c];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            a[c][k] = plane[om[k]];
            if (LEVEL == 0) b[c][k] = plane[of[k]];
        }
    }
    typedef float f2 __attribute__((ext_vector_type(2)));
    auto relu2 = [](f2 t) -> f2 { return f2{fmaxf(t.x, 0.f), fmaxf(t.y, 0.f)}; };
    const int CO = LEVEL == 0 ? 2 * C : C;
#pragma unroll
    for (int c = 0; c < 8; c += 2) {
        const f2 mm = {m[c], m[c + 1]}, rr = {r[c], r[c + 1]};
        f2 s1 = relu2((f2{a[c][0], a[c + 1][0]} - mm) * rr) * wm[0];
#pragma unroll
        for (int k = 1; k < 4; ++k) s1 += relu2((f2{a[c][k], a[c + 1][k]} - mm) * rr) * wm[k];
        out[((long long)n * CO + c0 + c) * HW + pix] = s1.x;
        out[((long long)n * CO + c0 + c + 1) * HW + pix] = s1.y;
        if (LEVEL == 0) {
            f2 s2 = relu2((f2{b[c][0], b[c + 1][0]} - mm) * rr) * wf[0];
#pragma unroll
            for (int k = 1; k < 4; ++k) s2 += relu2((f2{b[c][k], b[c + 1][k]} - mm) * rr) * wf[k];
            out[((long long)n * CO + C + c0 + c) * HW + pix] = keep ? s2.x : -1.f;
            out[((long long)n * CO + C + c0 + c + 1) * HW + pix] = keep ? s2.y : -1.f;
        }
    }
}


// ---- LEVEL 20 / 21: the suspected core of the hazard alone.  What separates the failing kernels from the clean ones on BOTH sides is
// VOP3 instructions that take a lane mask from an SGPR pair (v_cndmask_b32_e64 ..., s[n:n+1]); the wrong lanes are 16-31 and 48-63,
// i.e. the upper halves of the pair's two dwords.  Here a lane mask is made by v_cmp_lt_f32_e64 into an SGPR pair, kept there
// (level 20: across a chain of dependent gathers, as `keep` is in the product kernel; level 21: used at once, 64 times in a row)
// and consumed by v_cndmask_b32_e64.  in: 2^19 floats in [0, 1); out: 2^21 floats.
template <int LEVEL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4, 4))) void mask_victim_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int tid = blockIdx.x * 256 + threadIdx.x, n = 1 << 19;
    const float mk = in[tid & (n - 1)];
    if (LEVEL == 20) {
        unsigned long long m;
        asm volatile("v_cmp_lt_f32_e64 %0, 0.5, %1" : "=s"(m) : "v"(mk));
        float acc = 0.f;
        int idx = tid;
        for (int it = 0; it < iters; ++it) {
            idx = (idx * 1103515245 + 12345) & (n - 1);
            acc += in[idx];
        }
        float r;
        asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(r) : "v"(-1.0f), "v"(acc), "s"(m));
        out[tid] = r;
    } else {
        float r = 0.f, thr = 0.25f;
        for (int it = 0; it < 64; ++it) {
            unsigned long long m;
            float t;
            asm volatile("v_cmp_lt_f32_e64 %0, %1, %2" : "=s"(m) : "v"(thr), "v"(mk));
            asm volatile("v_cndmask_b32_e64 %0, %1, %2, %3" : "=v"(t) : "v"(-1.0f), "v"(thr), "s"(m));
            r += t;
            thr += 0.0078125f;
        }
        out[tid] = r;
    }
}

extern "C" size_t victim_out_floats(int level, int N, int C, int H, int W) {
    if (level >= 20) return (size_t)1 << 21;
    return (size_t)N * H * W * (level == 0 ? 2 * C : (level == 1 ? C : (C / 8) * 10));
}
extern "C" hipError_t launch_victim(int level, const float* x, const float* mean, const float* rstd, const float* motion, const float* flow,
                                    const float* ifmask, float* out, int N, int C, int H, int W, int S, float flow_scale, hipStream_t stream) {
    if (level == 20) { hipLaunchKernelGGL(mask_victim_kernel<20>, dim3(8192), dim3(256), 0, stream, ifmask, out, 12); return hipGetLastError(); }
    if (level == 21) { hipLaunchKernelGGL(mask_victim_kernel<21>, dim3(8192), dim3(256), 0, stream, ifmask, out, 0); return hipGetLastError(); }
    dim3 grid(H * W / 256, C / 8, N);
    if (level == 0) hipLaunchKernelGGL(victim_kernel<0>, grid, dim3(256), 0, stream, x, mean, rstd, motion, flow, ifmask, out, C, H, W, S, flow_scale);
    else if (level == 1) hipLaunchKernelGGL(victim_kernel<1>, grid, dim3(256), 0, stream, x, mean, rstd, motion, flow, ifmask, out, C, H, W, S, flow_scale);
    else hipLaunchKernelGGL(victim_kernel<2>, grid, dim3(256), 0, stream, x, mean, rstd, motion, flow, ifmask, out, C, H, W, S, flow_scale);
    return hipGetLastError();
}

// number of 32-bit words of a that differ from ref (added to *count), and the lane (index mod 64) histogram of the differences
__global__ void count_diff_kernel(const unsigned* __restrict__ a, const unsigned* __restrict__ ref, size_t n, unsigned long long* count, unsigned* lane_hist) {
    unsigned long long c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        if (a[i] != ref[i]) { ++c; atomicAdd(&lane_hist[i & 63], 1u); }
    if (c) atomicAdd(count, c);
}
extern "C" hipError_t launch_count_diff(const void* a, const void* ref, size_t n_words, unsigned long long* count, unsigned* lane_hist, hipStream_t stream) {
    hipLaunchKernelGGL(count_diff_kernel, dim3(1024), dim3(256), 0, stream, (const unsigned*)a, (const unsigned*)ref, n_words, count, lane_hist);
    return hipGetLastError();
}
