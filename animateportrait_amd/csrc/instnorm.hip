// instnorm.hip -- InstanceNorm2d(affine=False) statistics finalisation and the residual apply pass.
// Reference: nn.InstanceNorm2d via get_norm_layer('instance'), Module2/models/networks.py:33-34;
// residual adds of ResnetBlock / ResnetBlock2, networks.py:2358-2360, 2418-2420.
// Both kernels are HBM-bound streaming passes (float4 per lane).
#include "common.h"

namespace apamd {

// One wave per (n, c) plane: the conv epilogue's per-tile (sum, sumsq) are combined in fp64 (sums of fp32 values in fp64
// are exact here, so the lane grouping does not change the result).
//
// E[x^2] - E[x]^2 from fp32 tile sums loses log2(mean^2 / var) bits: a plane whose |mean| is many standard deviations
// (a PatchGAN fed a mostly-white masked crop, base_model.py:245-247) would get a variance that is rounding noise,
// hidden by the var > 0 clamp.  When `y` is given, planes with mean^2 > kInstNormRefineRatio * var are therefore
// recomputed from the data with the shifted two-pass formula (shift = the first estimate of the mean, itself
// accurate) by the same wave -- a well-conditioned plane skips that loop.  The same rule is inlined in
// norm_split_kernel.
__global__ __launch_bounds__(64) void instnorm_finalize_kernel(const float* __restrict__ partials,
                                                               const float* __restrict__ y, int tiles, int HW,
                                                               float eps, float* __restrict__ mean,
                                                               float* __restrict__ rstd, int oct_C) {
    const int i = blockIdx.x, lane = threadIdx.x;
    const double inv_count = 1.0 / (double)HW;
    double s = 0.0, q = 0.0;
    const float2* p = reinterpret_cast<const float2*>(partials) + (long long)i * tiles;
    for (int t = lane; t < tiles; t += 64) {
        const float2 v = p[t];
        s += (double)v.x;
        q += (double)v.y;
    }
#pragma unroll
    for (int sh = 1; sh < 64; sh <<= 1) {
        s += __shfl_xor(s, sh, 64);
        q += __shfl_xor(q, sh, 64);
    }
    double m = s * inv_count;
    double var = q * inv_count - m * m;   // biased variance, as F.instance_norm
    var = var > 0.0 ? var : 0.0;
    if (y != nullptr && m * m > (double)kInstNormRefineRatio * var) {       // wave-uniform
        const float m0 = (float)m;
        // oct_C > 0: y is the channel-octet layout [n][C/8][HW][8] (ap_conv2d_fwd_octet): plane (n, c) starts at octet
        // c / 8 of image n, its elements are 8 floats apart
        const float* py = y + (long long)i * HW;
        int ys = 1;
        if (oct_C > 0) {
            const int n = i / oct_C, c = i - n * oct_C;
            py = y + ((long long)n * (oct_C >> 3) + (c >> 3)) * HW * 8 + (c & 7);
            ys = 8;
        }
        double ds = 0.0, dq = 0.0;
        for (int k = lane; k < HW; k += 64) {
            const float d = py[(long long)k * ys] - m0;
            ds += (double)d;
            dq += (double)d * (double)d;
        }
#pragma unroll
        for (int sh = 1; sh < 64; sh <<= 1) {
            ds += __shfl_xor(ds, sh, 64);
            dq += __shfl_xor(dq, sh, 64);
        }
        const double dm = ds * inv_count;
        var = dq * inv_count - dm * dm;
        var = var > 0.0 ? var : 0.0;
        m = (double)m0 + dm;
    }
    if (lane == 0) {
        mean[i] = (float)m;
        rstd[i] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

__device__ __forceinline__ float act1(float v, int act) {
    if (act == 1) return v > 0.f ? v : 0.f;
    if (act == 2) return v > 0.f ? v : 0.2f * v;
    return v;
}

// grid: (ceil(HW/4/256), NC).  HW % 4 == 0 is required (checked on the host).
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x, const float* __restrict__ mean,
                                                             const float* __restrict__ rstd, int act,
                                                             const float* __restrict__ res,
                                                             const float* __restrict__ res_mean,
                                                             const float* __restrict__ res_rstd,
                                                             float* __restrict__ out, int HW4) {
    const int nc = blockIdx.y;
    const float m = mean[nc], r = rstd[nc];
    float rm = 0.f, rr = 1.f;
    if (res_mean != nullptr) { rm = res_mean[nc]; rr = res_rstd[nc]; }
    const float4* x4 = reinterpret_cast<const float4*>(x) + (long long)nc * HW4;
    const float4* r4 = res ? reinterpret_cast<const float4*>(res) + (long long)nc * HW4 : nullptr;
    float4* o4 = reinterpret_cast<float4*>(out) + (long long)nc * HW4;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < HW4; i += gridDim.x * 256) {
        float4 v = x4[i];
        v.x = act1((v.x - m) * r, act);
        v.y = act1((v.y - m) * r, act);
        v.z = act1((v.z - m) * r, act);
        v.w = act1((v.w - m) * r, act);
        if (r4) {
            const float4 q = r4[i];
            v.x += (q.x - rm) * rr;
            v.y += (q.y - rm) * rr;
            v.z += (q.z - rm) * rr;
            v.w += (q.w - rm) * rr;
        }
        o4[i] = v;
    }
}

}  // namespace apamd

using namespace apamd;

extern "C" {

int ap_instnorm_finalize(const float* stat_partials, const float* y, int32_t NC, int32_t tiles, int32_t count, float eps,
                         float* mean, float* rstd, ap_stream_t stream) {
    if (!stat_partials || !mean || !rstd) return fail(AP_ERR_INVALID, "instnorm_finalize: null pointer");
    if (NC < 1 || tiles < 1 || count < 1) return fail(AP_ERR_INVALID, "instnorm_finalize: bad sizes");
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(NC), dim3(64), 0, (hipStream_t)stream, stat_partials, y, tiles, count,
                       eps, mean, rstd, 0);
    return check_launch("instnorm_finalize_kernel");
}

int ap_instnorm_finalize_octet(const float* stat_partials, const float* y, int32_t N, int32_t C, int32_t tiles, int32_t count,
                               float eps, float* mean, float* rstd, ap_stream_t stream) {
    if (!stat_partials || !mean || !rstd) return fail(AP_ERR_INVALID, "instnorm_finalize_octet: null pointer");
    if (N < 1 || C < 8 || (C & 7) || tiles < 1 || count < 1) return fail(AP_ERR_INVALID, "instnorm_finalize_octet: bad sizes");
    hipLaunchKernelGGL(instnorm_finalize_kernel, dim3(N * C), dim3(64), 0, (hipStream_t)stream, stat_partials, y, tiles, count,
                       eps, mean, rstd, C);
    return check_launch("instnorm_finalize_kernel");
}

int ap_instnorm_apply(const float* x, const float* mean, const float* rstd, int32_t act, const float* res,
                      const float* res_mean, const float* res_rstd, float* out, int32_t NC, int32_t HW,
                      ap_stream_t stream) {
    if (!x || !mean || !rstd || !out) return fail(AP_ERR_INVALID, "instnorm_apply: null pointer");
    if ((res_mean == nullptr) != (res_rstd == nullptr) || (res_mean && !res))
        return fail(AP_ERR_INVALID, "instnorm_apply: inconsistent residual arguments");
    if (act < 0 || act > 2) return fail(AP_ERR_INVALID, "instnorm_apply: act %d", act);
    if (NC < 1 || HW < 4 || (HW & 3)) return fail(AP_ERR_UNSUPPORTED, "instnorm_apply: H*W=%d must be a multiple of 4", HW);
    const int HW4 = HW / 4;
    int bx = (HW4 + 255) / 256;
    if (bx > 64) bx = 64;
    if (NC > 65535) return fail(AP_ERR_UNSUPPORTED, "instnorm_apply: N*C=%d too large", NC);
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3(bx, NC), dim3(256), 0, (hipStream_t)stream, x, mean, rstd, act,
                       res, res_mean, res_rstd, out, HW4);
    return check_launch("instnorm_apply_kernel");
}

}  // extern "C"
