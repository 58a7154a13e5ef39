// tps.hip -- polyharmonic-spline (order 2) sparse image warp, batched, on device.
//
// Reference: Module2/models/sparse_image_warp.py:35-58 and callees (solve_interpolation :93-132,
// phi :157-183, apply_interpolation :186-217, dense_image_warp :220-264, interpolate_bilinear :267-361).
// The reference runs it on one sample at a time (its bmm does not broadcast, and torch.solve drops to pdb
// on failure); here B samples are solved by B workgroups.
//   tps_solve_kernel : builds [[phi(|ci-cj|^2), [c 1]], [[c 1]^T, 0]] (n+3)^2 in LDS and solves it for the
//                      two flow components by Gaussian elimination with partial pivoting (what LAPACK
//                      gesv, behind torch.solve, does); squared distances as |x|^2 - 2x.y + |y|^2 in fp32.
//   tps_warp_kernel  : per pixel q=(row,col): flow(q) = sum_i phi(|q-ci|^2) w_i + [q 1] v, then
//                      out(q) = bilinear(img, q - flow(q)) with floor clamped to [0,size-2] and alpha to [0,1].
#include "common.h"

namespace apamd {

constexpr int kTpsMaxN = 125;   // control points (+3 <= 128)

__device__ __forceinline__ float phi2(float r) { return 0.5f * r * logf(fmaxf(r, 1e-10f)); }

// grid: (B); block 1024.  src/dst: B x n x 2 (row, col).  coef: B x (n+3) x 2 (w rows then v rows)
__global__ __launch_bounds__(1024) void tps_solve_kernel(const float* __restrict__ src, const float* __restrict__ dst,
                                                        int n, float* __restrict__ coef, int* __restrict__ status) {
    extern __shared__ float sm[];
    const int m = n + 3, ld = m + 2;           // augmented matrix [A | f0 f1]
    float* A = sm;                              // m x ld
    float* cx = sm + m * ld;                    // n
    float* cy = cx + n;                         // n
    __shared__ int piv_row;
    __shared__ float piv_val;
    const int b = blockIdx.x, tid = threadIdx.x, nt = blockDim.x;   // 1024 threads: the elimination steps are LDS-latency bound
    const float* s = src + (long long)b * n * 2;
    const float* d = dst + (long long)b * n * 2;
    for (int i = tid; i < n; i += nt) { cx[i] = d[i * 2]; cy[i] = d[i * 2 + 1]; }
    __syncthreads();
    for (int e = tid; e < m * ld; e += nt) {
        const int i = e / ld, j = e - i * ld;
        float v = 0.f;
        if (i < n && j < n) {
            const float ni = cx[i] * cx[i] + cy[i] * cy[i], nj = cx[j] * cx[j] + cy[j] * cy[j];
            const float dot = cx[i] * cx[j] + cy[i] * cy[j];
            v = phi2(ni - 2.f * dot + nj);
        } else if (i < n && j < m) {
            v = j == n ? cx[i] : (j == n + 1 ? cy[i] : 1.f);
        } else if (i >= n && j < n) {
            v = i == n ? cx[j] : (i == n + 1 ? cy[j] : 1.f);
        } else if (i < n && j >= m) {
            v = d[i * 2 + (j - m)] - s[i * 2 + (j - m)];      // control-point flow = dst - src
        }
        A[e] = v;
    }
    __syncthreads();
    bool singular = false;
    for (int k = 0; k < m; ++k) {
        if (tid < 64) {                                        // partial pivoting: one wave scans column k
            float best = -1.f;
            int bi = k;
            for (int i = k + tid; i < m; i += 64) {
                const float a = fabsf(A[i * ld + k]);
                if (a > best) { best = a; bi = i; }
            }
            for (int sh = 32; sh > 0; sh >>= 1) {
                const float ob = __shfl_xor(best, sh, 64);
                const int oi = __shfl_xor(bi, sh, 64);
                if (ob > best || (ob == best && oi < bi)) { best = ob; bi = oi; }
            }
            if (tid == 0) { piv_row = bi; piv_val = best; }
        }
        __syncthreads();
        const int pr = piv_row;
        if (piv_val == 0.f) { singular = true; break; }
        if (pr != k) {
            for (int j = tid; j < ld; j += nt) {
                const float t = A[k * ld + j];
                A[k * ld + j] = A[pr * ld + j];
                A[pr * ld + j] = t;
            }
        }
        __syncthreads();
        const float inv = 1.f / A[k * ld + k];
        // (nt / 32) rows x 32 columns of the trailing block per trip; same products, same order as a flat walk
        for (int i = k + 1 + (tid >> 5); i < m; i += nt >> 5) {
            const float f = A[i * ld + k] * inv;
            for (int j = k + 1 + (tid & 31); j < ld; j += 32) A[i * ld + j] -= f * A[k * ld + j];
        }
        __syncthreads();
    }
    if (singular) {
        if (tid == 0 && status) atomicExch(status, 1);
        for (int e = tid; e < m * 2; e += nt) coef[(long long)b * m * 2 + e] = 0.f;
        return;
    }
    // back substitution for the two right-hand sides, column-oriented: x[i] is final once the rows below it are done, and
    // every row above subtracts its A[r][i] x[i] at once (thread = (row, right-hand side)).  (Two threads walking the
    // triangle row by row -- 2.5 K dependent LDS round trips each -- were most of the kernel's 190 us.)
    for (int i = m - 1; i >= 0; --i) {
        if (tid < 2) A[i * ld + m + tid] = A[i * ld + m + tid] / A[i * ld + i];
        __syncthreads();
        if (tid < 2 * i) {
            const int r = tid >> 1, c = tid & 1;
            A[r * ld + m + c] -= A[r * ld + i] * A[i * ld + m + c];
        }
        __syncthreads();
    }
    for (int e = tid; e < m * 2; e += nt) coef[(long long)b * m * 2 + e] = A[(e >> 1) * ld + m + (e & 1)];
}

// grid: (ceil(H*W/256), B).  img/out: B x C x H x W; dst: B x n x 2; coef: B x (n+3) x 2; flow (optional): B x H x W x 2
__global__ __launch_bounds__(256) void tps_warp_kernel(const float* __restrict__ img, const float* __restrict__ dst,
                                                       const float* __restrict__ coef, int n, int C, int H, int W,
                                                       float* __restrict__ out, float* __restrict__ flow_out) {
    __shared__ float cx[kTpsMaxN], cy[kTpsMaxN], cn[kTpsMaxN], w0[kTpsMaxN + 3], w1[kTpsMaxN + 3];
    const int b = blockIdx.y, tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) {
        const float x = dst[((long long)b * n + i) * 2], y = dst[((long long)b * n + i) * 2 + 1];
        cx[i] = x; cy[i] = y; cn[i] = x * x + y * y;
    }
    for (int i = tid; i < n + 3; i += 256) {
        w0[i] = coef[((long long)b * (n + 3) + i) * 2];
        w1[i] = coef[((long long)b * (n + 3) + i) * 2 + 1];
    }
    __syncthreads();
    const int pix = blockIdx.x * 256 + tid;
    if (pix >= H * W) return;
    const float qr = (float)(pix / W), qc = (float)(pix % W);
    const float qn = qr * qr + qc * qc;
    float f0 = 0.f, f1 = 0.f;
    for (int i = 0; i < n; ++i) {
        const float ph = phi2(qn - 2.f * (qr * cx[i] + qc * cy[i]) + cn[i]);
        f0 += ph * w0[i];
        f1 += ph * w1[i];
    }
    f0 += qr * w0[n] + qc * w0[n + 1] + w0[n + 2];
    f1 += qr * w1[n] + qc * w1[n + 1] + w1[n + 2];
    if (flow_out) {
        flow_out[((long long)b * H * W + pix) * 2] = f0;
        flow_out[((long long)b * H * W + pix) * 2 + 1] = f1;
    }
    const float sy = qr - f0, sx = qc - f1;
    const float fy = fminf(fmaxf(floorf(sy), 0.f), (float)(H - 2));
    const float fx = fminf(fmaxf(floorf(sx), 0.f), (float)(W - 2));
    const float ay = fminf(fmaxf(sy - fy, 0.f), 1.f), ax = fminf(fmaxf(sx - fx, 0.f), 1.f);
    const int y0 = (int)fy, x0 = (int)fx;
    for (int c = 0; c < C; ++c) {
        const float* p = img + ((long long)b * C + c) * H * W;
        const float tl = p[y0 * W + x0], tr = p[y0 * W + x0 + 1], bl = p[(y0 + 1) * W + x0], br = p[(y0 + 1) * W + x0 + 1];
        const float top = ax * (tr - tl) + tl, bot = ax * (br - bl) + bl;
        out[((long long)b * C + c) * H * W + pix] = ay * (bot - top) + top;
    }
}

// fused Adam (torch.optim.Adam semantics, no weight decay / amsgrad): geomgm_ifw_fore_model.py:346-360
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                            float* __restrict__ v, long long n, float lr, float b1, float b2, float eps, float bc1,
                            float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = b1 * m[i] + (1.f - b1) * gi;
        const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] -= (lr / bc1) * (mi / denom);
    }
}

}  // namespace apamd

using namespace apamd;

extern "C" {

int ap_tps_solve(const float* src, const float* dst, int32_t B, int32_t n, float* coef, int32_t* status,
                 ap_stream_t stream) {
    if (!src || !dst || !coef) return fail(AP_ERR_INVALID, "tps_solve: null pointer");
    if (B < 1 || n < 3 || n > kTpsMaxN) return fail(AP_ERR_UNSUPPORTED, "tps_solve: n=%d (3..%d)", n, kTpsMaxN);
    const int m = n + 3;
    const size_t lds = ((size_t)m * (m + 2) + 2 * n) * sizeof(float);
    if (lds > 64 * 1024) return fail(AP_ERR_UNSUPPORTED, "tps_solve: system of %d unknowns does not fit LDS", m);
    hipLaunchKernelGGL(tps_solve_kernel, dim3(B), dim3(1024), lds, (hipStream_t)stream, src, dst, n, coef, status);
    return check_launch("tps_solve_kernel");
}

int ap_tps_warp(const float* img, const float* dst, const float* coef, int32_t B, int32_t n, int32_t C, int32_t H,
                int32_t W, float* out, float* flow_out, ap_stream_t stream) {
    if (!img || !dst || !coef || !out) return fail(AP_ERR_INVALID, "tps_warp: null pointer");
    if (B < 1 || B > 65535 || n < 3 || n > kTpsMaxN || C < 1 || H < 2 || W < 2)
        return fail(AP_ERR_INVALID, "tps_warp: bad sizes");
    hipLaunchKernelGGL(tps_warp_kernel, dim3((H * W + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, img, dst, coef,
                       n, C, H, W, out, flow_out);
    return check_launch("tps_warp_kernel");
}

int ap_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                 float beta2, float eps, int32_t step, ap_stream_t stream) {
    if (!param || !grad || !exp_avg || !exp_avg_sq || n < 1 || step < 1) return fail(AP_ERR_INVALID, "adam_step: bad arguments");
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    int blocks = (int)std::min<long long>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(adam_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg, exp_avg_sq,
                       (long long)n, lr, beta1, beta2, eps, bc1, sqrtf(bc2));
    return check_launch("adam_kernel");
}

}  // extern "C"
