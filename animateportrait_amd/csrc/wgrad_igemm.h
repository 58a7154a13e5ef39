// wgrad_igemm.h -- weight gradient of a convolution as an implicit GEMM over pixels (exact fp32 MFMA).
//
//   dW[m][q] = sum_{n, oy, ox} G[n][m][oy][ox] * A[n][ci][oy*S + ky - pad][ox*S + kx - pad],   q = ci*K*K + ky*K + kx
//
// For nn.Conv2d: G = gradient w.r.t. the conv output (m = cout), A = the conv's input, and dW[m][q] is
// exactly the OIHW weight-gradient tensor.  For nn.ConvTranspose2d(k, s=2): G = the layer INPUT (m = cin),
// A = the gradient w.r.t. its output, S = 2, and dW[m][q] is the IOHW tensor.  (What autograd computes for
// the layers of Module2/models/networks.py:1218-1282, 2329-2421, 2620-2643.)
//
// Both operands arrive as PLAIN, PADDED tensors produced by pad_materialize_kernel (one streaming pass per
// layer): A already carries the producer's InstanceNorm + activation, the concatenation of its source
// segments and the convolution's zero / reflection padding, and both are zero-filled out to the tile grid.
// The GEMM kernel therefore stages its tiles with unconditional, branch-free loads.
//
// GEMM mapping: M = m (lane = channel), N = q (lane = (ci, tap) with a per-lane precomputed LDS offset, so
// any Cin / kernel size packs densely: the 7x7 stems with Cin = 3 use 147 of 160 columns), K = pixels:
// the two half-waves hold two x-adjacent pixels.  Both operands are staged through LDS once per pixel
// tile (PR rows x 32 columns); the shifted taps are reads of the same A tile (no im2col buffer).
// Workgroup = 4 waves as 2 (m) x 2 (q), each wave MT x NT tiles of 32 x 32.  The pixel range is split
// over P workgroups per (m, q) tile; partial results go to a workspace and are summed by a second
// kernel in a fixed order (deterministic, no atomics).
#pragma once
#include "conv_igemm.h"

namespace apamd {

struct WgradKParams {
    const float* g;           // [N][M][GHp][GWp]   (GWp = tiles_x * 32, GHp = tiles_y * PR; zero padded)
    const float* a;           // [N][Cin][Hp][Wp]   (origin = padded coordinate (0,0) = input (-pad,-pad); zero filled)
    int N, M, Cin, Q;         // Q = Cin * K * K
    int GHp, GWp, Hp, Wp;
    int tiles_x, tiles_y;     // pixel tiles per image
    int nstages, P;           // total pixel tiles (N * tiles_y * tiles_x), number of splits
    int m_tiles, q_tiles;
    float* partial;           // [P][M][Q]
    int ablate;               // debugging only (APAMD_ABLATE): 1 = no refill of the pipeline buffers
};

template <int S_, int K_, int MT_, int NT_, int PR_>
struct WgradCfg {
    static constexpr int S = S_, K = K_, MT = MT_, NT = NT_, PR = PR_;
    static constexpr int T = K * K;
    static constexpr int M_TILE = 2 * MT * 32, Q_TILE = 2 * NT * 32;
    static constexpr int NPIX = PR * 32;
    static constexpr int GS = NPIX + 1;                  // odd row stride of the G tile: conflict-free A reads
    static constexpr int IH = (PR - 1) * S + K, IW = 31 * S + K;
    static constexpr int PLANE = (IH * IW) | 1;          // odd plane stride
    static constexpr int NCI = Q_TILE / T + 2;           // channels a q-tile can touch
    static constexpr int GE = M_TILE * NPIX, AE = NCI * PLANE;
    static constexpr int NG4 = (GE / 4 + 255) / 256, NA = (AE + 255) / 256;
    static size_t lds_floats() { return 2 * ((size_t)M_TILE * GS + (size_t)NCI * PLANE); }
};

template <class C>
__global__ __launch_bounds__(256, C::NA <= 20 ? 2 : 1) void wgrad_igemm_f32(const WgradKParams p) {
    constexpr int S = C::S, K = C::K, T = C::T, MT = C::MT, NT = C::NT, PR = C::PR;
    constexpr int GS = C::GS, IW = C::IW, PLANE = C::PLANE, NCI = C::NCI, NA = C::NA, NG4 = C::NG4;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const gbuf = smem;                                   // [2][M_TILE][GS]
    float* const abuf = gbuf + 2 * C::M_TILE * GS;              // [2][NCI][PLANE]

    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, half = lane >> 5, l32 = lane & 31;
    const int wm = wave & 1, wq = wave >> 1;
    int b = blockIdx.x;
    const int split = b % p.P; b /= p.P;
    const int qt = b % p.q_tiles;
    const int mt_ = b / p.q_tiles;
    const int m0 = mt_ * C::M_TILE, q0 = qt * C::Q_TILE;
    const int ci_lo = q0 / T;
    int ci_hi = (q0 + C::Q_TILE - 1) / T;
    if (ci_hi > p.Cin - 1) ci_hi = p.Cin - 1;
    const int nci = ci_hi - ci_lo + 1;
    const int st0 = (int)((long long)p.nstages * split / p.P), st1 = (int)((long long)p.nstages * (split + 1) / p.P);
    const int GHW = p.GHp * p.GWp, HW = p.Hp * p.Wp;

    // per-lane LDS offsets of this lane's NT columns (q -> channel plane + tap shift)
    int boff[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        int q = q0 + (wq * NT + t) * 32 + l32;
        if (q >= p.Q) q = q0;                     // masked at the store; keep the address legal
        const int ci = q / T, tap = q - ci * T;
        boff[t] = (ci - ci_lo) * PLANE + (tap / K) * IW + (tap % K) + half * S;
    }
    const int aoff = (wm * MT * 32 + l32) * GS + half;

    // ---- staging geometry, computed once: global offsets relative to the tile origin
    int g_off[NG4];                 // float4 quads of the G tile: element e4 -> (m, px..px+3)
#pragma unroll
    for (int k = 0; k < NG4; ++k) {
        const int e4 = tid + k * 256;
        int m = e4 / (C::NPIX / 4);
        const int px = (e4 % (C::NPIX / 4)) * 4;
        if (m0 + m >= p.M) m = 0;   // rows beyond M are never stored; any legal address will do
        g_off[k] = m * GHW + (px / 32) * p.GWp + (px & 31);
    }
    int a_off[NA];                  // scalars of the A tile: element e -> (cl, ly, lx)
#pragma unroll
    for (int k = 0; k < NA; ++k) {
        const int e = tid + k * 256;
        int cl = e / PLANE;
        const int r = e - cl * PLANE;
        int ly = r / IW;
        const int lx = r - ly * IW;
        if (cl >= nci) cl = 0;      // unused plane / pad element: harmless duplicate
        if (ly >= C::IH) ly = 0;
        a_off[k] = cl * HW + ly * p.Wp + lx;
    }

    float4 gr[NG4];
    float ar[NA];
    auto issue = [&](int st) __attribute__((always_inline)) {
        const int tx = st % p.tiles_x;
        int t2 = st / p.tiles_x;
        const int ty = t2 % p.tiles_y;
        const int n = t2 / p.tiles_y;
        const float* gsrc = p.g + ((long long)n * p.M + m0) * GHW + (long long)(ty * PR) * p.GWp + tx * 32;
        const float* asrc = p.a + ((long long)n * p.Cin + ci_lo) * HW + (long long)(ty * PR * S) * p.Wp + tx * 32 * S;
#pragma unroll
        for (int k = 0; k < NG4; ++k) gr[k] = *reinterpret_cast<const float4*>(gsrc + g_off[k]);
#pragma unroll
        for (int k = 0; k < NA; ++k) ar[k] = asrc[a_off[k]];
    };
    auto commit = [&](int buf) __attribute__((always_inline)) {
        float* gd = gbuf + buf * C::M_TILE * GS;
        float* ad = abuf + buf * NCI * PLANE;
#pragma unroll
        for (int k = 0; k < NG4; ++k) {
            const int e4 = tid + k * 256;
            if (e4 < C::GE / 4) {
                float* d = gd + (e4 / (C::NPIX / 4)) * GS + (e4 % (C::NPIX / 4)) * 4;
                d[0] = gr[k].x; d[1] = gr[k].y; d[2] = gr[k].z; d[3] = gr[k].w;
            }
        }
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const int e = tid + k * 256;
            if (e < C::AE) ad[e] = ar[k];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.f;

    if (st0 < st1) {
        issue(st0);
        commit(0);
    }
    __syncthreads();
    for (int st = st0; st < st1; ++st) {
        const int cur = (st - st0) & 1;
        const bool more = st + 1 < st1 && !AP_ABLATE(p, 1);
        if (more) issue(st + 1);
        const float* G = gbuf + cur * C::M_TILE * GS + aoff;
        const float* A = abuf + cur * NCI * PLANE;
#pragma unroll
        for (int py = 0; py < PR; ++py) {
#pragma unroll
            for (int sx = 0; sx < 16; ++sx) {
                float a[MT], bb[NT];
#pragma unroll
                for (int m = 0; m < MT; ++m) a[m] = G[m * 32 * GS + py * 32 + 2 * sx];
#pragma unroll
                for (int t = 0; t < NT; ++t) bb[t] = A[boff[t] + (py * S) * IW + 2 * sx * S];
#pragma unroll
                for (int m = 0; m < MT; ++m)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[m], bb[t], acc[m][t], 0, 0, 0);
            }
        }
        if (more) commit(cur ^ 1);
        __syncthreads();
    }

    // partial[split][m][q]: C/D layout col j = lane & 31 (q), row i = (r&3) + 8*(r>>2) + 4*half (m)
    float* out = p.partial + (long long)split * p.M * p.Q;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int q = q0 + (wq * NT + t) * 32 + l32;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int mm = m0 + (wm * MT + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                if (mm < p.M && q < p.Q) out[(long long)mm * p.Q + q] = acc[m][t][r];
            }
        }
}

// dW[i] = sum_s partial[s][i]   (fixed order)
__global__ void wgrad_reduce_kernel(const float* __restrict__ partial, int P, long long n, float* __restrict__ dw) {
    // (dw may be a slot of a network's gradient block: 4-byte aligned only when an odd-sized parameter precedes it)
    if ((n & 3) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0) {
        // 16-byte lanes, four partial buffers in flight per step (the sum order k = 0, 1, 2, ... is unchanged)
        const long long n4 = n >> 2;
        const float4* p4 = reinterpret_cast<const float4*>(partial);
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
            float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
            int k = 0;
            for (; k + 4 <= P; k += 4) {
                const float4 a = p4[(long long)k * n4 + i], b = p4[(long long)(k + 1) * n4 + i];
                const float4 c = p4[(long long)(k + 2) * n4 + i], d = p4[(long long)(k + 3) * n4 + i];
                s.x = (((s.x + a.x) + b.x) + c.x) + d.x;
                s.y = (((s.y + a.y) + b.y) + c.y) + d.y;
                s.z = (((s.z + a.z) + b.z) + c.z) + d.z;
                s.w = (((s.w + a.w) + b.w) + c.w) + d.w;
            }
            for (; k < P; ++k) {
                const float4 a = p4[(long long)k * n4 + i];
                s.x += a.x; s.y += a.y; s.z += a.z; s.w += a.w;
            }
            reinterpret_cast<float4*>(dw)[i] = s;
        }
        return;
    }
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        float s = 0.f;
        for (int k = 0; k < P; ++k) s += partial[(long long)k * n + i];
        dw[i] = s;
    }
}

// The same sum for SMALL gradients with MANY partials (the 7x7 stems: 9408 elements x 128 partials; the final layer:
// 3136 x 512; the narrow first layers: 72 .. 2048 x thousands): wgrad_reduce_kernel gives such a launch a handful of
// lanes that each walk all P partials one after the other (a single workgroup ran 234 us).  Here S = 2^k threads share
// an element: thread (element, slice s) adds partials s, s + S, s + 2S, ... in that order, and the S slice sums are
// combined by a fixed-order tree through LDS -- deterministic for a given (n, P).  Requires n % 4 == 0 and 16-byte
// aligned pointers (the caller checks).
__device__ __forceinline__ float4 f4add(const float4 a, const float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float f4add(const float a, const float b) { return a + b; }
template <typename T> __device__ __forceinline__ T f4zero();
template <> __device__ __forceinline__ float4 f4zero<float4>() { return make_float4(0.f, 0.f, 0.f, 0.f); }
template <> __device__ __forceinline__ float f4zero<float>() { return 0.f; }

// T = float4 (n % 4 == 0, 16-byte aligned pointers) or float (anything: gradient-block slots are 4-byte aligned);
// nT = elements in units of T
template <typename T>
__global__ __launch_bounds__(256) void wgrad_reduce_sliced_kernel(const float* __restrict__ partial, int P, long long nT,
                                                                  float* __restrict__ dw, int S) {
    __shared__ T red[256];
    const int IL = 256 / S;                                     // elements per workgroup
    const int li = threadIdx.x % IL, sl = threadIdx.x / IL;
    const long long i = (long long)blockIdx.x * IL + li;
    const T* pT = reinterpret_cast<const T*>(partial);
    T s = f4zero<T>();
    if (i < nT) {
        int k = sl;
        for (; k + 3 * S < P; k += 4 * S) {                     // four partials in flight; the order k, k + S, ... is kept
            const T a = pT[(long long)k * nT + i], b = pT[(long long)(k + S) * nT + i];
            const T c = pT[(long long)(k + 2 * S) * nT + i], d = pT[(long long)(k + 3 * S) * nT + i];
            s = f4add(f4add(f4add(f4add(s, a), b), c), d);
        }
        for (; k < P; k += S) s = f4add(s, pT[(long long)k * nT + i]);
    }
    red[threadIdx.x] = s;
    __syncthreads();
    for (int h = S >> 1; h >= 1; h >>= 1) {
        if (sl < h) red[threadIdx.x] = f4add(red[threadIdx.x], red[threadIdx.x + h * IL]);
        __syncthreads();
    }
    if (sl == 0 && i < nT) reinterpret_cast<T*>(dw)[i] = red[li];
}

// picks the form: slices when the plain kernel would run fewer than ~16 K threads over many partials
inline void launch_wgrad_reduce(hipStream_t stream, const float* partial, int P, long long n, float* dw) {
    const bool vec = (n & 3) == 0 && (reinterpret_cast<uintptr_t>(dw) & 15) == 0 && (reinterpret_cast<uintptr_t>(partial) & 15) == 0;
    const long long nT = vec ? n >> 2 : n;
    if (P >= 16 && nT < 16384) {
        int S = 2;
        while (S < 64 && S * 2 <= P / 4 && nT * S * 2 <= 65536) S *= 2;
        const int IL = 256 / S;
        const dim3 grid((unsigned)((nT + IL - 1) / IL));
        if (vec) hipLaunchKernelGGL(wgrad_reduce_sliced_kernel<float4>, grid, dim3(256), 0, stream, partial, P, nT, dw, S);
        else hipLaunchKernelGGL(wgrad_reduce_sliced_kernel<float>, grid, dim3(256), 0, stream, partial, P, nT, dw, S);
        return;
    }
    const int blocks = (int)std::min<long long>((n + 255) / 256, 4096);
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3(blocks), dim3(256), 0, stream, partial, P, n, dw);
}

// ---- operand preparation: out[n][c][y][x] (Hp x Wp) = padded view of act(IN(concat(src)))
//   (y, x) <-> source (y - pad, x - pad); reflection or zero padding inside [0, H+2pad) x [0, W+2pad), zeros beyond.
struct PadParams {
    SrcSeg seg[kMaxSeg];      // chunk_begin = first concat channel
    int nseg;
    int N, C, H, W, pad, pad_mode, Hp, Wp;
    float* out;
};

// grid: (ceil(Hp*Wp/256), C, N)
__global__ __launch_bounds__(256) void pad_materialize_kernel(const PadParams p) {
    const int c = blockIdx.y, n = blockIdx.z;
    int s = 0;
    if (p.nseg > 1 && c >= p.seg[1].chunk_begin) s = 1;
    if (p.nseg > 2 && c >= p.seg[2].chunk_begin) s = 2;
    // block-uniform segment: select once, by value
    const SrcSeg sg = s == 0 ? p.seg[0] : (s == 1 ? p.seg[1] : p.seg[2]);
    const int cs = c - sg.chunk_begin;
    float m = 0.f, r = 1.f;
    if (sg.mean != nullptr) { m = sg.mean[n * sg.C + cs]; r = sg.rstd[n * sg.C + cs]; }
    const float* src = sg.data + ((long long)n * sg.C + cs) * p.H * p.W;
    float* dst = p.out + ((long long)n * p.C + c) * p.Hp * p.Wp;
    const int He = p.H + 2 * p.pad, We = p.W + 2 * p.pad;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.Hp * p.Wp; i += gridDim.x * 256) {
        const int y = i / p.Wp, x = i - y * p.Wp;
        float v = 0.f;
        if (y < He && x < We) {
            int sy = y - p.pad, sx = x - p.pad;
            bool ok = true;
            if (p.pad_mode == 1) {
                sy = reflect_clamp(sy, p.H);
                sx = reflect_clamp(sx, p.W);
            } else {
                ok = sy >= 0 && sy < p.H && sx >= 0 && sx < p.W;
            }
            if (ok) {
                v = (src[sy * p.W + sx] - m) * r;
                v = sg.act == 1 ? fmaxf(v, 0.f) : (sg.act == 2 ? (v > 0.f ? v : 0.2f * v) : v);
            }
        }
        dst[i] = v;
    }
}

}  // namespace apamd
